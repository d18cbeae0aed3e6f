#!/bin/bash
# Quick A/B helper for the GPU box: bench lines (no CPU baseline) of the C3 train and eval steps, kernel times printed.
# usage: profiles/tools/quick_bench.sh TAG [WORKLOAD]      -> gpurun_out/qb_TAG_{train,eval}.json
tag=${1:-x}; wl=${2:-C3}
mkdir -p gpurun_out
for mode in train eval; do
  python bench.py --workload $wl --mode $mode --no-cpu-baseline > gpurun_out/qb_${tag}_${mode}.json 2> gpurun_out/qb_${tag}_${mode}.err || { tail -5 gpurun_out/qb_${tag}_${mode}.err; exit 1; }
done
python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
for mode in ("train", "eval"):
    d = json.loads(open("gpurun_out/qb_%s_%s.json" % (tag, mode)).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(tag, mode, "%.3f ms" % d["ms_per_step"], {k: round(v, 3) for k, v in r["kernel_ms"].items()}, "frac %.3f" % r["frac"],
          "main", {k: round(v.get("ms", 0), 3) if isinstance(v, dict) else v for k, v in (r.get("main_piece") or {}).items()})
PY
