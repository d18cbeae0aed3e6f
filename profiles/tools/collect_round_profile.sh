#!/bin/bash
# Everything DESIGN.md section 6 quotes, in one GPU-box call (run from the repo root):
#   bash profiles/tools/collect_round_profile.sh gpurun_out/final
# -> bench lines (C3 train/eval with CPU baseline, C1/C2/C4 train), rocprofv3 kernel stats of the C3 train step,
#    PMC traffic of the C3 kernels.  Copy what should be judged into profiles/<round>/.
set -e
OUT=${1:-gpurun_out/final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT
cd $R
bash profiles/tools/collect_traffic.sh C3 $OUT/traffic train
bash profiles/tools/collect_traffic.sh C3 $OUT/traffic_eval eval
mkdir -p profiles/r01
python3 - $OUT/traffic.traffic.json $OUT/traffic_eval.traffic.json profiles/r01/traffic.json <<'PY'
import json, sys
d = {}
for f in sys.argv[1:3]:
    d.update(json.load(open(f)))
json.dump(d, open(sys.argv[3], 'w'), indent=1)     # bench.py reads it below
PY
python3 bench.py > $OUT/bench_train_C3.json 2> $OUT/bench_train_C3.err
python3 bench.py --mode eval > $OUT/bench_eval_C3.json 2> $OUT/bench_eval_C3.err
for w in C1 C2 C4; do python3 bench.py --workload $w --mode train --no-cpu-baseline > $OUT/bench_train_$w.json 2> $OUT/bench_train_$w.err; done
python3 bench.py --workload C4 --mode eval --no-cpu-baseline > $OUT/bench_eval_C4.json 2> $OUT/bench_eval_C4.err
python3 bench.py --workload C5 --mode train --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_train_C5.json 2> $OUT/bench_train_C5.err
python3 bench.py --workload C5 --mode eval --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_eval_C5.json 2> $OUT/bench_eval_C5.err
python3 profiles/tools/dropin_throughput.py > $OUT/dropin_throughput.log 2>&1
cd /tmp && export TMPDIR=/tmp
CBFSSM_HIP_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/ktrace -o c3 -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $R/$OUT/ktrace.log 2>&1
cp $(ls $R/$OUT/ktrace/*kernel_stats.csv | head -1) $R/$OUT/train_C3_kernel_stats.csv
CBFSSM_HIP_GRAPH=0 CBFSSM_NO_SPLIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/ktrace_ns -o c3 -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $R/$OUT/bench_train_C3_nosplit_under_rocprof.json 2> $R/$OUT/ktrace_ns.log
cp $(ls $R/$OUT/ktrace_ns/*kernel_stats.csv | head -1) $R/$OUT/train_C3_kernel_stats_nosplit.csv
CBFSSM_HIP_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/ktrace_c4 -o c4 -- python3 $R/bench.py --workload C4 --mode train --steps 4 --warmup 2 --no-cpu-baseline > $R/$OUT/ktrace_c4.log 2>&1
cp $(ls $R/$OUT/ktrace_c4/*kernel_stats.csv | head -1) $R/$OUT/train_C4_kernel_stats.csv
ls -la $R/$OUT
