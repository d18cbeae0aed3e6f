#!/bin/bash
# Everything DESIGN.md section 6 quotes, in one GPU-box call (run from the repo root):
#   bash profiles/tools/collect_round_profile.sh gpurun_out/final [r02]
# -> bench lines (C3 train/eval with CPU baseline, C1/C2/C4 train), rocprofv3 kernel stats of the C3 train step,
#    PMC traffic of the C3 kernels.  Copy what should be judged into profiles/<round>/.
set -e
OUT=${1:-gpurun_out/final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT
cd $R
bash profiles/tools/collect_traffic.sh C3 $OUT/traffic train
bash profiles/tools/collect_traffic.sh C3 $OUT/traffic_eval eval
RND=${2:-r02}
mkdir -p profiles/$RND
python3 - $OUT/traffic.traffic.json $OUT/traffic_eval.traffic.json profiles/$RND/traffic.json <<'PY'
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
# MFMA utilisation counters of the four time-loop kernels (SQ_INSTS_VALU_MFMA_MOPS_F64, SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES)
cd $R
bash profiles/tools/collect_mfma_util.sh C3 $OUT/mfma_util train > $R/$OUT/mfma_util.log 2>&1 || true
# cost of the two-triangular GP form next to the dense one (C3 / C4 train + eval, C5 eval)
for f in dense tri; do
  for m in train eval; do CBFSSM_GP_FORM=$f python3 bench.py --mode $m --no-cpu-baseline > $OUT/form_${f}_C3_$m.json 2>/dev/null; done
  CBFSSM_GP_FORM=$f python3 bench.py --workload C4 --mode train --steps 6 --no-cpu-baseline > $OUT/form_${f}_C4_train.json 2>/dev/null
  CBFSSM_GP_FORM=$f python3 bench.py --workload C5 --mode eval --steps 3 --warmup 1 --no-cpu-baseline > $OUT/form_${f}_C5_eval.json 2>/dev/null
done
python3 profiles/tools/outputs_walltime.py 16 1000 > $OUT/outputs_walltime.log 2>&1 || true
# two ranks on this one GPU over gloo: the N > 1 bench path (C4 workload by default) end to end
CBFSSM_BENCH_ONE_DEVICE=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 1 > $OUT/bench_2ranks_one_device.json 2> $OUT/bench_2ranks_one_device.err || true
ls -la $R/$OUT
