#!/bin/bash
# Everything DESIGN.md section 6 quotes, in one GPU-box call (run from the repo root):
#   bash profiles/tools/collect_round_profile.sh gpurun_out/final [r03]
# -> bench lines (C3 train/eval with CPU baseline, C1/C2/C4 train), rocprofv3 kernel stats of the C3 train step,
#    PMC traffic of the C3 kernels.  Copy what should be judged into profiles/<round>/.
# A whole collection takes about half an hour of box time; a gpurun call is limited to 20 minutes, so it runs in two parts:
#   ... gpurun_out/final r03 A   (PMC traffic, bench lines, kernel traces)      ... gpurun_out/final r03 B   (the rest)
set -e
PART=${3:-all}
OUT=${1:-gpurun_out/final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT
cd $R
RND=${2:-r03}
if [ "$PART" != "B" ]; then
bash profiles/tools/collect_traffic.sh C3 $OUT/traffic train
bash profiles/tools/collect_traffic.sh C3 $OUT/traffic_eval eval
bash profiles/tools/collect_traffic.sh C4 $OUT/traffic_c4 train      # stash mode: adjoint launches and contraction summed per step
mkdir -p profiles/$RND
python3 - $OUT/traffic.traffic.json $OUT/traffic_eval.traffic.json $OUT/traffic_c4.traffic.json profiles/$RND/traffic.json <<'PY'
import json, sys
d = {}
for f in sys.argv[1:4]:
    d.update(json.load(open(f)))
json.dump(d, open(sys.argv[4], 'w'), indent=1)     # bench.py reads it below
PY
python3 bench.py > $OUT/bench_train_C3.json 2> $OUT/bench_train_C3.err
python3 bench.py --mode eval > $OUT/bench_eval_C3.json 2> $OUT/bench_eval_C3.err
for w in C1 C2 C4; do python3 bench.py --workload $w --mode train --no-cpu-baseline > $OUT/bench_train_$w.json 2> $OUT/bench_train_$w.err; done
python3 bench.py --workload C4 --mode eval --no-cpu-baseline > $OUT/bench_eval_C4.json 2> $OUT/bench_eval_C4.err
python3 bench.py --workload C5 --mode train --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_train_C5.json 2> $OUT/bench_train_C5.err
python3 bench.py --workload C5 --mode eval --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_eval_C5.json 2> $OUT/bench_eval_C5.err
fi
if [ "$PART" != "A" ]; then
# trained-like parameters (cond(K_mm) 2e6: the two-triangular GP form a trained model runs in) and the float32 path
python3 bench.py --params trained --no-cpu-baseline > $OUT/bench_train_C3_trained.json 2> $OUT/bench_train_C3_trained.err
python3 bench.py --params trained --mode eval --no-cpu-baseline > $OUT/bench_eval_C3_trained.json 2> $OUT/bench_eval_C3_trained.err
python3 bench.py --workload C4 --params trained --mode train --no-cpu-baseline > $OUT/bench_train_C4_trained.json 2> $OUT/bench_train_C4_trained.err
for w in C3 C5; do for m in train eval; do python3 bench.py --workload $w --mode $m --dtype float32 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_${m}_${w}_f32.json 2> $OUT/bench_${m}_${w}_f32.err; done; done
python3 profiles/tools/dropin_throughput.py > $OUT/dropin_throughput.log 2>&1
fi
if [ "$PART" != "B" ]; then
cd /tmp && export TMPDIR=/tmp
CBFSSM_HIP_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/ktrace -o c3 -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $R/$OUT/ktrace.log 2>&1
cp $(ls $R/$OUT/ktrace/*kernel_stats.csv | head -1) $R/$OUT/train_C3_kernel_stats.csv
CBFSSM_HIP_GRAPH=0 CBFSSM_NO_SPLIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/ktrace_ns -o c3 -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $R/$OUT/bench_train_C3_nosplit_under_rocprof.json 2> $R/$OUT/ktrace_ns.log
cp $(ls $R/$OUT/ktrace_ns/*kernel_stats.csv | head -1) $R/$OUT/train_C3_kernel_stats_nosplit.csv
CBFSSM_HIP_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/ktrace_c4 -o c4 -- python3 $R/bench.py --workload C4 --mode train --steps 4 --warmup 2 --no-cpu-baseline > $R/$OUT/ktrace_c4.log 2>&1
cp $(ls $R/$OUT/ktrace_c4/*kernel_stats.csv | head -1) $R/$OUT/train_C4_kernel_stats.csv
python3 $R/profiles/tools/step_timeline.py $(ls $R/$OUT/ktrace_c4/*kernel_trace.csv | head -1) 4 > $R/$OUT/train_C4_step_timeline.txt
# ... and with the two directions on ONE stream (no overlap): every launch by itself, the averages bench.py's HIP events must agree with
CBFSSM_HIP_GRAPH=0 CBFSSM_NO_SPLIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/ktrace_c4_ns -o c4 -- python3 $R/bench.py --workload C4 --mode train --steps 6 --warmup 2 --no-cpu-baseline > $R/$OUT/bench_train_C4_nosplit_under_rocprof.json 2> $R/$OUT/ktrace_c4_ns.log
cp $(ls $R/$OUT/ktrace_c4_ns/*kernel_stats.csv | head -1) $R/$OUT/train_C4_kernel_stats_nosplit.csv
fi
if [ "$PART" != "A" ]; then
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
fi
ls -la $R/$OUT
