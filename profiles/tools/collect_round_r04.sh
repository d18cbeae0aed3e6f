#!/bin/bash
# Everything DESIGN.md section 6.0 (round 4) quotes.  A gpurun call is limited to 20 minutes, so it runs in parts:
#   bash profiles/tools/collect_round_r04.sh gpurun_out/r04_final A|B|C|D
# A: PMC traffic (C3 train/eval, C4 train) + bench lines; B: kernel traces (C3, C4) + MFMA counters (C3, C4);
# C: C5 lines + C5 traffic; D: float32 / trained-like / variants / two-rank rehearsal.  Copy what should be judged into profiles/r04/.
set -e
OUT=${1:-gpurun_out/r04_final}
PART=${2:-A}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT
cd $R
b() { python3 bench.py "$@" --no-cpu-baseline; }
if [ "$PART" = "A" ]; then
bash profiles/tools/collect_traffic.sh C3 $OUT/traffic train
bash profiles/tools/collect_traffic.sh C3 $OUT/traffic_eval eval
bash profiles/tools/collect_traffic.sh C4 $OUT/traffic_c4 train      # stash mode: adjoint launches and contraction summed per step
python3 - $OUT/traffic.traffic.json $OUT/traffic_eval.traffic.json $OUT/traffic_c4.traffic.json $OUT/traffic_A.json <<'PY'
import json, sys
d = {}
for f in sys.argv[1:4]:
    d.update(json.load(open(f)))
json.dump(d, open(sys.argv[4], 'w'), indent=1)
PY
mkdir -p profiles/r04 && cp $OUT/traffic_A.json profiles/r04/traffic.json     # bench.py reads it below
python3 bench.py > $OUT/bench_train_C3.json 2> $OUT/bench_train_C3.err
b --mode eval > $OUT/bench_eval_C3.json 2> $OUT/bench_eval_C3.err
for w in C1 C2 C4; do b --workload $w --mode train > $OUT/bench_train_$w.json 2> $OUT/bench_train_$w.err; done
b --workload C4 --mode eval > $OUT/bench_eval_C4.json 2> $OUT/bench_eval_C4.err
fi
if [ "$PART" = "B" ]; then
cd /tmp && export TMPDIR=/tmp
CBFSSM_HIP_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/ktrace -o c3 -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $R/$OUT/ktrace.log 2>&1
cp $(ls $R/$OUT/ktrace/*kernel_stats.csv | head -1) $R/$OUT/train_C3_kernel_stats.csv
python3 $R/profiles/tools/step_timeline.py $(ls $R/$OUT/ktrace/*kernel_trace.csv | head -1) 4 > $R/$OUT/train_C3_step_timeline.txt
CBFSSM_HIP_GRAPH=0 CBFSSM_NO_SPLIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/ktrace_ns -o c3 -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $R/$OUT/bench_train_C3_nosplit_under_rocprof.json 2> $R/$OUT/ktrace_ns.log
cp $(ls $R/$OUT/ktrace_ns/*kernel_stats.csv | head -1) $R/$OUT/train_C3_kernel_stats_nosplit.csv
CBFSSM_HIP_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/ktrace_c4 -o c4 -- python3 $R/bench.py --workload C4 --mode train --steps 4 --warmup 2 --no-cpu-baseline > $R/$OUT/ktrace_c4.log 2>&1
cp $(ls $R/$OUT/ktrace_c4/*kernel_stats.csv | head -1) $R/$OUT/train_C4_kernel_stats.csv
python3 $R/profiles/tools/step_timeline.py $(ls $R/$OUT/ktrace_c4/*kernel_trace.csv | head -1) 4 > $R/$OUT/train_C4_step_timeline.txt
CBFSSM_HIP_GRAPH=0 CBFSSM_NO_SPLIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/ktrace_c4_ns -o c4 -- python3 $R/bench.py --workload C4 --mode train --steps 6 --warmup 2 --no-cpu-baseline > $R/$OUT/bench_train_C4_nosplit_under_rocprof.json 2> $R/$OUT/ktrace_c4_ns.log
cp $(ls $R/$OUT/ktrace_c4_ns/*kernel_stats.csv | head -1) $R/$OUT/train_C4_kernel_stats_nosplit.csv
cd $R
bash profiles/tools/collect_mfma_util.sh C3 $OUT/mfma_util train > $R/$OUT/mfma_util.log 2>&1 || true
bash profiles/tools/collect_mfma_util.sh C4 $OUT/mfma_util_c4 train > $R/$OUT/mfma_util_c4.log 2>&1 || true
fi
if [ "$PART" = "C" ]; then
b --workload C5 --mode train --steps 3 --warmup 1 > $OUT/bench_train_C5.json 2> $OUT/bench_train_C5.err
b --workload C5 --mode eval --steps 5 --warmup 1 > $OUT/bench_eval_C5.json 2> $OUT/bench_eval_C5.err
bash profiles/tools/collect_traffic.sh C5 $OUT/traffic_c5 train
fi
if [ "$PART" = "D" ]; then
b --params trained > $OUT/bench_train_C3_trained.json 2> $OUT/bench_train_C3_trained.err
b --params trained --mode eval > $OUT/bench_eval_C3_trained.json 2> $OUT/bench_eval_C3_trained.err
b --workload C4 --params trained --mode train > $OUT/bench_train_C4_trained.json 2> $OUT/bench_train_C4_trained.err
for w in C3 C4 C5; do for m in train eval; do b --workload $w --mode $m --dtype float32 --steps 5 --warmup 2 > $OUT/bench_${m}_${w}_f32.json 2> $OUT/bench_${m}_${w}_f32.err; done; done
for m in half prssm; do for w in C2 C3; do python3 bench.py --workload $w --model $m > $OUT/bench_train_${w}_${m}.json 2> $OUT/bench_train_${w}_${m}.err; done; done
python3 profiles/tools/dropin_throughput.py > $OUT/dropin_throughput.log 2>&1 || true
CBFSSM_BENCH_ONE_DEVICE=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 6 --warmup 2 > $OUT/bench_2ranks_one_device.json 2> $OUT/bench_2ranks_one_device.err || true
./cbf-ssm_amd/csrc/probe/mfma_f64_probe > $OUT/mfma_probe_f64_f32.log 2>&1 || true
fi
ls -la $R/$OUT | tail -50
