#!/bin/bash
# MFMA utilisation of the four time-loop kernels from rocprofv3 PMC counters (SQ block, one pass: 8 slots on gfx950;
# GRBM_GUI_ACTIVE rides in the GRBM block).  Run on the GPU box from the repo root:
#   bash profiles/tools/collect_mfma_util.sh C3 gpurun_out/mfma_util
# The program goes directly after `--` (the profiler's preloaded library has initialised the GPU before it starts).
set -e
WL=${1:-C3}
OUT=${2:-gpurun_out/mfma_util}
MODE=${3:-train}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
export CBFSSM_HIP_GRAPH=0     # plain launches (the step is otherwise one graph replay)
export CBFSSM_NO_SPLIT=1      # whole launches only, so that per-dispatch counters belong to full kernels
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE \
  --output-format csv -d $R/$OUT/pmc -- python3 $R/bench.py --workload $WL --mode $MODE --steps 1 --warmup 1 --no-cpu-baseline > $R/$OUT/pmc.log 2>&1
python3 $R/profiles/tools/mfma_util_summary.py $WL $R/$OUT/pmc > $R/$OUT/mfma_util_summary_$WL.log
cat $R/$OUT/mfma_util_summary_$WL.log
