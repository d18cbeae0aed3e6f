#!/bin/bash
# HBM traffic of the hot-path kernels from rocprofv3 PMC counters (MI355X_MICROARCH.md, section HBM):
# FETCH_SIZE and WRITE_SIZE need separate passes (TCC slots), both are in KiB.  Run on the GPU box from the repo root:
#   bash profiles/tools/collect_traffic.sh C3 gpurun_out/traffic
set -e
WL=${1:-C3}
OUT=${2:-gpurun_out/traffic}
MODE=${3:-train}    # the train step's forward kernels also write what the adjoint re-reads (A2 tiles, fmean/fvar)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export CBFSSM_HIP_GRAPH=0     # plain launches (the step is otherwise one graph replay)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$OUT/fetch -- python3 $R/bench.py --workload $WL --mode $MODE --steps 1 --warmup 1 --no-cpu-baseline > $R/$OUT.fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$OUT/write -- python3 $R/bench.py --workload $WL --mode $MODE --steps 1 --warmup 1 --no-cpu-baseline > $R/$OUT.write.log 2>&1
python3 $R/profiles/tools/pmc_summary.py $R/$OUT/fetch $R/$OUT/write > $R/$OUT.pmc_summary.log
python3 $R/profiles/tools/make_traffic_json.py $WL:$MODE $R/$OUT/fetch $R/$OUT/write $R/$OUT.traffic.json
