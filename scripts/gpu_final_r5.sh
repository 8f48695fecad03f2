#!/bin/bash
# Round-5 end-of-round evidence in ONE gpurun call (about 17 GPU-minutes):
#   TA / L1 counters of the MSDA rows, FETCH / WRITE of the training backward's scatter kernels, PMC of the render kernels
#   (C = 1 / 4 / 25), kernel traces of bench / eval / occ / kitti / train, and the driver's bench command LAST so that its JSON
#   line carries the counters taken in this same call.   usage: gpu_final_r5.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-r5_c}; cd $R; mkdir -p gpurun_out
bash scripts/pmc_msda_rows.sh $T > /dev/null 2>&1 && cp gpurun_out/${T}_pmc_msda.json profiles/pmc_msda.json
bash scripts/pmc_train_bwd.sh $T > /dev/null 2>&1 && cp gpurun_out/${T}_pmc_bwd.json profiles/pmc_bwd.json
TOPN=40 bash scripts/gpu_trace.sh ${T}_kitti python scripts/bench_hotpath_kitti.py > /dev/null
bash scripts/gpu_final.sh $T
ls gpurun_out | grep "^$T" 
