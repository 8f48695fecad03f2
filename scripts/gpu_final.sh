#!/bin/bash
# End-of-round evidence in one gpurun call: PMC of the render kernels (C = 1 / 4 / 25, incl. FETCH / WRITE), kernel traces of
# the bench / depth-eval frame / occupancy frame / training iteration, the driver's bench command.   usage: gpu_final.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-r3_g}; cd $R
for c in 1 4 25; do bash scripts/pmc_render.sh $c ${T}_c$c full > /dev/null 2>&1; done
python scripts/pmc_traffic_update.py $T
TOPN=14 bash scripts/gpu_trace.sh ${T}_bench python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hotpath --no-extras > /dev/null
TOPN=40 bash scripts/gpu_trace.sh ${T}_eval python scripts/bench_hotpath_eval.py > /dev/null
TOPN=40 bash scripts/gpu_trace.sh ${T}_occ python scripts/bench_hotpath_occ.py > /dev/null
TOPN=60 bash scripts/gpu_trace.sh ${T}_train python scripts/bench_hotpath_train.py > /dev/null
bash scripts/gpu_bench.sh ${T}_bench
