#!/bin/bash
# Round-6 end-of-round evidence in ONE gpurun call (about 15 GPU-minutes):
#   the per-lease packed-FP32 survey row; TA / L1 counters of the MSDA rows (-> profiles/pmc_msda.json, source-hash keyed);
#   FETCH / WRITE of the training backward's scatter kernels (-> profiles/pmc_bwd.json, source-hash keyed); PMC of the render
#   kernels C = 1 / 4 / 25 (-> profiles/pmc_traffic.json); kernel traces of bench / every eval entry kind / the training
#   iteration; the concurrency matrix; and the driver's bench command LAST so that its JSON line carries the counters taken in
#   this same call.   usage: gpu_final_r6.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-r6_e}; cd $R; mkdir -p gpurun_out
SURVEY_SECONDS=6 python scripts/pk_swizzle_survey.py > gpurun_out/pk_swizzle_last.json 2>/dev/null
bash scripts/pmc_msda_rows.sh $T > /dev/null 2>&1 && cp gpurun_out/${T}_pmc_msda.json profiles/pmc_msda.json
bash scripts/pmc_train_bwd.sh $T > /dev/null 2>&1 && cp gpurun_out/${T}_pmc_bwd.json profiles/pmc_bwd.json
for c in 1 4 25; do bash scripts/pmc_render.sh $c ${T}_c$c full > /dev/null 2>&1; done
python scripts/pmc_traffic_update.py $T
TOPN=14 bash scripts/gpu_trace.sh ${T}_bench python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hotpath --no-extras > /dev/null
TOPN=40 bash scripts/gpu_trace.sh ${T}_eval python scripts/bench_hotpath_all.py --only nuscenes_depth --no-train --iters 10 > /dev/null
TOPN=40 bash scripts/gpu_trace.sh ${T}_occ python scripts/bench_hotpath_all.py --only nuscenes_occ --no-train --iters 10 > /dev/null
TOPN=40 bash scripts/gpu_trace.sh ${T}_kitti python scripts/bench_hotpath_all.py --only kitti_novel_depth --no-train --iters 10 > /dev/null
TOPN=64 bash scripts/gpu_trace.sh ${T}_train python scripts/bench_hotpath_all.py --only nuscenes_occ --no-eval > /dev/null
python scripts/bench_hotpath_all.py > gpurun_out/${T}_hotpath_all.json 2>/dev/null
rm -f gpurun_out/msda_pro_replay.jsonl
python scripts/diag/msda_pro_replay.py 10000 packed > /dev/null 2>&1; python scripts/diag/msda_pro_replay.py 10000 plain > /dev/null 2>&1
cp gpurun_out/msda_pro_replay.jsonl gpurun_out/${T}_msda_pro_replay.jsonl 2>/dev/null
rm -f gpurun_out/concurrency_matrix.jsonl gpurun_out/shipped_routes_parity.jsonl
python -m pytest tests/test_concurrency_gpu.py -m gpu -q 2>&1 | tail -2 > gpurun_out/${T}_concurrency.log
cp gpurun_out/concurrency_matrix.jsonl gpurun_out/${T}_concurrency_matrix.jsonl 2>/dev/null
python -m pytest tests/test_shipped_configs_gpu.py -m gpu -q 2>&1 | tail -2 > gpurun_out/${T}_shipped_configs.log
cp gpurun_out/shipped_routes_parity.jsonl gpurun_out/${T}_shipped_routes_parity.jsonl 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
cp gpurun_out/bench_detail.json gpurun_out/${T}_bench_detail.json 2>/dev/null
wc -c gpurun_out/${T}_bench.json; ls gpurun_out | grep "^$T"
