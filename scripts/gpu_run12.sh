cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash scripts/gpu_prof.sh > gpurun_out/prof12.log 2>&1
timeout 900 python bench.py > gpurun_out/bench_r2c.json 2> gpurun_out/bench_r2c.err; tail -c 9000 gpurun_out/bench_r2c.json; tail -3 gpurun_out/bench_r2c.err
cat gpurun_out/r2_render_c1_pmc.txt | head -32
