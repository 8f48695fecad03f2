# round-2 GPU pass 1: A/B of the fast-path switches (two library builds), the GPU test suite, the bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python scripts/ab_render.py 1 4 25 > gpurun_out/ab_default.txt 2>&1
SELFOCC_HIP_LIB=$GRAFT_REPO_ROOT/selfocc_amd/libselfocc_hip_flat.so python scripts/ab_render.py 1 4 25 > gpurun_out/ab_flat.txt 2>&1
cat gpurun_out/ab_default.txt gpurun_out/ab_flat.txt
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/pytest_gpu.txt 2>&1
grep -E "parity|switch|passed|failed|Error|error|assert|FAILED" gpurun_out/pytest_gpu.txt | head -80
timeout 600 python bench.py > gpurun_out/bench_r2a.json 2> gpurun_out/bench_r2a.err; tail -c 7000 gpurun_out/bench_r2a.json; tail -5 gpurun_out/bench_r2a.err
