cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_linear_gpu.py -x -q 2>&1 | tail -5
for s in 0 512 1024 2048; do echo slots=$s; SELFOCC_LINEAR_SLOTS=$s timeout 200 python scripts/bench_linear.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/linear_fwd_bench_$s.txt; done
