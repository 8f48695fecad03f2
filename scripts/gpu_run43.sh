cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_linear_gpu.py -x -q 2>&1 | tail -3
for pipe in 1 0; do for p in 1 2 3; do echo "== pipe=$pipe percu=$p"; SELFOCC_LINEAR_PIPE=$pipe SELFOCC_LINEAR_PERCU=$p timeout 200 python scripts/bench_linear.py 2>&1 | grep -v amdgpu.ids | grep "^sum\|^proj"; done; done
SELFOCC_LINEAR_PIPE=1 timeout 200 python scripts/bench_linear.py 2>&1 | grep -v amdgpu.ids | cut -c1-110
