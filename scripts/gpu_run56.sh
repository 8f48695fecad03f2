cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_linear_gpu.py -x -q 2>&1 | tail -3
for m in 1 0; do echo "== WGRAD_LDS=$m"; SELFOCC_WGRAD_LDS=$m timeout 200 python scripts/micro/wgrad_bench.py 2>&1 | grep -v amdgpu; done
for m in 1 0; do echo "== WGRAD_LDS=$m"; SELFOCC_WGRAD_LDS=$m timeout 300 python scripts/bench_hotpath_train.py 2>&1 | tail -1; done
