cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_linear_gpu.py -x -q 2>&1 | tail -2
for p in 2 3; do echo "== percu=$p"; SELFOCC_LINEAR_PERCU=$p timeout 200 python scripts/bench_linear.py 2>&1 | grep -v amdgpu.ids | grep "^sum\|^proj"; done
