cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_occ_gpu.py tests/test_head_gpu.py tests/test_golden_gpu.py -x -q 2>&1 | tail -3
timeout 300 python scripts/bench_hotpath_occ.py 2>&1 | tail -1
