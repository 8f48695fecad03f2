cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_golden_gpu.py tests/test_linear_gpu.py tests/test_msda_gpu.py tests/test_dist_gpu.py -x -q 2>&1 | tail -3
for t in 1 0; do echo HEAD_MAJOR_PROJ_TRAIN=$t; SELFOCC_HEAD_MAJOR_PROJ_TRAIN=$t timeout 300 python scripts/bench_hotpath_train.py 2>&1 | tail -1; done
