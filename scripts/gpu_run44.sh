cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_field_gpu.py tests/test_golden_gpu.py tests/test_head_gpu.py -x -q 2>&1 | tail -4
timeout 200 python scripts/bench_hotpath_eval.py 2>&1 | tail -1
timeout 300 python scripts/bench_hotpath_train.py 2>&1 | tail -1
timeout 100 python scripts/prof_field.py 2>&1 | tail -3
