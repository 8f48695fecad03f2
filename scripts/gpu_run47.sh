cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_linear_gpu.py tests/test_golden_gpu.py -x -q 2>&1 | tail -4
for hm in 1 0; do echo HEAD_MAJOR_PROJ=$hm; SELFOCC_HEAD_MAJOR_PROJ=$hm timeout 200 python scripts/bench_hotpath_eval.py 2>&1 | tail -1; done
