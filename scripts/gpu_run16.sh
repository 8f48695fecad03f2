cd $GRAFT_REPO_ROOT
python -m pytest tests/test_msda_gpu.py tests/test_golden_gpu.py -x -q 2>&1 | tail -5
for hm in 0 1; do echo head_major=$hm; SO_HEAD_MAJOR=$hm python scripts/bench_hotpath_eval.py 2>&1 | tail -1; SO_HEAD_MAJOR=$hm python scripts/bench_hotpath_train.py 2>&1 | tail -1; done
python scripts/bench_msda.py 2>&1 | tail -3
