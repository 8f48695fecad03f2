cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
python -m pytest tests/test_linear_gpu.py -x -q 2>&1 | tail -2
python scripts/micro/wgrad_bench.py
for i in 1 2; do python scripts/bench_hotpath_train.py 2>&1 | tail -1; done
