cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do python scripts/bench_hotpath_train.py 2>&1 | tail -1; done
