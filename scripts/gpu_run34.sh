cd $GRAFT_REPO_ROOT
python -m pytest tests/test_golden_gpu.py tests/test_msda_gpu.py -x -q 2>&1 | tail -3
for b in 0 1; do echo bf16=$b; SELFOCC_VALUE_BF16=$b python scripts/bench_hotpath_eval.py 2>&1 | tail -1; SELFOCC_VALUE_BF16=$b python scripts/bench_hotpath_train.py 2>&1 | tail -1; done
