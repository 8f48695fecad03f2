cd $GRAFT_REPO_ROOT
python -m pytest tests/test_golden_gpu.py tests/test_msda_gpu.py -x -q 2>&1 | tail -2
python scripts/prof_encoder_stages.py 2>&1 | tail -1
python scripts/bench_hotpath_eval.py 2>&1 | tail -1
