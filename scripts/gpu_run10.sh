cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu10.txt 2>&1
grep -E "passed|failed|Error|error|FAILED|assert" gpurun_out/pytest_gpu10.txt | head -30
python scripts/bench_hotpath_train.py 2>&1 | tail -1
