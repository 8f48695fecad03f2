cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_layernorm_gpu.py tests/test_golden_gpu.py tests/test_msda_gpu.py tests/test_head_gpu.py -m gpu -q -x > gpurun_out/pytest_gpu11.txt 2>&1
grep -E "passed|failed|Error|error|FAILED|assert" gpurun_out/pytest_gpu11.txt | head -30
python scripts/bench_hotpath_eval.py 2>&1 | tail -1
python scripts/bench_hotpath_train.py 2>&1 | tail -1
