cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python scripts/ab_render.py 1 > gpurun_out/ab6.txt 2>&1; cat gpurun_out/ab6.txt
timeout 900 python -m pytest tests/test_render_gpu.py tests/test_head_gpu.py -m gpu -q -s > gpurun_out/pytest_gpu6.txt 2>&1
grep -E "parity cfg2 full|switch|passed|failed|Error|error|FAILED" gpurun_out/pytest_gpu6.txt | head -40
