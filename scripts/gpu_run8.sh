cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python scripts/ab_render.py 1 2>&1 | grep "C="
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/pytest_gpu8.txt 2>&1
grep -E "parity cfg2 full|switch|passed|failed|Error|error|FAILED" gpurun_out/pytest_gpu8.txt | head -40
