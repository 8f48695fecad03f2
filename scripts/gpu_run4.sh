cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python scripts/ab_render.py 1 4 25 > gpurun_out/ab4.txt 2>&1; cat gpurun_out/ab4.txt
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/pytest_gpu4.txt 2>&1
grep -E "parity cfg2|switch|passed|failed|Error|error|FAILED" gpurun_out/pytest_gpu4.txt | head -40
python scripts/bench_msda.py > gpurun_out/msda4.txt 2>&1; grep -v "^{\"peak" gpurun_out/msda4.txt | cut -c1-260
