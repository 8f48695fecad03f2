cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_linear_gpu.py -x -q 2>&1 | tail -15
timeout 200 python scripts/bench_linear.py 2>&1 | tail -20 | tee gpurun_out/linear_fwd_bench.txt
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 200 python scripts/bench_hotpath_eval.py 2>&1 | tail -2
