cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_linear_gpu.py tests/test_golden_gpu.py tests/test_head_gpu.py -x -q 2>&1 | tail -8
timeout 200 python scripts/bench_linear.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/linear_fwd_bench_final.txt
for f in 1 0; do echo FUSED=$f; SELFOCC_FUSED_LINEAR=$f timeout 200 python scripts/bench_hotpath_eval.py 2>&1 | tail -1; SELFOCC_FUSED_LINEAR=$f timeout 300 python scripts/bench_hotpath_train.py 2>&1 | tail -1; done
