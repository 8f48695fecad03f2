cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_linear_gpu.py -x -q 2>&1 | tail -3
for p in 1 2 3; do echo "== percu=$p"; SELFOCC_LINEAR_PERCU=$p timeout 200 python scripts/bench_linear.py 2>&1 | grep -v amdgpu.ids | awk '{ if ($1=="sum" || $1=="proj") print; else printf "%s %s | ", $1, $10; } END{print ""}'; done
