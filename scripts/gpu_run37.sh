cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_linear_gpu.py -x -q 2>&1 | tail -5
run() { echo "== $*"; env "$@" timeout 200 python scripts/bench_linear.py 2>&1 | grep -v amdgpu.ids | awk '{ if ($1=="sum" || $1=="proj") print; else printf "%s %s %s | ", $1, $2, $10; } END{print ""}'; }
run SELFOCC_LINEAR_TILE=32 SELFOCC_LINEAR_SLOTS=512
run SELFOCC_LINEAR_TILE=16 SELFOCC_LINEAR_WAVES=4
run SELFOCC_LINEAR_TILE=16 SELFOCC_LINEAR_WAVES=8
run SELFOCC_LINEAR_TILE=16 SELFOCC_LINEAR_WAVES=4 SELFOCC_LINEAR_PERCU=3
run SELFOCC_LINEAR_TILE=16 SELFOCC_LINEAR_WAVES=4 SELFOCC_LINEAR_PERCU=2
run SELFOCC_LINEAR_TILE=16 SELFOCC_LINEAR_WAVES=8 SELFOCC_LINEAR_PERCU=1
run SELFOCC_LINEAR_TILE=16 SELFOCC_LINEAR_WAVES=4 SELFOCC_LINEAR_PERCU=8
