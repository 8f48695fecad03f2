cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
python -m pytest tests/test_linear_gpu.py -x -q 2>&1 | tail -2
python scripts/micro/wgrad_bench.py
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pw
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pw -o p -- python $R/scripts/micro/wgrad_bench.py > /dev/null 2>&1
python $R/scripts/top_kernels.py $(find /tmp/pw -name "*kernel_stats.csv" | head -1) 12 | cut -c1-150
