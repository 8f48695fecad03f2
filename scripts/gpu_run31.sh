cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pe
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe -o p -- python $R/scripts/bench_hotpath_eval.py > /dev/null 2>&1
python $R/scripts/top_kernels.py $(find /tmp/pe -name "*kernel_stats.csv" | head -1) 40 | cut -c1-170
