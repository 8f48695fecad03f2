cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -o p -- python $R/scripts/bench_hotpath_train.py > /dev/null 2>&1
python $R/scripts/top_kernels.py $(find /tmp/pt -name "*kernel_stats.csv" | head -1) 70 | cut -c1-200 > $R/gpurun_out/train_top.txt
