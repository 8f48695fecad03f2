cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_occ
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_occ -o p -- python $R/scripts/bench_hotpath_occ.py > $R/gpurun_out/prof_occ.log 2>&1
f=$(find /tmp/prof_occ -name "*kernel_stats.csv" | head -1)
echo "# rocprofv3 --kernel-trace --stats -- python scripts/bench_hotpath_occ.py" > $R/gpurun_out/r2_g_occ_kernel_trace.txt
python $R/scripts/top_kernels.py $f 45 >> $R/gpurun_out/r2_g_occ_kernel_trace.txt
tail -1 $R/gpurun_out/prof_occ.log >> $R/gpurun_out/r2_g_occ_kernel_trace.txt
