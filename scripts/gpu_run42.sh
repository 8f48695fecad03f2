cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2_f_bench.json 2> gpurun_out/r2_f_bench.err ) 2>&1 | tail -4
tail -c 600 gpurun_out/r2_f_bench.err
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_bench
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o p -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-hotpath > $R/gpurun_out/prof_bench.log 2>&1
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1)
echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-hotpath" > $R/gpurun_out/r2_f_bench_kernel_trace.txt
python $R/scripts/top_kernels.py $f 40 >> $R/gpurun_out/r2_f_bench_kernel_trace.txt
