# round-2 profiles: kernel traces of bench / eval frame / training iteration + PMC passes of the C=1 render launch
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for tag in bench eval train; do
  case $tag in
    bench) CMD="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-hotpath";;
    eval)  CMD="python $R/scripts/bench_hotpath_eval.py";;
    train) CMD="python $R/scripts/bench_hotpath_train.py";;
  esac
  rm -rf /tmp/prof_$tag
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- $CMD > $R/gpurun_out/prof_$tag.log 2>&1
  f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
  echo "# rocprofv3 --kernel-trace --stats -- $CMD" > $R/gpurun_out/r2_${tag}_kernel_trace.txt
  python $R/scripts/top_kernels.py $f 40 >> $R/gpurun_out/r2_${tag}_kernel_trace.txt
  tail -2 $R/gpurun_out/prof_$tag.log >> $R/gpurun_out/r2_${tag}_kernel_trace.txt
done
cd $R
bash scripts/pmc.sh 1 r2c1 > gpurun_out/r2_render_c1_pmc.txt 2>&1
head -50 gpurun_out/r2_bench_kernel_trace.txt; cat gpurun_out/r2_render_c1_pmc.txt
