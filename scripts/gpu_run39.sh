# round-2 final profiles: kernel traces of eval frame / training iteration
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for tag in eval train; do
  case $tag in
    eval)  CMD="python $R/scripts/bench_hotpath_eval.py";;
    train) CMD="python $R/scripts/bench_hotpath_train.py";;
  esac
  rm -rf /tmp/prof_$tag
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- $CMD > $R/gpurun_out/prof_$tag.log 2>&1
  f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
  echo "# rocprofv3 --kernel-trace --stats -- $CMD" > $R/gpurun_out/r2_f_${tag}_kernel_trace.txt
  python $R/scripts/top_kernels.py $f 60 >> $R/gpurun_out/r2_f_${tag}_kernel_trace.txt
  tail -2 $R/gpurun_out/prof_$tag.log >> $R/gpurun_out/r2_f_${tag}_kernel_trace.txt
done
