# end-of-round profiles: bench line (driver command) + kernel traces of the three hot-path scripts
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2_h_bench.json 2> gpurun_out/r2_h_bench.err
cd /tmp; export TMPDIR=/tmp
for tag in eval occ train; do
  case $tag in
    eval)  CMD="python $R/scripts/bench_hotpath_eval.py";;
    occ)   CMD="python $R/scripts/bench_hotpath_occ.py";;
    train) CMD="python $R/scripts/bench_hotpath_train.py";;
  esac
  rm -rf /tmp/prof_$tag
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- $CMD > $R/gpurun_out/prof_$tag.log 2>&1
  f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
  echo "# rocprofv3 --kernel-trace --stats -- $CMD" > $R/gpurun_out/r2_h_${tag}_kernel_trace.txt
  python $R/scripts/top_kernels.py $f 60 >> $R/gpurun_out/r2_h_${tag}_kernel_trace.txt
  tail -1 $R/gpurun_out/prof_$tag.log >> $R/gpurun_out/r2_h_${tag}_kernel_trace.txt
done
