cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_msda_gpu.py tests/test_golden_gpu.py -x -q 2>&1 | tail -4
cd /tmp; export TMPDIR=/tmp
run() {  # name, env...
  v=$1; shift
  rm -rf /tmp/prof_$v
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o p -- python $R/scripts/bench_hotpath_train.py > /tmp/log_$v 2>&1
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v $@"; grep iteration_total /tmp/log_$v | cut -c1-200; python $R/scripts/top_kernels.py $f 60 | grep -E "total|band|bin_|key_range|bwd_point"
}
run list512 A=1
