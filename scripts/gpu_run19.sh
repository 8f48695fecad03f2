cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_msda_gpu.py tests/test_golden_gpu.py -x -q 2>&1 | tail -4
cd /tmp; export TMPDIR=/tmp
run() {  # name, env...
  v=$1; shift
  rm -rf /tmp/prof_$v
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o p -- python $R/scripts/bench_hotpath_train.py > /tmp/log_$v 2>&1
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v $@"; tail -1 /tmp/log_$v | cut -c1-200; python $R/scripts/top_kernels.py $f 60 | grep -E "total|band|bin_|key_range"
}
run list512 A=1
run scan SELFOCC_BAND_SCAN=1
run list1024 SELFOCC_BAND_THREADS=1024
run seg4k SELFOCC_BAND_SEG=4096
run seg16k SELFOCC_BAND_SEG=16384
