cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for v in base exp1 exp2; do
  lib=$R/selfocc_amd/libselfocc_hip.so; [ $v != base ] && lib=$R/selfocc_amd/libselfocc_hip_$v.so
  rm -rf /tmp/prof_$v
  SELFOCC_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o p -- python $R/scripts/bench_hotpath_train.py > /tmp/log_$v 2>&1
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; python $R/scripts/top_kernels.py $f 60 | grep -E "total|msda|key_range"
done
