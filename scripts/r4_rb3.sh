#!/bin/bash
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/r4
timeout 600 python -m pytest $R/tests/test_render_bwd_gpu.py -m gpu -q -x 2>&1 | tail -2
: > $R/gpurun_out/r4/rb_ab.txt
for spec in "default 0" "default 4" "rbw3 0" "rbw4 0"; do
  set -- $spec
  lib=$R/selfocc_amd/libselfocc_hip.so; [ $1 != default ] && lib=$R/selfocc_amd/libselfocc_hip_$1.so
  rm -rf /tmp/prof_rb
  SELFOCC_HIP_LIB=$lib SELFOCC_RB_DBG=$2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rb -o p -- python $R/scripts/time_render_bwd.py binned > /tmp/out.log 2>&1
  f=$(find /tmp/prof_rb -name "*kernel_stats.csv" | head -1)
  echo "== lib=$1 dbg=$2  $(grep render_bwd_ms /tmp/out.log)" >> $R/gpurun_out/r4/rb_ab.txt
  python $R/scripts/top_kernels.py $f 30 | grep -E "rb_|render_bwd" | cut -c1-150 >> $R/gpurun_out/r4/rb_ab.txt
done
cat $R/gpurun_out/r4/rb_ab.txt
