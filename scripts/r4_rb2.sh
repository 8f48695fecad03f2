#!/bin/bash
# round 4: where does rb_brick_kernel's time go (dev switches)
mkdir -p gpurun_out/r4
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
timeout 600 python -m pytest $R/tests/test_render_bwd_gpu.py -m gpu -q -x 2>&1 | tail -3 | tee $R/gpurun_out/r4/rb_tests.log
: > $R/gpurun_out/r4/rb_dbg.txt
for spec in ${SPECS:-"0 4096 512" "1 4096 512" "2 4096 512" "0 4096 1024" "0 4096 256" "0 2048 512" "0 8192 512"}; do
  set -- $spec
  rm -rf /tmp/prof_rb
  SELFOCC_RB_DBG=$1 SELFOCC_RB_CHUNK=$2 SELFOCC_RB_THREADS=$3 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rb -o p -- python $R/scripts/time_render_bwd.py binned > /tmp/out.log 2>&1
  f=$(find /tmp/prof_rb -name "*kernel_stats.csv" | head -1)
  echo "== dbg=$1 chunk=$2 threads=$3  $(grep render_bwd_ms /tmp/out.log)" >> $R/gpurun_out/r4/rb_dbg.txt
  python $R/scripts/top_kernels.py $f 30 | grep -E "rb_|render_bwd" | cut -c1-150 >> $R/gpurun_out/r4/rb_dbg.txt
done
cat $R/gpurun_out/r4/rb_dbg.txt
cd $R
SO_NSEM=-1 timeout 300 python scripts/time_render_bwd.py atomic binned 2>&1 | tail -1
