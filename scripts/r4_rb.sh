#!/bin/bash
# round 4: binned render_bwd scatter: parity tests + timing sweep
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_render_bwd_gpu.py -m gpu -q -x 2>&1 | tail -15 | tee gpurun_out/r4/rb_tests.log
for nsem in 21 -1; do
  SO_NSEM=$nsem timeout 300 python scripts/time_render_bwd.py atomic binned 2>&1 | tail -1 | tee -a gpurun_out/r4/rb_time.log
done
for ch in 1024 2048 8192 16384; do
  echo "chunk $ch" | tee -a gpurun_out/r4/rb_time.log
  SELFOCC_RB_CHUNK=$ch timeout 300 python scripts/time_render_bwd.py binned 2>&1 | tail -1 | tee -a gpurun_out/r4/rb_time.log
done
for nt in 256 1024; do
  echo "threads $nt" | tee -a gpurun_out/r4/rb_time.log
  SELFOCC_RB_THREADS=$nt timeout 300 python scripts/time_render_bwd.py binned 2>&1 | tail -1 | tee -a gpurun_out/r4/rb_time.log
done
# per-kernel split of the binned path
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_rb
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rb -o p -- python $R/scripts/time_render_bwd.py binned > /dev/null 2>&1
f=$(find /tmp/prof_rb -name "*kernel_stats.csv" | head -1)
python $R/scripts/top_kernels.py $f 14 | cut -c1-200 | tee $R/gpurun_out/r4/rb_trace.txt
cd $R
timeout 600 python scripts/bench_hotpath_train.py 2>&1 | tail -1 | tee gpurun_out/r4/train.log
