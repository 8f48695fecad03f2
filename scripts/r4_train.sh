#!/bin/bash
# round 4: training-iteration stage times + kernel trace
mkdir -p gpurun_out/r4
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
timeout 600 python $R/scripts/bench_hotpath_train.py 2>&1 | tail -1 | tee $R/gpurun_out/r4/train.log
rm -rf /tmp/prof_train
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -o p -- python $R/scripts/bench_hotpath_train.py > /tmp/out.log 2>&1
f=$(find /tmp/prof_train -name "*kernel_stats.csv" | head -1)
echo "# rocprofv3 --kernel-trace --stats -- python scripts/bench_hotpath_train.py" > $R/gpurun_out/r4/train_kernel_trace.txt
python $R/scripts/top_kernels.py $f 60 | cut -c1-220 >> $R/gpurun_out/r4/train_kernel_trace.txt
head -30 $R/gpurun_out/r4/train_kernel_trace.txt
