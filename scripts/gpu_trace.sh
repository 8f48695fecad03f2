#!/bin/bash
# rocprofv3 --kernel-trace --stats of one command; top kernels -> gpurun_out/<tag>_kernel_trace.txt
#   gpurun -- 'bash scripts/gpu_trace.sh r3_x_bench python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-hotpath'
R=${GRAFT_REPO_ROOT:-$(pwd)}; tag=$1; shift
mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
( cd $R && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- "$@" ) > $R/gpurun_out/prof_$tag.log 2>&1
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
out=$R/gpurun_out/${tag}_kernel_trace.txt
echo "# rocprofv3 --kernel-trace --stats -- $*" > $out
python $R/scripts/top_kernels.py $f ${TOPN:-40} >> $out
tail -2 $R/gpurun_out/prof_$tag.log >> $out
head -${TOPN:-40} $out
