#!/bin/bash
# Round 6, review item 7 ("render backward with the forward's interpolated features kept"): what the corner gathers cost the ray
# kernel, measured with an A/B BUILD (-DSO_RB_NO_GATHER: the ray kernel gathers NO feature corners — timing only, gradients
# wrong), crossed with the round-5 record switches (SELFOCC_RB_DBG 8 / 16).  rocprofv3 kernel trace of the training iteration, average per launch.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
out=gpurun_out/r6_c_render_bwd_gather_bound.txt
echo "# rocprofv3 --kernel-trace --stats -- python scripts/bench_hotpath_all.py --only nuscenes_occ --no-eval   (7 iterations; us per launch)" > $out
# libselfocc_hip_nogather.so: scripts/build_variant.sh nogather render_bwd.hip -DSO_RB_NO_GATHER (built where hipcc is; travels with the snapshot)
for v in shipped nogather; do
  for dbg in 0 8 24; do
    export SELFOCC_RB_DBG=$dbg
    if [ $v = nogather ]; then export SELFOCC_HIP_LIB=$R/selfocc_amd/libselfocc_hip_nogather.so; else unset SELFOCC_HIP_LIB; fi
    TOPN=60 bash scripts/gpu_trace.sh rb_${v}_$dbg python scripts/bench_hotpath_all.py --only nuscenes_occ --no-eval > /dev/null 2>&1
    echo "library $v   SELFOCC_RB_DBG=$dbg   (8: the ray kernel writes 32-byte records; 24: and the brick kernel reads 32-byte records)" >> $out
    grep -E "render_bwd_kernel|rb_brick_kernel|rb_count_kernel|render_fwd_samples" gpurun_out/rb_${v}_${dbg}_kernel_trace.txt | cut -c1-110 >> $out
  done
done
unset SELFOCC_RB_DBG SELFOCC_HIP_LIB
cat $out
