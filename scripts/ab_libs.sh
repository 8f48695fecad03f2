#!/bin/bash
# A/B of library builds on the cfg2 render launch: scripts/ab_libs.sh "<channels>" lib1.so lib2.so ...   (paths relative to selfocc_amd/)
R=${GRAFT_REPO_ROOT:-$(pwd)}; ch=$1; shift
mkdir -p $R/gpurun_out
for round in 1 2; do
  for l in "$@"; do
    SELFOCC_HIP_LIB=$R/selfocc_amd/$l python $R/scripts/ab_render.py $ch 2>/dev/null | grep -E "default|no_face_safe |inv_s_200" | grep -v no_ahead
  done
done | tee $R/gpurun_out/ab_libs.txt
