cd $GRAFT_REPO_ROOT
python scripts/ab_render.py 1 2>&1 | grep "C="
for w in 7 8; do SELFOCC_HIP_LIB=$GRAFT_REPO_ROOT/selfocc_amd/libselfocc_hip_w$w.so python scripts/ab_render.py 1 2>&1 | grep -E "default|inv_s_200 " ; done
