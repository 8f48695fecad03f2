cd $GRAFT_REPO_ROOT
for lib in libselfocc_hip.so libselfocc_hip_ho.so; do for hm in 0 1; do echo "$lib head_major=$hm"; SELFOCC_HIP_LIB=$GRAFT_REPO_ROOT/selfocc_amd/$lib SO_HEAD_MAJOR=$hm python scripts/bench_hotpath_eval.py 2>&1 | tail -1; done; done
SELFOCC_HIP_LIB=$GRAFT_REPO_ROOT/selfocc_amd/libselfocc_hip_ho.so python scripts/bench_msda.py 2>&1 | grep -v '^{"peak' | grep "_fwd" | cut -c1-230
