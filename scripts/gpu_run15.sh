cd $GRAFT_REPO_ROOT
echo base; python scripts/bench_hotpath_eval.py 2>&1 | tail -1
for w in 4 5 6; do echo waves=$w; SELFOCC_HIP_LIB=$GRAFT_REPO_ROOT/selfocc_amd/libselfocc_hip_mw$w.so python scripts/bench_hotpath_eval.py 2>&1 | tail -1; done
