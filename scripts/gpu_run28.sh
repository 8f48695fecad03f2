cd $GRAFT_REPO_ROOT
for v in _e3; do echo "lib$v"; SELFOCC_HIP_LIB=$GRAFT_REPO_ROOT/selfocc_amd/libselfocc_hip$v.so python scripts/micro/wgrad_bench.py | head -4; done
