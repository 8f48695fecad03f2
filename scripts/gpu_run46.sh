cd $GRAFT_REPO_ROOT
for tag in "" _b2w4 _b2w3 _b4w3 _b1w3; do
  echo "== lib$tag"
  SELFOCC_HIP_LIB=$GRAFT_REPO_ROOT/selfocc_amd/libselfocc_hip$tag.so timeout 200 python scripts/bench_hotpath_eval.py 2>&1 | tail -1
done
SELFOCC_HIP_LIB=$GRAFT_REPO_ROOT/selfocc_amd/libselfocc_hip_b2w3.so timeout 300 python -m pytest tests/test_msda_gpu.py tests/test_golden_gpu.py -x -q 2>&1 | tail -2
