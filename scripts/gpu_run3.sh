cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python scripts/ab_render.py 1 > gpurun_out/ab3.txt 2>&1
for w in 5 7 8; do SELFOCC_HIP_LIB=$GRAFT_REPO_ROOT/selfocc_amd/libselfocc_hip_w$w.so python scripts/ab_render.py 1 2>&1 | grep skip3 >> gpurun_out/ab3.txt; done
cat gpurun_out/ab3.txt
