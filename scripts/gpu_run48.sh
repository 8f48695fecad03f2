cd $GRAFT_REPO_ROOT
for tag in "" _x0; do
  echo "== lib$tag"
  SELFOCC_HIP_LIB=$GRAFT_REPO_ROOT/selfocc_amd/libselfocc_hip$tag.so timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-hotpath --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d.get('parity',{}).get('parity_frac_1e-4'))"
  SELFOCC_HIP_LIB=$GRAFT_REPO_ROOT/selfocc_amd/libselfocc_hip$tag.so timeout 200 python scripts/time_render.py 2>&1 | grep pixgrid
  SELFOCC_HIP_LIB=$GRAFT_REPO_ROOT/selfocc_amd/libselfocc_hip$tag.so timeout 200 python scripts/bench_hotpath_eval.py 2>&1 | tail -1
done
timeout 600 python -m pytest tests/test_render_gpu.py tests/test_head_gpu.py -x -q 2>&1 | tail -2
