cd $GRAFT_REPO_ROOT
for tag in "" _r0; do
  echo "== lib$tag"
  for inv in 20 200; do
  SELFOCC_HIP_LIB=$GRAFT_REPO_ROOT/selfocc_amd/libselfocc_hip$tag.so timeout 200 python bench.py --steps 100 --warmup 10 --inv-s $inv --no-cpu-baseline --no-hotpath --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('inv_s', d['config']['inv_s'], d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
  done
  SELFOCC_HIP_LIB=$GRAFT_REPO_ROOT/selfocc_amd/libselfocc_hip$tag.so timeout 200 python scripts/bench_hotpath_eval.py 2>&1 | tail -1
done
timeout 600 python -m pytest tests/test_render_gpu.py -x -q 2>&1 | tail -2
