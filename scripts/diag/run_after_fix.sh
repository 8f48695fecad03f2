echo "=== two-stream check, shipped build"
DIAG_EXTRA_VICTIMS=1 DIAG_REPEAT=100 DIAG_DISTURB=bricks.linear_fwd,encoder_pass python scripts/diag/two_stream_race.py 2>&1 | grep "^disturber" | grep -v "victim torch" | cut -c1-150
echo "=== msda + encoder tests"
python -m pytest tests/test_msda_gpu.py tests/test_golden_encoder_full_gpu.py tests/test_golden_msda_gpu.py -q -x 2>&1 | tail -3
echo "=== two-rank tests"
python -m pytest tests/test_dist_gpu.py -q -x 2>&1 | tail -15
echo "=== perf"
python scripts/bench_hotpath_eval.py 2>&1 | tail -1
python scripts/bench_hotpath_occ.py 2>&1 | tail -1
python scripts/bench_hotpath_train.py 2>&1 | tail -1
python scripts/bench_msda.py 2>&1 | grep -E "head-major|plain" | cut -c1-200 | head -20
