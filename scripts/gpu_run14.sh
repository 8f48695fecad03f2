cd $GRAFT_REPO_ROOT
echo default; python scripts/bench_hotpath_eval.py 2>&1 | tail -1
echo rocblas; TORCH_BLAS_PREFER_HIPBLASLT=0 python scripts/bench_hotpath_eval.py 2>&1 | tail -1
echo hipblaslt; TORCH_BLAS_PREFER_HIPBLASLT=1 python scripts/bench_hotpath_eval.py 2>&1 | tail -1
echo tunableop; PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_VERBOSE=0 PYTORCH_TUNABLEOP_FILENAME=/tmp/tunable.csv timeout 600 python scripts/bench_hotpath_eval.py 2>&1 | tail -1
echo tunableop-second-run; PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=0 PYTORCH_TUNABLEOP_FILENAME=/tmp/tunable.csv timeout 600 python scripts/bench_hotpath_eval.py 2>&1 | tail -1
wc -l /tmp/tunable*.csv 2>/dev/null | tail -2
