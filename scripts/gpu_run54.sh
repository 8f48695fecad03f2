cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2_i_bench.json 2> gpurun_out/r2_i_bench.err ) 2>&1 | tail -3
