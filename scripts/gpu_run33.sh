cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > gpurun_out/bench_r2d.json 2> gpurun_out/bench_r2d.err; tail -c 600 gpurun_out/bench_r2d.json
