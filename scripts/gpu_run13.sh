cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
export SELFOCC_BENCH_SHARE_GPU=1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 3 2>&1 | grep '^{' | python -c "
import sys, json
l = json.loads(sys.stdin.readline()); print({k: l[k] for k in ('value','n_gpus','ms_per_step','scaling')}, l['config']['sharding'], l.get('strong_scaling'))"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 20 --warmup 3 --shard rays 2>&1 | grep '^{' | python -c "
import sys, json
l = json.loads(sys.stdin.readline()); print({k: l[k] for k in ('value','n_gpus','ms_per_step','scaling')}, l['config']['sharding'], l['config']['rays_per_step_per_gpu'])"
