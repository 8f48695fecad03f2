cd $GRAFT_REPO_ROOT
python -m pytest tests/test_linear_gpu.py tests/test_golden_gpu.py tests/test_field_gpu.py tests/test_head_gpu.py -x -q 2>&1 | tail -5
python scripts/bench_hotpath_train.py 2>&1 | tail -1
python scripts/bench_hotpath_train.py 2>&1 | tail -1
python - <<'PY'
import torch, sys
sys.path.insert(0,'.')
from selfocc_amd.linear import linear_wgrad
for T,N,K in [(66049,384,96),(78899,432,96),(7967,2304,96),(78899,96,96),(178500,288,96),(78899,96,192),(78899,192,96)]:
    dy=torch.randn(T,N,device='cuda'); x=torch.randn(T,K,device='cuda')
    for _ in range(3): linear_wgrad(dy,x)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(20): linear_wgrad(dy,x)
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)/20*1e3
    print(f"T={T} N={N} K={K}: {us:.1f} us  {4*(T*N+T*K)/us/1e3:.0f} GB/s  {2*T*N*K/us/1e6:.1f} TF/s")
PY
