import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from selfocc_amd.linear import linear_wgrad
tot=0.0
for T,N,K,cnt in [(66049,384,96,4),(66049,192,96,4),(66049,96,96,4),(6425,768,96,8),(6425,384,96,8),(6425,96,96,8),(78899,96,96,8),(78899,48,96,4),(78899,192,96,4),(78899,96,192,4),(153000,288,96,4)]:
    dy=torch.randn(T,N,device='cuda'); x=torch.randn(T,K,device='cuda')
    for _ in range(3): linear_wgrad(dy,x)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(20): linear_wgrad(dy,x)
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)/20*1e3
    tot+=us*cnt
    print(f"T={T} N={N} K={K}: {us:.1f} us  {4*(T*N+T*K)/us/1e3:.0f} GB/s  {2*T*N*K/us/1e6:.1f} TF/s")
print(f"weighted {tot/1e3:.2f} ms  NTW={os.environ.get('SELFOCC_WGRAD_NTW')} CHUNKS={os.environ.get('SELFOCC_WGRAD_CHUNKS')}")
