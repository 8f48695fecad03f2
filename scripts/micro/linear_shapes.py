"""dev: time the encoder's Linear shapes (nuscenes_depth, dim 96) through torch.addmm (hipBLASLt) against the
bytes they must move (x once + y once): how far are the vendor GEMMs from write-bound on tall-skinny f32 shapes?"""
import torch
d = torch.device("cuda:0")
shapes = [("self off", 78899, 96, 432), ("self aw", 78899, 96, 216), ("self val/out", 78899, 96, 96),
          ("hw off", 66049, 96, 384), ("hw aw", 66049, 96, 192), ("hw out", 66049, 96, 96),
          ("zh off", 7967, 96, 2304), ("zh aw", 7967, 96, 1152), ("zh out", 7967, 96, 96),
          ("cross val x3", 178500, 96, 288), ("ffn1", 78899, 96, 192), ("ffn2", 78899, 192, 96),
          ("self off+aw", 78899, 96, 648), ("hw off+aw", 66049, 96, 576), ("zh off+aw", 7967, 96, 3456)]
tot = 0.0
for name, T, K, N in shapes:
    x = torch.randn(T, K, device=d); w = torch.randn(N, K, device=d); b = torch.randn(N, device=d)
    for _ in range(5): y = torch.addmm(b, x, w.t())
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): y = torch.addmm(b, x, w.t())
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    byt = 4 * (T * K + T * N + N * K)
    print(f"{name:14s} T={T:6d} K={K:3d} N={N:4d}  {us:7.1f} us  {byt / us / 1e3:7.1f} GB/s  {2 * T * K * N / us / 1e6:6.1f} TF/s  bytes-bound@4TB/s {byt / 4e6:6.1f} us")
