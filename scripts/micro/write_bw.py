"""dev: achievable HBM write / copy bandwidth with torch's own streaming kernels (fill_, copy_, add) at the sizes the
encoder's projections write (60 - 200 MB)."""
import torch
d = torch.device("cuda:0")
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for mb in (30, 60, 100, 200, 400):
    n = mb * 1000 * 1000 // 4
    y = torch.empty(n, device=d); x = torch.randn(n, device=d); z = torch.randn(n, device=d)
    t_fill = timeit(lambda: y.fill_(1.0))
    t_copy = timeit(lambda: y.copy_(x))
    t_add = timeit(lambda: torch.add(x, z, out=y))
    print(f"{mb:4d} MB: fill {t_fill:6.1f} us = {mb / t_fill:5.2f} TB/s written | copy {t_copy:6.1f} us = {2 * mb / t_copy:5.2f} TB/s moved "
          f"| add {t_add:6.1f} us = {3 * mb / t_add:5.2f} TB/s moved")
