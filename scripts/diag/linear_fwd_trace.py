"""Dev: per-wave timeline of linear_fwd_b3_kernel (build: scripts/build_variant.sh lintrace linear_fwd.hip -DSO_LIN_TRACE)."""
import os, sys, ctypes
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SELFOCC_HIP_LIB"] = os.path.join(R, "selfocc_amd", "libselfocc_hip_lintrace.so")
sys.path.insert(0, R)
import numpy as np, torch
from selfocc_amd.linear import linear_fwd
from selfocc_amd import abi
d = torch.device("cuda:0")
T, K, N = (int(sys.argv[1]), 96, int(sys.argv[2])) if len(sys.argv) > 2 else (78899, 96, 576)
x = torch.randn(T, K, device=d); w = torch.randn(N, K, device=d); b = torch.randn(N, device=d); y = torch.empty(T, N, device=d)
for _ in range(4): linear_fwd(x, w, b, out=y)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["SELFOCC_HIP_LIB"])
buf = np.zeros(4096 * 64, dtype=np.uint64)
lib.selfocc_diag_lin_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
rc = lib.selfocc_diag_lin_trace(buf.ctypes.data, buf.size)
assert rc == 0, rc
tr = buf.reshape(4096, 64).astype(np.int64)
live = tr[:, 0] > 0
tr = tr[live]
t0 = tr[:, 0].min()
print("waves", len(tr), "(s_memtime: shader cycles, one counter per XCD - only differences within a wave mean anything)")
start = tr[:, 0] - t0; end = tr[:, 63] - t0
q = lambda a: np.percentile(a, [0, 10, 50, 90, 100]).round(0).tolist()
print("start      ", q(start)); print("staged-start", q(tr[:, 1] - tr[:, 0])); print("end        ", q(end)); print("alive      ", q(end - start))
nt = ((tr[:, 2:62:5] > 0).sum(1)); print("tiles per wave", np.bincount(nt))
for name, a, b_ in (("load wait", 2, 3), ("split+mfma", 3, 4), ("store issue", 4, 5), ("store ack", 5, 6)):
    v = []
    for i in range(12):
        m = tr[:, 2 + 5 * i] > 0
        if m.any(): v.append((tr[m, b_ + 5 * i] - tr[m, a + 5 * i]))
    v = np.concatenate(v); print(f"{name:12s} per tile", q(v), "mean", v.mean().round(1))
m = tr[:, 7] > 0
print("tile period (top to top)", q(tr[m, 7] - tr[m, 2]))
# per-wave mean tile period by XCD (block % 8) and by column block
blk = np.nonzero(live)[0] // 4
per = (tr[:, 63] - tr[:, 1]) / np.maximum(nt, 1)
print("mean cycles per tile by XCD      ", [int(per[blk % 8 == x].mean()) for x in range(8)])
ncb = (N + 95) // 96
nb = blk.max() + 1
xcd = blk % 8; k = blk // 8; q, r = nb // 8, nb % 8
logical = xcd * q + np.minimum(xcd, r) + k
cb = logical % ncb
print("mean cycles per tile by col block", [int(per[cb == c].mean()) for c in range(ncb)])
print("alive by XCD", [int((tr[:, 63] - tr[:, 0])[blk % 8 == x].mean()) for x in range(8)], "max", [int((tr[:, 63] - tr[:, 0])[blk % 8 == x].max()) for x in range(8)])
wv = np.nonzero(live)[0] % 4
print("mean cycles per tile by wave in block", [int(per[wv == w].mean()) for w in range(4)])
# do the two blocks of a CU differ?  blocks b and b + 256? (unknown placement) — print the distribution of per instead
print("per-wave mean period percentiles", np.percentile(per, [0, 5, 25, 50, 75, 95, 100]).round(0).tolist())
