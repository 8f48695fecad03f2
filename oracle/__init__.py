"""CPU oracle of the SelfOcc hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  selfocc_amd/ never does (the product path has no CPU fallback).

Two independent restatements live here:
  * ``oracle_*.c``   — plain C, canonical float32 operation order (the checker);
  * ``torch_port.py``— the same path written with the torch ops the reference itself
    calls (F.grid_sample, softmax, cumprod ...), i.e. what the reference would run on
    CPU; used to pin the C oracle and as bench.py's ``cpu_baseline`` (kind "port").
PARITY STATUS: see the header of oracle_render.c ("parity unpinned" for the NeuS
internals that live in the absent sdfstudio fork; lookups / mappings / MSDA / losses are
pinned against torch ops and the imported reference modules in tests/golden).
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.startswith("oracle_") and f.endswith(".c")]
    hdr = os.path.join(_HERE, "..", "include", "selfocc_hip.h")
    stale = (not os.path.exists(_LIB)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs + [hdr])
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _LIB


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
    return _lib


def render_fwd(vol, rays, cfg, **kw):
    """C oracle of selfocc_render_fwd on CPU tensors; same signature/outputs as
    selfocc_amd.render.render_rays."""
    from selfocc_amd import abi
    from selfocc_amd.render import marshal_render_args
    assert not vol.sdf.is_cuda
    a, out, _keep = marshal_render_args(vol, rays, cfg, **kw)
    fn = lib().oracle_render_fwd
    fn.restype, fn.argtypes = C.c_int, [C.POINTER(abi.SoRenderArgs)]
    rc = fn(a)
    assert rc == 0, f"oracle_render_fwd rc={rc}"
    return out


def render_fwd_f64(vol, rays, cfg):
    """float64 evaluation of the canonical formulas (depth, acc) — the yardstick for the float32
    oracle's own rounding noise (oracle_render.c: one_ray_f64); SDF-only semantics."""
    import torch
    from selfocc_amd import abi
    from selfocc_amd.render import marshal_render_args
    a, _out, _keep = marshal_render_args(vol, rays, cfg)
    n = rays.n_rays
    depth, acc = torch.empty(n, dtype=torch.float64), torch.empty(n, dtype=torch.float64)
    fn = lib().oracle_render_fwd_f64
    fn.restype, fn.argtypes = C.c_int, [C.POINTER(abi.SoRenderArgs), C.c_void_p, C.c_void_p]
    rc = fn(a, depth.data_ptr(), acc.data_ptr())
    assert rc == 0, f"oracle_render_fwd_f64 rc={rc}"
    return {'depth': depth, 'acc': acc}


def meter2grid(mapping, xyz, normalize=False):
    import torch
    from selfocc_amd import abi
    m = mapping.to_abi()
    xyz = xyz.contiguous().float()
    out = torch.empty_like(xyz)
    fn = lib().oracle_meter2grid
    fn.restype = None
    fn.argtypes = [C.POINTER(abi.SoMapping), C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    fn(m, xyz.data_ptr(), xyz.shape[0], int(normalize), out.data_ptr())
    return out


def field_sdf(mapping, sdf_vol, xyz, want_grad=True):
    import torch
    from selfocc_amd import abi
    m = mapping.to_abi()
    xyz = xyz.contiguous().float()
    sdf = torch.empty(xyz.shape[0])
    grad = torch.empty(xyz.shape[0], 3) if want_grad else None
    fn = lib().oracle_field_sdf
    fn.restype = None
    fn.argtypes = [C.POINTER(abi.SoMapping), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    fn(m, sdf_vol.contiguous().data_ptr(), xyz.data_ptr(), xyz.shape[0], sdf.data_ptr(),
       None if grad is None else grad.data_ptr())
    return sdf, grad


def expf(x):
    fn = lib().oracle_expf
    fn.restype, fn.argtypes = C.c_float, [C.c_float]
    return fn(x)


def linspace01(j, n):
    fn = lib().oracle_linspace01
    fn.restype, fn.argtypes = C.c_float, [C.c_int, C.c_int]
    return fn(j, n)


def msda_fwd(value, shapes, starts, loc, attw):
    """C oracle of MSDA forward.  value (bs,nv,h,d) f32; shapes (L,2) int; loc (bs,nq,h,L,P,2)."""
    import torch
    bs, nv, heads, d = value.shape
    _, nq, _, L, P, _ = loc.shape
    out = torch.empty(bs, nq, heads * d)
    sh = shapes.to(torch.int32).contiguous(); st = starts.to(torch.int32).contiguous()
    fn = lib().oracle_msda_fwd
    fn.restype, fn.argtypes = C.c_int, [C.c_void_p] * 6 + [C.c_int] * 7
    fn(value.contiguous().data_ptr(), sh.data_ptr(), st.data_ptr(), loc.contiguous().data_ptr(),
       attw.contiguous().data_ptr(), out.data_ptr(), bs, nv, nq, heads, d, L, P)
    return out


def msda_bwd(value, shapes, starts, loc, attw, g_out):
    import torch
    bs, nv, heads, d = value.shape
    _, nq, _, L, P, _ = loc.shape
    gv, gl, ga = torch.zeros_like(value), torch.empty_like(loc), torch.empty_like(attw)
    sh = shapes.to(torch.int32).contiguous(); st = starts.to(torch.int32).contiguous()
    fn = lib().oracle_msda_bwd
    fn.restype, fn.argtypes = C.c_int, [C.c_void_p] * 9 + [C.c_int] * 7
    fn(value.contiguous().data_ptr(), sh.data_ptr(), st.data_ptr(), loc.contiguous().data_ptr(),
       attw.contiguous().data_ptr(), g_out.contiguous().data_ptr(), gv.data_ptr(), gl.data_ptr(), ga.data_ptr(),
       bs, nv, nq, heads, d, L, P)
    return gv, gl, ga
