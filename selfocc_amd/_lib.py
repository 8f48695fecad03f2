"""Loader of libselfocc_hip.so (the HIP hot path).  There is NO fallback: if the
library is missing or a symbol of include/selfocc_hip.h is absent, importing the
compute entry points raises — a silent CPU/eager path would void every parity claim.
"""
import ctypes as C
import os

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SELFOCC_HIP_LIB", os.path.join(_HERE, "libselfocc_hip.so"))  # env: kernel A/B builds
_lib = None


class SelfOccHipError(RuntimeError):
    pass


def lib():
    """Return the loaded C-ABI library (cached)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SelfOccHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `bash selfocc_amd/csrc/build.sh` (hipcc --offload-arch=gfx950). "
            f"selfocc_amd has no CPU fallback.")
    l = C.CDLL(LIB_PATH)
    for name, (res, args) in abi.SYMBOLS.items():
        try:
            fn = getattr(l, name)
        except AttributeError as e:  # pragma: no cover
            raise SelfOccHipError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    v = l.selfocc_abi_version()
    if v != abi.ABI_VERSION:
        raise SelfOccHipError(f"ABI mismatch: library {v}, python mirror {abi.ABI_VERSION}")
    _lib = l
    return l


def check(rc, what):
    if rc != 0:
        msg = lib().selfocc_last_error().decode(errors="replace")
        raise SelfOccHipError(f"{what} failed (rc={rc}): {msg}")


def ptr(t):
    """Device pointer of a tensor as c_void_p (None -> NULL)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def current_stream(device):
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def upload(arr, device, dtype=None):
    """Host array (numpy / nested list: the per-frame camera matrices of the metas) -> device tensor WITHOUT a stream
    synchronisation: staged through torch's caching pinned-memory allocator and copied with non_blocking=True (a plain
    ``tensor.to(device)`` from pageable memory blocks the host until the stream has drained — SURVEY section 8 f-4).
    CPU targets get an ordinary tensor."""
    import numpy as np
    import torch
    t = torch.as_tensor(np.asarray(arr))
    if dtype is not None:
        t = t.to(dtype)
    dev = torch.device(device)
    if dev.type != 'cuda':
        return t.to(dev)
    return t.pin_memory().to(dev, non_blocking=True)
