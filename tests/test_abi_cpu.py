"""CPU: the C-ABI library loads (no GPU needed) and exports every symbol include/selfocc_hip.h
declares; the ctypes mirror matches the header; the product never touches the oracle."""
import ctypes as C
import os
import re
import subprocess

from selfocc_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "selfocc_hip.h")).read()


def test_every_declared_symbol_is_exported_and_mirrored():
    declared = set(re.findall(r"^\s*(?:int|size_t|const char \*)\s*\*?\s*(selfocc_\w+)\s*\(", HEADER, re.M))
    assert declared == set(abi.SYMBOLS), declared ^ set(abi.SYMBOLS)
    from selfocc_amd._lib import lib
    l = lib()                       # raises if the .so is missing or a symbol is absent
    assert l.selfocc_abi_version() == abi.ABI_VERSION
    m = re.search(r"#define SELFOCC_ABI_VERSION (\d+)", HEADER)
    assert int(m.group(1)) == abi.ABI_VERSION


def test_struct_layouts_match_header(tmp_path):
    """compile a tiny C program against the header and compare sizeof / offsetof with ctypes"""
    src = tmp_path / "sz.c"
    fields = [("so_axis", "tot_len", abi.SoAxis), ("so_mapping", "d", abi.SoMapping),
              ("so_render_args", "grad", abi.SoRenderArgs), ("so_render_args", "sdf_brick", abi.SoRenderArgs), ("so_render_args", "inv_s", abi.SoRenderArgs),
              ("so_render_args", "t_rand", abi.SoRenderArgs), ("so_render_bwd_args", "g_inv_s", abi.SoRenderBwdArgs),
              ("so_render_bwd_args", "scatter_ws_bytes", abi.SoRenderBwdArgs),
              ("so_query_args", "sem_argmax", abi.SoQueryArgs), ("so_occ_args", "sem", abi.SoOccArgs),
              ("so_occ_args", "thresh", abi.SoOccArgs), ("so_reproj_args", "wnorm", abi.SoReprojArgs),
              ("so_reproj_args", "img_h", abi.SoReprojArgs)]
    body = "\n".join(f'printf("%zu %zu\\n", sizeof({s}), offsetof({s}, {f}));' for s, f, _ in fields)
    src.write_text(f'#include <stdio.h>\n#include <stddef.h>\n#include "{ROOT}/include/selfocc_hip.h"\n'
                   f'int main(void) {{ {body} return 0; }}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", str(src), "-o", str(exe)])
    lines = subprocess.check_output([str(exe)]).decode().split("\n")
    for (s, f, cls), line in zip(fields, lines):
        size, off = map(int, line.split())
        assert C.sizeof(cls) == size, (s, C.sizeof(cls), size)
        assert getattr(cls, f).offset == off, (s, f, getattr(cls, f).offset, off)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "selfocc_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".sh")):
                txt = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M), os.path.join(dirpath, fn)
                assert 'liboracle' not in txt and 'oracle_render_fwd' not in txt, os.path.join(dirpath, fn)


def test_banded_support_query_is_pure_host_logic():
    """selfocc_msda_banded_supported / _workspace run without a GPU: decomposition decisions only."""
    from selfocc_amd._lib import lib
    l = lib()

    def sup(shapes, bs, nq, heads, d, P):
        L = len(shapes)
        arr = (C.c_int32 * (2 * L))(*[v for hw in shapes for v in hw])
        return l.selfocc_msda_banded_supported(C.cast(arr, C.c_void_p), bs, nq, heads, d, L, P)

    fpn = [(96, 200), (48, 100), (24, 50), (12, 25)]
    assert sup(fpn, 6, 22016, 6, 16, 8) == 1                      # nuscenes_occ cross-attention
    assert sup([(257, 257), (25, 257), (257, 25)], 1, 78899, 6, 16, 12) == 1   # cross-view self-attention
    assert sup([(8, 5000)], 1, 1000, 2, 16, 4) == 0               # a level wider than the LDS tile (103 KB / 128 B = 824 px)
    assert sup([(30000, 400)], 1, 10, 1, 16, 1) == 0              # 15 000 bands: scanning every band's keys costs more than atomics
    assert sup(fpn * 3, 1, 100, 2, 16, 2) == 0                    # more than 8 levels
    assert sup(fpn, 6, 22016, 6, 6, 8) < 0                        # bad channel count: argument error
    assert b"channels per head" in l.selfocc_last_error()
    n = 6 * 22016 * 6 * 4 * 8
    ws = l.selfocc_msda_bwd_banded_workspace(6, 22016, 6, 4, 8)
    assert ws >= 18 * n and ws % 16 == 0


def test_render_bwd_scatter_workspace_size_is_pure_host_logic():
    """selfocc_render_bwd_ws_bytes: one record per sample + counters / cursors / the item list of the brick-binned scatter."""
    from selfocc_amd._lib import lib
    l = lib()
    ba = abi.SoRenderBwdArgs()
    for ax, n in ((ba.fwd.map.h, 257), (ba.fwd.map.w, 257), (ba.fwd.map.d, 25)):
        ax.tot_len = n
    ba.fwd.n_rays, ba.fwd.n_samples, ba.fwd.n_rgb, ba.fwd.n_sem = 28800, 256, 3, 21
    total = 28800 * 256
    ws = l.selfocc_render_bwd_ws_bytes(ba)
    assert total * 128 <= ws <= total * 128 + (8 << 20) and ws % 256 == 0     # 24 channels: 128-byte records + counters / items
    ba.fwd.n_rgb, ba.fwd.n_sem = 0, 0
    assert total * 32 <= l.selfocc_render_bwd_ws_bytes(ba) <= total * 32 + (8 << 20)   # SDF only: 32-byte records
    ba.fwd.map.h.tot_len = 2000                                     # cell coordinates are packed in 10 bits per axis
    assert l.selfocc_render_bwd_ws_bytes(ba) == 0
