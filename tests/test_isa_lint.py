"""CPU: the built libselfocc_hip.so contains no packed-FP32 VALU instruction in a half-swapping op_sel form.

Measured on MI355X in round 5 (profiles/r5_b_packed_fp32_mfma.txt, scripts/micro/xlane_probe_lib.hip): `v_pk_mul_f32 ... op_sel:[0,1]
op_sel_hi:[1,0]` (the low result taking the HIGH half of a source) returns wrong values — sporadically, ~1e-7 of the results — while a
wave that executes v_mfma_f32_16x16x32_bf16 is resident on the same SIMD, i.e. whenever one of this library's bf16-MFMA kernels shares
the GPU with the kernel in question (second stream, second process, or two phases of one kernel).  The straight and the low-half
broadcast forms were clean over 1e11 results.  csrc/build.sh therefore disables the compiler's vectorizers (the only source of such
forms) and csrc/msda_device.h hand-writes the broadcast form where the packed rate matters; this test keeps it that way."""
import glob
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
PK = re.compile(r"\bv_pk_(mul|add|fma)_f32\b.*")
OPSEL = re.compile(r"op_sel:\[([01,]+)\]")


def test_no_half_swapping_packed_fp32_in_the_library():
    lib = os.path.join(ROOT, "selfocc_amd", "libselfocc_hip.so")
    if not os.path.exists(lib) or not os.path.exists(OBJDUMP):
        pytest.skip("library or llvm-objdump not present")
    with tempfile.TemporaryDirectory() as td:
        shutil.copy(lib, td)
        subprocess.run([OBJDUMP, "--offloading", "libselfocc_hip.so"], cwd=td, check=True, capture_output=True)
        objs = glob.glob(os.path.join(td, "*gfx950*"))
        assert objs, "no gfx950 code object in the library"
        n_pk, bad = 0, []
        for o in objs:
            out = subprocess.run([OBJDUMP, "-d", o], check=True, capture_output=True, text=True).stdout
            for line in out.splitlines():
                m = PK.search(line)
                if not m:
                    continue
                n_pk += 1
                sel = OPSEL.search(m.group(0))
                if sel and "1" in sel.group(1):           # a LOW result fed from a HIGH source half
                    bad.append(m.group(0).split("//")[0].strip())
    assert not bad, f"{len(bad)} half-swapping packed-FP32 instructions, e.g. {bad[:3]}"
    print(f"{n_pk} packed-FP32 instructions, all straight / low-half broadcast")
