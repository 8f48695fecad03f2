"""One row per GPU lease: is the half-swapping packed-FP32 failure of DESIGN.md section 3.8 a property of gfx950 or of one chip?
Runs the isolated instruction-form victims of scripts/micro/xlane_probe_lib.hip on one HIP stream while the LIBRARY's bf16 x 3
GEMM (selfocc_linear_fwd, the strong disturber of profiles/r5_b_packed_fp32_mfma.txt) loops on a second one, and appends
    {time, host, device name / uuid / pci bus, arch, per-form wrong-result counts, results per form}
to gpurun_out/pk_swizzle_boxes/<time>_<uuid>.json (collected into profiles/r6_pk_swizzle_boxes.jsonl).  ~10 s.  The probe library is built on first use:
    hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o scripts/micro/libxlane_probe.so scripts/micro/xlane_probe_lib.hip"""
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

SO = os.path.join(ROOT, "scripts", "micro", "libxlane_probe.so")
if not os.path.exists(SO):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", SO,
                           os.path.join(ROOT, "scripts", "micro", "xlane_probe_lib.hip")])
if len(sys.argv) > 1 and sys.argv[1] == "--build-only":
    sys.exit(0)
from selfocc_amd.linear import linear_fwd

d = torch.device("cuda:0")
probe = C.CDLL(SO)
secs = float(os.environ.get("SURVEY_SECONDS", "4"))
names = ["dpp", "ds_bpermute/permute", "gather dwordx4", "exp/rcp", "v_pk_fma_f32 plain", "f32 division", "64-bit address math",
         "16 gathers in flight", "32 gathers in flight", "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0] fresh-cvt", "v_pk_mul_f32 straight fresh-cvt",
         "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0] old-regs", "v_pk_fma_f32 op_sel_hi:[0,1,1]", "v_pk_add_f32 inline const",
         "v_pk_fma_f32 op_sel_hi:[1,0,1]", "v_pk_fma_f32 op_sel_hi:[0,1,0] const", "v_pk_fma_f32 op_sel_hi:[1,0,0] const"]
table = torch.empty(1 << 22, 4, dtype=torch.int32, device=d)
probe.probe_fill_table(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(table.data_ptr()), 1 << 22)
g = torch.Generator(device=d).manual_seed(1)
x = torch.randn(78899, 96, device=d, generator=g)
w = torch.randn(432, 96, device=d, generator=g) * 0.1
b = torch.randn(432, device=d, generator=g)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
def _ms(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


# One probe_victims call launches 17 victim kernels back to back (~2 ms); the half-swapping forms are the 10th and 12th of them.
# The disturber must be RUNNING while those execute: per victim round, as many disturber launches are queued on the other stream
# as it takes to cover the whole round.  (The first version of this survey queued one 45 us disturber per 2 ms round — the
# swizzled victims never overlapped it — and reported 0 errors on 16 GPUs on which scripts/diag/two_stream_race.py, whose 150
# queued disturbers cover its first rounds, shows ~20 000: profiles/r6_pk_swizzle_boxes.jsonl keeps those rows, marked.)
with torch.no_grad():
    dist_ms = _ms(lambda: linear_fwd(x, w, b), 20)
    cnt0 = torch.zeros(24, dtype=torch.int64, device=d)
    round_ms = _ms(lambda: probe.probe_victims(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(cnt0.data_ptr()), 1,
                                               C.c_void_p(table.data_ptr()), 1 << 20), 3)
n_dist = int(round_ms / dist_ms * 1.25) + 4
row = {"disturber_ms": round(dist_ms, 4), "victim_round_ms": round(round_ms, 3), "disturber_launches_per_victim_round": n_dist}
for dist in ("none", "selfocc_linear_fwd(bf16x3)"):
    cnt = torch.zeros(24, dtype=torch.int64, device=d)
    t0, it = time.time(), 0
    with torch.no_grad():
        while time.time() - t0 < secs:
            if dist != "none":
                with torch.cuda.stream(sa):
                    for _ in range(n_dist):
                        linear_fwd(x, w, b)
            with torch.cuda.stream(sb):
                probe.probe_victims(C.c_void_p(sb.cuda_stream), C.c_void_p(cnt.data_ptr()), it * 7919, C.c_void_p(table.data_ptr()), 1 << 20)
            it += 1
            torch.cuda.synchronize()
    c = cnt[:17].tolist()
    row[dist] = dict(victim_launches=it, wrong={n: v for n, v in zip(names, c) if v}, swizzled_wrong=c[9] + c[11],
                     other_forms_wrong=sum(c) - c[9] - c[11])
p = torch.cuda.get_device_properties(0)
rec = dict(time=time.strftime("%Y-%m-%dT%H:%M:%S"), host=socket.gethostname(), device=p.name, arch=getattr(p, 'gcnArchName', ''),
           uuid=str(getattr(p, 'uuid', '')), pci=f"{getattr(p, 'pci_domain_id', 0):04x}:{getattr(p, 'pci_bus_id', 0):02x}:{getattr(p, 'pci_device_id', 0):02x}",
           cus=p.multi_processor_count, results_per_form_per_launch=4096 * 256 * 400, survey_version=2, **row)
try:
    out = subprocess.run(["rocm-smi", "--showserial", "--showuniqueid", "--showvbios", "--showdriverversion", "--showperflevel", "--json"], capture_output=True, text=True, timeout=20).stdout
    rec["rocm_smi"] = json.loads(out) if out.strip().startswith("{") else out.strip()[:300]
except Exception as e:
    rec["rocm_smi"] = repr(e)[:100]
# one FILE per lease (gpurun merges gpurun_out/ back by file name: a single appended file would be overwritten by the next lease)
os.makedirs(os.path.join(ROOT, "gpurun_out", "pk_swizzle_boxes"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "pk_swizzle_boxes", f"{time.strftime('%Y%m%dT%H%M%S')}_{rec['uuid'][:8]}.json"), "w") as f:
    f.write(json.dumps(rec) + "\n")
print(json.dumps(rec))
