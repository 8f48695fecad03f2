#!/bin/bash
# dev: SQ / LDS / HBM counters of linear_fwd_b3_kernel at ONE shape (default 78899 x 96 -> 648), one pass per counter group
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-78899}; N=${2:-648}; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pmcl
cat > /tmp/lin_one.py <<PY
import sys, torch
sys.path.insert(0, "$R")
from selfocc_amd.linear import linear_fwd
d = torch.device("cuda:0")
T, K, N = $T, 96, $N
x = torch.randn(T, K, device=d); w = torch.randn(N, K, device=d); b = torch.randn(N, device=d); y = torch.empty(T, N, device=d)
for _ in range(6): linear_fwd(x, w, b, out=y)
torch.cuda.synchronize()
PY
i=0
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
  "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $pass --output-format csv -d /tmp/pmcl/p$i -o p -- python /tmp/lin_one.py > /tmp/pmcl_$i.log 2>&1 || tail -3 /tmp/pmcl_$i.log
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("/tmp/pmcl/p*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if 'linear_fwd_b3' in r['Kernel_Name']:
            agg[r['Grid_Size']][r['Counter_Name']].append(float(r['Counter_Value']))
for k in agg:
    print("grid", k)
    for c, v in agg[k].items(): print("   %-32s %14.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
