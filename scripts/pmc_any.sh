#!/bin/bash
# usage (GPU box): scripts/pmc_any.sh <tag> <kernel-regex> <python script + args...>   -> per-kernel PMC means
TAG=$1; KRE=$2; shift 2; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
i=0
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
            "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
            "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_TRANS_F32" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmc_$TAG/p$i -o p -- python "$@" > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$R/gpurun_out/pmc_$TAG/p*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if re.search(r"$KRE", n):
            agg[(__import__('re').search(r'(\w+_kernel\w*)', n) or [n, n])[1]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print(k)
    for c, v in d.items():
        print(f"   {c:28s} n={len(v)} mean={sum(v)/len(v):.5g}")
PY
