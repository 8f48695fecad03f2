#!/bin/bash
# usage (GPU box): scripts/pmc_linear.sh <tag>   -> per-kernel PMC means of the linear_fwd16 launches in scripts/bench_linear.py
TAG=$1; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
i=0
for pass in "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
            "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmc_$TAG/p$i -o p -- python $R/scripts/bench_linear.py --json > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$R/gpurun_out/pmc_$TAG/p*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'linear_fwd16' in n:
            key = re.sub(r'\(anonymous namespace\)::', '', n)[:60] + " grid=" + r.get('Grid_Size', '?')
            agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(agg.items()):
    print(k)
    for c, v in d.items():
        print(f"   {c:28s} n={len(v)} mean={sum(v)/len(v):.5g}")
PY
