#!/bin/bash
# usage (GPU box): scripts/pmc_mfma.sh <tag> <kernel-regex> <python script + args...>   -> per-kernel PMC means with the
# matrix-pipe counters (separate --pmc passes, --kernel-trace / --stats only: the guide's rule); summary -> gpurun_out/<tag>_pmc.txt
TAG=$1; KRE=$2; shift 2; R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
i=0
for pass in "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
            "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  ( cd $R && timeout 200 rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmc_$TAG/p$i -o p -- python "$@" ) > /dev/null 2>&1
done
python - <<PY | tee $R/gpurun_out/${TAG}_pmc.txt
import csv, glob, collections, re
print("# rocprofv3 --pmc <pass> -- python $*   (kernels matching /$KRE/; mean over the launches of one run)")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$R/gpurun_out/pmc_$TAG/p*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if re.search(r"$KRE", n):
            key = re.sub(r'\(anonymous namespace\)::', '', n).split('(')[0][:70] + " grid=" + r.get('Grid_Size', '?')
            agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(agg.items()):
    print(k)
    for c, v in d.items():
        print(f"   {c:30s} n={len(v)} mean={sum(v)/len(v):.5g}")
PY
