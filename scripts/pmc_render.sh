#!/bin/bash
# usage (GPU box): scripts/pmc_render.sh <channels> <tag> [quick]  -> PMC means of the render_fwd kernels, stdout + gpurun_out/<tag>_pmc.txt
#   env SELFOCC_HIP_LIB selects an A/B build; "quick" skips the FETCH_SIZE / WRITE_SIZE passes
C=$1; TAG=$2; Q=${3:-full}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
passes=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
        "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
        "GRBM_GUI_ACTIVE GRBM_TA_BUSY" "TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum")
[ "$Q" = full ] && passes+=("FETCH_SIZE" "WRITE_SIZE")
i=0; rm -rf $R/gpurun_out/pmc_$TAG
for pass in "${passes[@]}"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmc_$TAG/p$i -o p -- python $R/scripts/prof_render.py $C 3 > /dev/null 2>&1
done
python - <<PY | tee $R/gpurun_out/${TAG}_pmc.txt
import sys, os; sys.path.insert(0, "$R"); import bench
print("# rocprofv3 --pmc (one pass per line group), python scripts/prof_render.py $C 3, lib =", os.environ.get("SELFOCC_HIP_LIB", "default"))
print("kernel_source_sha1", bench.kernel_source_hash())
import csv, glob, collections
for f in sorted(glob.glob("$R/gpurun_out/pmc_$TAG/p*/p_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        import re
        m = re.search(r'(render_fwd_\w+<[^>]*>|sdf_brickify_kernel)', r['Kernel_Name'])
        if m:
            agg[m.group(1)][r['Counter_Name']].append(float(r['Counter_Value']))
    ta_pass = any('TA_BUSY_avr' in d for d in agg.values())
    for kn, d in agg.items():
        for k, v in d.items():
            # GRBM_GUI_ACTIVE is collected twice: the copy of the TA pass (same launches as TA_BUSY_avr) gets its own name
            name = 'GRBM_GUI_ACTIVE_ta_pass' if (ta_pass and k == 'GRBM_GUI_ACTIVE') else k
            print(f"{kn:46s} {name:28s} n={len(v)} mean={sum(v)/len(v):.5g}")
PY
