#!/bin/bash
# usage (GPU box): scripts/pmc.sh <channels> <tag>   -> gpurun_out/pmc_<tag>/*.csv summary to stdout
C=$1; TAG=$2; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
i=0
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
            "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
            "GRBM_GUI_ACTIVE GRBM_TA_BUSY" "SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_LEVEL_WAVES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VALU_TRANS_F32" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmc_$TAG/p$i -o p -- python $R/scripts/prof_render.py $C 3 > /dev/null 2>&1
done
python - <<PY
import sys; sys.path.insert(0, "$R"); import bench; print("kernel_source_sha1", bench.kernel_source_hash())
import csv, glob, collections
for f in sorted(glob.glob("$R/gpurun_out/pmc_$TAG/p*/p_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'render_fwd' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        print(f"{k:28s} n={len(v)} mean={sum(v)/len(v):.5g}")
PY
