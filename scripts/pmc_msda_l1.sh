#!/bin/bash
# usage (GPU box): scripts/pmc_msda_l1.sh [tag]  -> vector-L1 (TCP) / TA counters of the MSDA forward kernels inside the eval encoder
#   (the prologue-fused and LDS-staged variants this script once compared were removed in round 5)
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-l1}; cd /tmp; export TMPDIR=/tmp; rm -rf $R/gpurun_out/pmc_$TAG
i=0
for pass in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
            "TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
            "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmc_$TAG/p$i -o p -- python $R/scripts/bench_hotpath_eval.py > /dev/null 2>&1
done
python - <<PY | tee $R/gpurun_out/${TAG}_pmc.txt
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$R/gpurun_out/pmc_$TAG/p*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        m = re.search(r'(msda_\w+_kernel<[^>]*>|render_fwd_pixgrid<[^>]*>|linear_fwd16_kernel<[^>]*>)', n)
        if m:
            agg[m.group(1)][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(agg.items()):
    print(k)
    for c, v in d.items():
        print(f"   {c:34s} n={len(v)} mean={sum(v)/len(v):.5g}")
PY
