#!/bin/bash
# usage (GPU box): scripts/pmc_msda_rows.sh <tag>
# Texture-addresser utilisation of the MSDA kernels per benchmark shape (scripts/bench_msda.py --case X, one rocprofv3
# session per shape): TA_BUSY_avr, TA_TA_BUSY_sum and GRBM_GUI_ACTIVE in ONE pass (same launches), TCP accesses in a second.
#   -> gpurun_out/<tag>_pmc_msda.json (copy to profiles/pmc_msda.json; bench_msda.py attaches it to its rows) + a text summary
TAG=${1:-r5}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pmcm
for case in cross_hw cross_zh self_xview; do
  i=0
  for pass in "TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $pass --output-format csv -d /tmp/pmcm/$case/p$i -o p -- python $R/scripts/bench_msda.py --case $case > /dev/null 2>&1
  done
done
python - <<PY
import csv, glob, collections, re, json, subprocess
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n).strip()
out = {"source": "scripts/pmc_msda_rows.sh $TAG (rocprofv3 --pmc, one session per shape of scripts/bench_msda.py)", "round": "$TAG", "cases": {}}
txt = ["# rocprofv3 --pmc 'TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE' / 'TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum' -- python scripts/bench_msda.py --case <shape>",
       "# mean over the launches of one session; ta_util = TA_BUSY_avr / (GRBM_GUI_ACTIVE / 8), ta_unit_util = TA_TA_BUSY_sum / 256 / (GRBM_GUI_ACTIVE / 8)"]
for case in ("cross_hw", "cross_zh", "self_xview"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(f"/tmp/pmcm/{case}/p*/p_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if 'msda' in r['Kernel_Name']:
                agg[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    out["cases"][case] = {}
    txt.append(case)
    for k in sorted(agg):
        m = {c: sum(v) / len(v) for c, v in agg[k].items()}
        if 'GRBM_GUI_ACTIVE' not in m:
            continue
        e = dict(n=len(agg[k]['GRBM_GUI_ACTIVE']), gui_active=m['GRBM_GUI_ACTIVE'], ta_busy_avr=m.get('TA_BUSY_avr'),
                 ta_ta_busy_sum=m.get('TA_TA_BUSY_sum'), tcp_total_cache_accesses=m.get('TCP_TOTAL_CACHE_ACCESSES_sum'),
                 tcp_tcc_read_req=m.get('TCP_TCC_READ_REQ_sum'))
        out["cases"][case][k] = e
        per = e['gui_active'] / 8
        txt.append(f"   {k:58s} n={e['n']:3d} gui_active={e['gui_active']:.4g} ta_util={e['ta_busy_avr'] / per:.3f} ta_unit_util={e['ta_ta_busy_sum'] / 256 / per:.3f} tcp_accesses={e['tcp_total_cache_accesses']}")
import sys; sys.path.insert(0, "$R"); import bench
out["sources_sha1"] = bench.sources_hash(bench.MSDA_SOURCES)      # keys the record to the kernels it measured (bench_msda.py checks)
json.dump(out, open("$R/gpurun_out/${TAG}_pmc_msda.json", "w"), indent=1)
open("$R/gpurun_out/${TAG}_msda_ta_pmc.txt", "w").write("\n".join(txt) + "\n")
print("\n".join(txt))
PY
