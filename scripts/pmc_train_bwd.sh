#!/bin/bash
# usage (GPU box): scripts/pmc_train_bwd.sh <tag>
# HBM-side bytes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) + a kernel trace of the training
# iteration's backward scatter kernels -> gpurun_out/<tag>_train_bwd_pmc.txt and gpurun_out/<tag>_pmc_bwd.json
# (copy to profiles/<tag>_train_bwd_pmc.txt and profiles/pmc_bwd.json; bench.py reports the json as roofline_bwd).
TAG=${1:-r4}; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
KRE='band_list|render_bwd_kernel|rb_brick|rb_count|bwd_point_kernel|msda_bin_kernel|field_volume_bwd'
rm -rf /tmp/pmcb; i=0
for pass in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $pass --output-format csv -d /tmp/pmcb/p$i -o p -- python $R/scripts/bench_hotpath_all.py --only nuscenes_occ --no-eval > /dev/null 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pmcb/trace -o p -- python $R/scripts/bench_hotpath_all.py --only nuscenes_occ --no-eval > /dev/null 2>&1
python - <<PY
import csv, glob, collections, re, json
KRE = re.compile(r"$KRE")
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n).strip()
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("/tmp/pmcb/p*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if KRE.search(r['Kernel_Name']):
            agg[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
dur = {}
for f in glob.glob("/tmp/pmcb/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if KRE.search(r['Name']):
            dur[short(r['Name'])] = (int(r['Calls']), float(r['AverageNs']) / 1e3)
out = {}
with open("$R/gpurun_out/${TAG}_train_bwd_pmc.txt", "w") as fo:
    fo.write("# rocprofv3 --pmc <pass> -- python scripts/bench_hotpath_train.py   (kernels matching /$KRE/; mean over the launches of one run; FETCH_SIZE / WRITE_SIZE in KB)\n")
    for k in sorted(agg):
        c, us = dur.get(k, (0, 0.0))
        fo.write(f"{k}   calls={c} avg_us={us:.1f}\n")
        for cn, v in sorted(agg[k].items()):
            fo.write(f"   {cn:28s} n={len(v)} mean={sum(v)/len(v):.5g}\n")
        g = lambda cn: (sum(agg[k][cn]) / len(agg[k][cn])) if agg[k].get(cn) else None
        out[k] = dict(calls=c, avg_us=round(us, 1), fetch_kb=g('FETCH_SIZE'), write_kb=g('WRITE_SIZE'))
out["_round"] = "$TAG"
import sys; sys.path.insert(0, "$R"); import bench
out["_sources_sha1"] = bench.sources_hash(bench.BWD_SOURCES)      # keys the record to the kernels it measured (bench.py: roofline_bwd.sources_match)
json.dump(out, open("$R/gpurun_out/${TAG}_pmc_bwd.json", "w"), indent=1)
print(open("$R/gpurun_out/${TAG}_train_bwd_pmc.txt").read())
PY
