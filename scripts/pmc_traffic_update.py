"""profiles/pmc_traffic.json from the PMC text summaries scripts/pmc_render.sh wrote (one per volume width):
    python scripts/pmc_traffic_update.py r3_g          # reads gpurun_out/r3_g_c{1,4,25}_pmc.txt
`traffic_bytes` follows /opt/skills/guides/MI355X_MICROARCH.md (HBM): FETCH_SIZE and WRITE_SIZE from separate --pmc
passes, in KB; on gfx950 FETCH_SIZE tallies the 128-byte requests of 16-byte-per-lane loads at 64 bytes, so it is doubled
(the march reads 16-byte records / feature quarters per lane); WRITE_SIZE is taken as reported (uncalibrated)."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
tag = sys.argv[1]
out = {}
for c, key in ((1, "c1_f32_fast"), (4, "c4_f32_fast"), (25, "c25_f32_fast")):
    f = os.path.join(ROOT, "gpurun_out", f"{tag}_c{c}_pmc.txt")
    if not os.path.exists(f):
        continue
    vals, sha = {}, None
    for line in open(f):
        if line.startswith("kernel_source_sha1"):
            sha = line.split()[1]
        m = re.match(r"\s*(\S.*?)\s+(\w+)\s+n=(\d+) mean=(\S+)", line)
        if m and "render_fwd" in m.group(1):
            vals[m.group(2)] = float(m.group(4))
            kern = m.group(1)
    assert sha == bench.kernel_source_hash(), (sha, bench.kernel_source_hash())
    fetch, write = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
    out[key] = {"kernel_source_sha1": sha, "traffic_bytes": int((2 * fetch + write) * 1024), "fetch_kb": round(fetch),
                "write_kb": round(write), "traffic_bytes_uncorrected": int((fetch + write) * 1024),
                "valu_wave_insts": vals.get("SQ_INSTS_VALU"),
                "gui_active_cycles": vals.get("GRBM_GUI_ACTIVE"), "ta_busy_cycles": vals.get("GRBM_TA_BUSY"),
                # per-unit view (one pass: TA_BUSY_avr, TA_TA_BUSY_sum and the GRBM_GUI_ACTIVE of the same launches)
                "ta_busy_avr": vals.get("TA_BUSY_avr"), "ta_ta_busy_sum": vals.get("TA_TA_BUSY_sum"),
                "gui_active_cycles_ta_pass": vals.get("GRBM_GUI_ACTIVE_ta_pass"),
                "tcp_total_cache_accesses": vals.get("TCP_TOTAL_CACHE_ACCESSES_sum"), "tcp_tcc_read_req": vals.get("TCP_TCC_READ_REQ_sum"),
                "round": tag,
                "note": (f"{kern.strip()} only: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes "
                         f"(scripts/pmc_render.sh {c} {tag}_c{c}), mean of the launches of one run, MI355X, session {tag} "
                         f"(profiles/{tag}_render_c{c}_pmc.txt). traffic_bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB: the guide's gfx950 "
                         "correction for 16-byte-per-lane loads (128-byte requests tallied at 64 bytes); WRITE_SIZE as reported. "
                         "The sdf_brickify_kernel that precedes the march is a separate kernel (its counters are in the same file).")}
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
# gpurun only merges gpurun_out/ back: a copy to carry into profiles/ by hand
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"{tag}_pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: (v["traffic_bytes"], v["fetch_kb"], v["write_kb"]) for k, v in out.items()}))
