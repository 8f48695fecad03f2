"""Turn a rocprofv3 results .db (rocpd sqlite) into the text summary kept under profiles/."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
print("# rocprofv3 --kernel-trace --stats summary (durations in us)")
print(f"# source: {sys.argv[1]}   command: {' '.join(sys.argv[2:])}")
print(f"{'calls':>6} {'total_us':>14} {'avg_us':>12} {'pct':>7}  kernel")
for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print(f"{calls:6d} {total:14.3f} {avg:12.3f} {pct:7.3f}  {name}")
try:
    cols = [d[1] for d in cur.execute("pragma table_info(kernels)")]
    want = [c for c in ("name", "vgpr_count", "accum_vgpr_count", "sgpr_count", "lds_size", "scratch_size", "grid_x", "workgroup_x") if c in cols]
    if want:
        print("\n# per-kernel launch resources (first dispatch of each kernel)")
        seen = set()
        for row in cur.execute(f"select {','.join(want)} from kernels"):
            if row[0] in seen: continue
            seen.add(row[0]); print(dict(zip(want, row)))
except Exception as e:
    print("# (no per-dispatch resource table:", e, ")")
