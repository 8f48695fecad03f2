import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 15
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms")
for r in rows[:n]:
    print(f'{int(r["Calls"]):6d} {float(r["TotalDurationNs"])/1e6:10.2f}ms {float(r["AverageNs"])/1e3:10.1f}us {float(r["Percentage"]):6.2f}%  {r["Name"][:120]}')
