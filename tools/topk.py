"""Print the top kernels of a rocprofv3 --stats kernel_stats.csv (name shortened)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 15
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:n]:
    name = r["Name"].split("(")[0][:70]
    print(f'{float(r["TotalDurationNs"]) / 1e6:9.2f} ms {100 * float(r["TotalDurationNs"]) / tot:5.1f}% n={r["Calls"]:>6} avg={float(r["AverageNs"]) / 1e3:8.1f} us  {name}')
print(f"total {tot / 1e6:.2f} ms")
