"""Print the top kernels of a rocprofv3 --kernel-trace --stats kernel_stats.csv.   python tools/kernel_stats.py <csv> [steps] [top]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6 / steps:.2f} ms per step ({len(rows)} kernels)")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:top]:
    name = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")[:60]
    print(f"{name:60s} calls/step {int(r['Calls']) / steps:6.1f} avg {float(r['AverageNs']) / 1e3:9.1f} us  /step {float(r['TotalDurationNs']) / 1e6 / steps:7.3f} ms")
