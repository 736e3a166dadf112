"""Per-kernel totals from a rocprofv3 kernel trace: python tools/trace_by_kernel.py <kernel_trace.csv> [n_slices]"""
import csv, sys, collections
tot = collections.defaultdict(lambda: [0, 0.0])
n = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k = r["Kernel_Name"].split("(")[0][:60]
        tot[k][0] += 1
        tot[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
s = sum(v[1] for v in tot.values())
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:60s} n={v[0] / n:8.0f} {v[1] / n:9.3f} ms/slice {100 * v[1] / s:5.1f}%")
print(f"total {s / n:.2f} ms/slice")
