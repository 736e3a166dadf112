"""Group a rocprofv3 kernel trace (x_kernel_trace.csv) by kernel name and grid: calls, total ms and us per call, per pass.
usage: python3 tools/trace_groups.py <kernel_trace.csv> [passes] [name filter]"""
import collections
import csv
import sys

fn = sys.argv[1]
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 1
flt = sys.argv[3] if len(sys.argv) > 3 else ""
agg = collections.defaultdict(lambda: [0, 0.0])
byname = collections.defaultdict(float)
for r in csv.DictReader(open(fn)):
    n = r["Kernel_Name"]
    if flt and flt not in n:
        continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    key = (n.split("(")[0][:40], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    agg[key][0] += 1
    agg[key][1] += d
    byname[n.split("(")[0][:40]] += d
tot = sum(byname.values())
for k, v in sorted(byname.items(), key=lambda kv: -kv[1])[:14]:
    print(f"{k:42s} {v / passes:9.1f} ms/pass {100 * v / tot:5.1f} %")
print(f"total {tot / passes:.1f} ms/pass")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:36]:
    print(k, v[0] // passes, f"{v[1] / passes:8.1f} ms/pass  {v[1] / v[0] * 1000:8.1f} us/call")
