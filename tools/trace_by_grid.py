"""Group a rocprofv3 kernel trace by (kernel, grid size): python tools/trace_by_grid.py <kernel_trace.csv> [substring]"""
import csv, sys, collections
rows = collections.defaultdict(list)
sub = sys.argv[2] if len(sys.argv) > 2 else ""
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        n = r["Kernel_Name"]
        if sub not in n:
            continue
        g = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
        rows[(n.split("(")[0][:40], g, int(r.get("Grid_Size_Y", 1)))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(rows.items()):
    print(f"{k[0]:40s} wgs={k[1]:6d} y={k[2]:3d} n={len(v):5d} avg={sum(v)/len(v):9.1f} us min={min(v):9.1f} total={sum(v)/1e3:8.2f} ms")
