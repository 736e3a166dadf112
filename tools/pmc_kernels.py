import csv, sys, collections
path = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
with open(path) as f:
    for row in csv.DictReader(f):
        k = row["Kernel_Name"].split("(")[0][:40]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:12]:
    wc = d.get("SQ_WAVE_CYCLES", 1.0)
    print(k, " ".join(f"{n.replace('SQ_','')}={v/wc:.3f}" if n != "SQ_WAVE_CYCLES" else f"WAVE_CYCLES={v:.3g}" for n, v in sorted(d.items())))
