"""Per-kernel sums of the SQ counters of rocprofv3 --pmc passes (tools/sq_counters.sh): python tools/summarize_sq.py <pass dir> [<pass dir> ...]
Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_BUSY_CYCLES per SE-ish;
SQ_VALU_MFMA_BUSY_CYCLES counts cycles.  Ratios of same-unit counters are what the table is for."""
import csv, glob, os, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.Counter()
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (d, r["Dispatch_Id"])
            if key not in seen and r["Counter_Name"] in ("SQ_WAVE_CYCLES", "SQ_INSTS_LDS"):
                seen.add(key); launches[(k, d)] += 1
names = sorted({c for v in rows.values() for c in v})
for k, v in sorted(rows.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    if v.get("SQ_WAVE_CYCLES", 0) < 1e6:
        continue
    n = max([launches[(k, d)] for d in sys.argv[1:]] + [1])
    print(f"== {k[:90]}  ({n} launches)")
    wc = v.get("SQ_WAVE_CYCLES", 0.0)
    for c in names:
        if c in v:
            extra = f"  = {v[c] / wc * 100:6.2f} % of SQ_WAVE_CYCLES" if wc and c.startswith(("SQ_WAIT", "SQ_ACTIVE", "SQ_INST_CYCLES")) else ""
            print(f"   {c:34s} {v[c] / n:16.0f} per launch{extra}")
