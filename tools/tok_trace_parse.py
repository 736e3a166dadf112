import csv, glob, sys
f = glob.glob("gpurun_out/prof_tok/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "dec_tokens" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print(len(d), "dec_tokens launches; last 16 (us):", [round(x, 1) for x in d[-16:]])
