import csv, glob, sys
f = max(glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"))
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
prev_end = None
out = []
for r in rows:
    n = r["Kernel_Name"]
    if "detector_kernel" in n:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        gap = (int(r["Start_Timestamp"]) - prev_end) / 1e6 if prev_end else 0
        out.append(f"{d:.2f}(gap {gap:.1f})")
    prev_end = int(r["End_Timestamp"])
print(" ".join(out))
