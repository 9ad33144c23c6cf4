#!/usr/bin/env python3
"""Kernel timeline around the last launch of a kernel whose name contains <pattern>, from a rocprofv3 kernel trace:
timeline.py <kernel_trace.csv> <pattern> [before] [after]  ->  start (us), duration, gap to the previous kernel's end, name."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
hits = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
i0 = hits[-1]
before = int(sys.argv[3]) if len(sys.argv) > 3 else 14
after = int(sys.argv[4]) if len(sys.argv) > 4 else 34
part = rows[max(0, i0 - before):i0 + after]
t0 = int(part[0]["Start_Timestamp"])
prev_end = None
for r in part:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    name = re.sub(r"\(.*", "", r["Kernel_Name"])[:70]
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {gap:7.1f}  {name}")
    prev_end = max(prev_end or 0, e)
