#!/usr/bin/env python3
"""Kernel time against wall time of a run traced with `rocprofv3 --kernel-trace`: per repetition window (between the markers
the traced script prints) the sum of kernel durations, the span from the first kernel's start to the last one's end, and the
largest gaps between consecutive kernels.  Usage: host_gaps.py <kernel_trace.csv> [last_n_kernels]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows)
rows = rows[-n:]
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
print(f"{len(rows)} kernels, span {(t1 - t0) / 1e6:.3f} ms, kernel time {busy / 1e6:.3f} ms")
gaps = []
end = int(rows[0]["End_Timestamp"])
for a, b in zip(rows[:-1], rows[1:]):
    end = max(end, int(a["End_Timestamp"]))
    g = int(b["Start_Timestamp"]) - end
    gaps.append((g, a["Kernel_Name"][:50], b["Kernel_Name"][:50]))
gaps.sort(reverse=True)
for g, a, b in gaps[:14]:
    print(f"  gap {g / 1e3:8.1f} us   after {a:50s} before {b}")
