#!/bin/bash
# VALU / SALU instructions per wave of the trace kernel for the generation variants of tools/trace_gen_variants.py
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-rgb_full rgb_plain gray_full rect_full rect_plain point_plain point_d65_plain point_lines_plain rect_mono_plain point_mono_iso rect_mono_conv rect_mono_full}; do
  OUT=$R/gpurun_out/sqv_$v; rm -rf "$OUT"; mkdir -p "$OUT"
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d "$OUT" -- python3 "$R/tools/trace_gen_variants.py" $v > "$OUT/log.txt" 2>&1
  python3 - "$OUT" $v <<'PY'
import csv, glob, os, sys, collections
out, v = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, "*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "trace_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = []
for f in glob.glob(os.path.join(out, "*/*_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        if "trace_kernel" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
m = {k: sorted(x)[len(x) // 2] for k, x in acc.items()}
w = m["SQ_WAVES"]
print(f"{v:12s} {sorted(dur)[len(dur)//2]:7.3f} ms  VALU/wave {m['SQ_INSTS_VALU']/w:7.1f}  SALU/wave {m['SQ_INSTS_SALU']/w:7.1f}  VMEM_RD/wave {m['SQ_INSTS_VMEM_RD']/w:6.1f}")
PY
done
