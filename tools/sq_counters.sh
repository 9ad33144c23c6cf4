#!/bin/bash
# SQ counters of the bench kernel: instruction counts per wave and how busy the vector ALU is.
# Usage (GPU box): bash tools/sq_counters.sh <out-name>   -> gpurun_out/<out-name>.txt
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
NAME=${1:-sq}
OUT=$R/gpurun_out/sq_$NAME
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"; do
  tag=$(echo $set | tr ' ' '_')
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/$tag" -- python3 "$R/bench.py" --steps 3 --warmup 2 --skip-cpu > "$OUT/$tag.log" 2>&1 || echo "failed: $set"
done
python3 - "$OUT" <<'PY' > "$R/gpurun_out/sq_$NAME.txt"
import csv, glob, os, sys
out = sys.argv[1]
acc = {}
for f in glob.glob(os.path.join(out, "*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "trace_kernel<true" in r["Kernel_Name"] or "trace_kernelILb1" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:28s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
w = sum(acc["SQ_WAVES"]) / len(acc["SQ_WAVES"])
for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_INSTS_SMEM"):
    if k in acc:
        print(f"{k} per wave: {sum(acc[k])/len(acc[k])/w:.1f}")
PY
cat "$R/gpurun_out/sq_$NAME.txt"
