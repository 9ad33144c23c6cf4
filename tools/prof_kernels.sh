#!/bin/bash
# per-kernel time table of any python command: bash tools/prof_kernels.sh <name> <script> [args]  -> gpurun_out/prof_<name>.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
NAME=$1; shift
SCRIPT=$1; shift
case "$SCRIPT" in /*) ;; *) SCRIPT=$R/$SCRIPT;; esac
OUT=$R/gpurun_out/prof_$NAME
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$SCRIPT" "$@" > "$OUT/log.txt" 2>&1
python3 - "$OUT" <<'PY' > "$R/gpurun_out/prof_$NAME.txt"
import csv, glob, os, sys, re
f = max(glob.glob(os.path.join(sys.argv[1], "*/*_kernel_stats.csv")), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print(f"{'kernel':70s} {'calls':>6s} {'avg ms':>9s} {'min ms':>9s} {'total ms':>9s} {'%':>6s}")
for r in rows[:25]:
    name = re.sub(r"\(.*", "", r["Name"])[:70]
    print(f"{name:70s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e6:9.3f} {float(r['MinNs'])/1e6:9.3f} {float(r['TotalDurationNs'])/1e6:9.2f} {float(r['Percentage']):6.2f}")
PY
cat "$R/gpurun_out/prof_$NAME.txt"
