#!/bin/bash
# Counters of the kernels whose name contains <pattern> in a python command, per launch: HBM traffic (FETCH_SIZE / WRITE_SIZE,
# separate passes), instruction mix and waiting.  Usage: kernel_counters.sh <pattern> <script> [args]  -> gpurun_out/kc_<pattern>.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
PAT=$1; shift
SCRIPT=$1; shift
case "$SCRIPT" in /*) ;; *) SCRIPT=$R/$SCRIPT;; esac
OUT=$R/gpurun_out/kc_$PAT
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/p$i" -- python3 "$SCRIPT" "$@" > "$OUT/p$i.log" 2>&1 || echo "failed: $set"
done
python3 - "$OUT" "$PAT" <<'PY' > "$R/gpurun_out/kc_$PAT.txt"
import csv, glob, os, sys, collections
out, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, "*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    big = [x for x in v if x > 0.5 * max(v)] or v
    print(f"{k:28s} launches {len(v):3d}  mean of the large launches {sum(big)/len(big):.5g}")
PY
cat "$R/gpurun_out/kc_$PAT.txt"
