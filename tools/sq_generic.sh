#!/bin/bash
# SQ instruction counts per wave for every kernel of a command.  Usage: sq_generic.sh <name> <python script> [args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
NAME=$1; shift
OUT=$R/gpurun_out/sqg_$NAME
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d "$OUT" -- python3 "$@" > "$OUT/log.txt" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(out, "*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0))[:14]:
    w = max(v.get("SQ_WAVES", 1), 1)
    print(f"{k:60s} waves {w:12.0f}  VALU/wave {v.get('SQ_INSTS_VALU',0)/w:8.1f}  SALU/wave {v.get('SQ_INSTS_SALU',0)/w:8.1f}  LDS/wave {v.get('SQ_INSTS_LDS',0)/w:7.1f}  SMEM/wave {v.get('SQ_INSTS_SMEM',0)/w:7.1f}")
PY
