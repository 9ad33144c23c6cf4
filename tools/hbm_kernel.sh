#!/bin/bash
# HBM traffic (FETCH_SIZE, WRITE_SIZE: separate passes) of the kernels whose name contains <substr>, median per launch.
# Usage: hbm_kernel.sh <substr> <python script> [args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SUB=$1; shift
OUT=$R/gpurun_out/hbm_$SUB
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/$c" -- python3 "$@" > "$OUT/$c.log" 2>&1 || echo "failed: $c"
done
python3 - "$OUT" "$SUB" <<'PY'
import csv, glob, os, sys, collections
out, sub = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
med = lambda v: sorted(v)[len(v) // 2]
for name, cs in acc.items():
    # units per the guide: FETCH_SIZE / WRITE_SIZE count kilobytes; on gfx950 FETCH_SIZE under-reports by half
    f, w = med(cs.get("FETCH_SIZE", [0])), med(cs.get("WRITE_SIZE", [0]))
    print(f"{name}\n  FETCH_SIZE {f/1e6:.3f} GB raw (x2 on gfx950: {2*f/1e6:.3f} GB)   WRITE_SIZE {w/1e6:.3f} GB   launches {len(cs.get('FETCH_SIZE', []))}")
PY
