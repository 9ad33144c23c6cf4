#!/bin/bash
# Latency side of a trace kernel: scalar-memory and instruction-fetch counters over tools/trace_cfg.py <config>.
#   bash tools/sq_latency.sh C4   ->  mean SMEM latency (SQ_INST_LEVEL_SMEM / SQ_INSTS_SMEM), scalar data / instruction cache hit rates
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CFG=${1:-C4}
OUT=$R/gpurun_out/sqlat_$CFG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/p$i" -- python3 "$R/tools/trace_cfg.py" $CFG 3 > "$OUT/p$i.log" 2>&1 || echo "failed: $set"
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, "*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "trace_kernel" in r["Kernel_Name"] or "trace_tail" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v[1:]) / max(len(v) - 1, 1) for k, v in acc.items()}
w = m.get("SQ_WAVES", 1)
for k in sorted(m): print(f"  {k} {m[k]:.4g}   per wave {m[k]/w:.4g}")
if "SQ_INST_LEVEL_SMEM" in m: print(f"  mean SMEM latency {m['SQ_INST_LEVEL_SMEM']/m['SQ_INSTS_SMEM']:.1f} cycles; summed per wave {m['SQ_INST_LEVEL_SMEM']/w:.0f} cycles of {m['SQ_WAVE_CYCLES']/w:.0f} wave cycles")
if "SQ_INST_LEVEL_VMEM" in m: print(f"  mean VMEM latency {m['SQ_INST_LEVEL_VMEM']/max(m['SQ_INSTS_VMEM_RD'],1):.1f} cycles (reads)")
if "SQC_DCACHE_REQ" in m: print(f"  scalar data cache hit rate {m['SQC_DCACHE_HITS']/max(m['SQC_DCACHE_REQ'],1):.4f}; instruction cache hit rate {m['SQC_ICACHE_HITS']/max(m['SQC_ICACHE_REQ'],1):.4f}")
if "SQ_IFETCH_LEVEL" in m: print(f"  mean instruction fetch latency {m['SQ_IFETCH_LEVEL']/max(m['SQ_IFETCH'],1):.1f} cycles, fetches per wave {m['SQ_IFETCH']/w:.1f}")
PY
