#!/bin/bash
# counter passes over tools/trace_cfg.py <config>; prints per-kernel means for the trace kernel
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CFG=${1:-C4}
OUT=$R/gpurun_out/sqcfg_$CFG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/p$i" -- python3 "$R/tools/trace_cfg.py" $CFG 3 > "$OUT/p$i.log" 2>&1 || echo "failed: $set"
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
dur = []
for f in glob.glob(os.path.join(out, "*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "trace_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(os.path.join(out, "p1/*/*_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        if "trace_kernel" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
m = {k: sum(v[1:]) / max(len(v) - 1, 1) for k, v in acc.items()}  # skip the first (cold) launch
t = sum(dur[1:]) / max(len(dur) - 1, 1)
w = m.get("SQ_WAVES", 1)
print(f"kernel {1e3*t:.3f} ms, waves {w:.0f}")
for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS"):
    if k in m: print(f"  {k}/wave {m[k]/w:.1f}")
if "GRBM_GUI_ACTIVE" in m:
    cyc = m["GRBM_GUI_ACTIVE"] / 8
    print(f"  clock {cyc/t/1e9:.3f} GHz   VALU busy {m['SQ_ACTIVE_INST_VALU']*4/1024/cyc:.3f}   wait_inst/wave_cycles {m['SQ_WAIT_INST_ANY']/m['SQ_WAVE_CYCLES']:.3f}  wait_any/wave_cycles {m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES']:.3f}  occupancy waves/SIMD {m['SQ_WAVE_CYCLES']*4/1024/cyc:.2f}")
PY
