#!/bin/bash
# SQ / GRBM counter passes over a command; prints the per-launch means for the kernels whose name contains <substr>.
# Usage: sq_kernel.sh <substr> <python script> [args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SUB=$1; shift
OUT=$R/gpurun_out/sqk_$SUB
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/p$i" -- python3 "$@" > "$OUT/p$i.log" 2>&1 || echo "failed: $set"
done
python3 - "$OUT" "$SUB" <<'PY'
import csv, glob, os, sys, collections
out, sub = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, "*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(os.path.join(out, "p1/*/*_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            dur[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
med = lambda v: sorted(v)[len(v) // 2]
for name, cs in acc.items():
    m = {k: med(v) for k, v in cs.items()}   # the launches of the script are alike: median over them
    t = med(dur[name])
    w = m.get("SQ_WAVES", 1)
    print(f"{name}\n  kernel {1e3*t:.3f} ms (median launch), waves {w:.0f}, launches {len(dur[name])}")
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS"):
        if k in m: print(f"  {k}/wave {m[k]/w:.1f}")
    if "GRBM_GUI_ACTIVE" in m:
        cyc = m["GRBM_GUI_ACTIVE"] / 8
        print(f"  clock {cyc/t/1e9:.3f} GHz   VALU busy {m['SQ_ACTIVE_INST_VALU']*4/1024/cyc:.3f}   wait_inst/wave_cycles {m['SQ_WAIT_INST_ANY']/m['SQ_WAVE_CYCLES']:.3f}  wait_any/wave_cycles {m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES']:.3f}  occupancy waves/SIMD {m['SQ_WAVE_CYCLES']*4/1024/cyc:.2f}")
        if "SQ_ACTIVE_INST_LDS" in m:
            print(f"  LDS busy {m['SQ_ACTIVE_INST_LDS']*4/1024/cyc:.3f}  bank-conflict cycles/SIMD-cycle {m['SQ_LDS_BANK_CONFLICT']*4/1024/cyc:.3f}  VMEM busy {m['SQ_ACTIVE_INST_VMEM']*4/1024/cyc:.3f}")
PY
