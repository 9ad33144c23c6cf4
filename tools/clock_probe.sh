#!/bin/bash
# effective clock (GRBM_GUI_ACTIVE / 8 / kernel time) and VALU activity of the bench kernel for several library builds
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for arm in "$@"; do
  name=${arm%%=*}; lib=${arm#*=}
  OUT=$R/gpurun_out/clk_$name
  rm -rf "$OUT"; mkdir -p "$OUT"
  OPTRACE_AMD_LIB=$R/$lib rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d "$OUT" -- python3 "$R/bench.py" --steps 20 --warmup 20 --skip-cpu > "$OUT/log.txt" 2>&1
  python3 - "$OUT" "$name" <<'PY'
import csv, glob, os, sys
out, name = sys.argv[1:3]
cnt = {}
for f in glob.glob(os.path.join(out, "*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "trace_kernel<true" in r["Kernel_Name"]:
            cnt.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
dur = []
for f in glob.glob(os.path.join(out, "*/*_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        if "trace_kernel<true" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
n = 15  # the last launches: full size, settled clocks
t = sum(dur[-n:]) / n
g = sum(cnt["GRBM_GUI_ACTIVE"][-n:]) / n
va = sum(cnt["SQ_ACTIVE_INST_VALU"][-n:]) / n
wi = sum(cnt["SQ_WAIT_INST_ANY"][-n:]) / n
wc = sum(cnt["SQ_WAVE_CYCLES"][-n:]) / n
print(f"{name:10s} kernel {1e3*t:.4f} ms  clock {g/8/t/1e9:.3f} GHz  VALU busy {va*4/1024/(g/8):.3f}  wait_inst/wave_cycles {wi/wc:.3f}")
PY
done
