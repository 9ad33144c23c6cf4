#!/bin/bash
# One rank's share of config 4 at eight ranks (2.5e7 rays, six positions): kernel time against wall time, largest gaps.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_c4shard
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 "$R/tools/bench_c4_sharded.py" --steps 3 > "$OUT/log.txt" 2>&1
grep "^{" "$OUT/log.txt"
F=$(ls "$OUT"/*/*_kernel_trace.csv | head -1)
python3 "$R/tools/timeline.py" "$F" trace_tail_kernel 14 34
