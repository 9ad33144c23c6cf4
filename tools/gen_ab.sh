#!/bin/bash
# SQ counters of the trace kernel for the few-surface configurations (generation-bound), one after the other:
# bash tools/gen_ab.sh [C3 C4 C5 ...]  ->  gpurun_out/sq_configs.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CFGS=${@:-C3 C4 C5}
OUT=$R/gpurun_out/sq_configs.txt
: > "$OUT"
for c in $CFGS; do
  echo "## $c" >> "$OUT"
  bash "$R/tools/sq_cfg.sh" $c >> "$OUT" 2>&1
done
cat "$OUT"
