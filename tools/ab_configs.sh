#!/bin/bash
# bench_configs.py (+ bench_detector.py with DET=1) for several library builds on one box.
# Usage: [CFGS="C3 C4"] [DET=1] ab_configs.sh name=lib.so ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for arm in "$@"; do
  name=${arm%%=*}; lib=${arm#*=}
  echo "=== $name"
  OPTRACE_AMD_LIB=$R/$lib python3 $R/tools/bench_configs.py $CFGS 2>&1 | grep -v amdgpu.ids
  if [ -n "$DET" ]; then OPTRACE_AMD_LIB=$R/$lib python3 $R/tools/bench_detector.py 2>&1 | grep -v amdgpu.ids; fi
done
