#!/bin/bash
# C4 iterative_render (tools/iter_c4.py) for several library builds on one box: bash tools/iter_ab.sh name=lib.so ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for arm in "$@"; do
  name=${arm%%=*}; lib=${arm#*=}
  echo "=== $name"
  OPTRACE_AMD_LIB=$R/$lib python3 $R/tools/iter_c4.py 2>&1 | grep -v amdgpu.ids | grep "render_only=True"
done
