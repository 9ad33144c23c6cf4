#!/bin/bash
# detector_image of C3 (spherical detector) per projection, and the hit-list chain for a detector inside / behind a lens stack
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
python tools/detector_one.py C3 6 auto > /dev/null 2>&1
for i in 1 2; do
python tools/det_c3.py 2>&1 | grep -v amdgpu
python tools/det_inside_ab.py 2>&1 | grep -v amdgpu
done
