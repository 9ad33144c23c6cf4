#!/usr/bin/env python3
"""One BASELINE configuration traced once, then its detector image a few times (for counter runs).
Usage: detector_one.py C4|C5|C3 [reps]"""
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests"), str(ROOT / "tools")]
which = sys.argv[1] if len(sys.argv) > 1 else "C4"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
AUTO = "auto" in sys.argv
sys.argv = [sys.argv[0], "NONE"]
import torch

import optrace_amd as ot
import bench_configs as bc

name = [k for k in bc.CONFIGS if k.startswith(which)][0]
build, N = bc.CONFIGS[name]
with ot.global_options.no_warnings():
    RT = build(ot)
    RT.trace(N)
    ext = None if AUTO else ([-8., 8., -8., 8.] if which == "C4" else ([-45., 45., -45., 45.] if which == "C2" else None))
    for _ in range(reps):
        img = RT.detector_image(extent=ext, _keep_on_device=True) if ext else RT.detector_image(_keep_on_device=True)
torch.cuda.synchronize()
print(name, N, img.power())
