#!/usr/bin/env python3
"""C4's source (or C3's) without any surface: what ray generation alone costs (for counter runs).
Usage: trace_gen_only.py C4|C3 [reps]"""
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests"), str(ROOT / "tools")]
which = sys.argv[1] if len(sys.argv) > 1 else "C4"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
sys.argv = [sys.argv[0], "NONE"]
import torch

import optrace_amd as ot
import bench_configs as bc

name = [k for k in bc.CONFIGS if k.startswith(which)][0]
build, N = bc.CONFIGS[name]
with ot.global_options.no_warnings():
    RT = build(ot)
    for el in list(RT.lenses) + list(RT.apertures) + list(RT.filters) + list(RT.detectors):
        RT.remove(el)
    for _ in range(reps):
        RT.trace(N)
torch.cuda.synchronize()
print(name, N, "surfaces removed; sections:", RT.rays.Nt)
