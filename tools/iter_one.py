#!/usr/bin/env python3
"""C4 iterative_render with six detector positions and a user extent, a few times (for rocprofv3 --stats)."""
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests"), str(ROOT / "tools")]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
AUTO = "auto" in sys.argv
sys.argv = [sys.argv[0], "NONE"]
import torch

import optrace_amd as ot
import bench_configs as bc

build, N = bc.CONFIGS[[k for k in bc.CONFIGS if k.startswith("C4")][0]]
pos = [[0, 0, z] for z in (30., 32., 34., 36., 38., 39.5)]
ext = None if AUTO else [[-8., 8., -8., 8.]] * len(pos)
with ot.global_options.no_warnings():
    RT = build(ot)
    RT.iterative_render(N, pos=pos, extent=ext)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        RT.iterative_render(N, pos=pos, extent=ext)
    torch.cuda.synchronize()
print("iterative_render", (time.perf_counter() - t0) / reps * 1e3, "ms")
