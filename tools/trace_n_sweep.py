#!/usr/bin/env python3
"""Raytracer.trace() of C4 at several ray counts: rays per second against the count (chunk sizes of iterative_render)."""
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests"), str(ROOT / "tools")]
counts = [int(float(a)) for a in sys.argv[1:]] or [200_000_000, 66_666_667, 67_108_864, 100_000_000, 50_000_000]
sys.argv = [sys.argv[0], "NONE"]
import torch

import optrace_amd as ot
import bench_configs as bc

build, _ = bc.CONFIGS[[k for k in bc.CONFIGS if k.startswith("C4")][0]]
with ot.global_options.no_warnings():
    RT = build(ot)
    for N in counts:
        for _ in range(3):
            RT.trace(N)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            RT.trace(N)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 5
        print(f"N={N:>11,d}  {1e3*t:7.3f} ms  {N/t:.3e} rays/s  ranges {len(RT.rays._ranges) if hasattr(RT.rays, '_ranges') and RT.rays._ranges is not None else '?'}")
