#!/usr/bin/env python3
"""Where the host time of Raytracer.trace() goes: cProfile over repeated calls on a small bundle (C1, 1e5 rays), plus
wall time per call with and without the profiler.  Usage: python tools/profile_api.py [N] [reps]"""
import cProfile
import pathlib
import pstats
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import torch

import optrace_amd as ot
import scenes

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
for name, build in (("C1", lambda: scenes.c1_single_lens(ot)), ("C2", lambda: scenes.double_gauss(ot))):
    with ot.global_options.no_warnings():
        RT = build()
        for _ in range(50):
            RT.trace(N)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            RT.trace(N)
        t = (time.perf_counter() - t0) / reps
        print(f"{name}: trace({N}) {1e6*t:.1f} us per call (unseeded tracer, {reps} calls)")
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(reps):
            RT.trace(N)
        pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(14)
