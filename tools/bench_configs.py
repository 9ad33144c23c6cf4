#!/usr/bin/env python3
"""Times Raytracer.trace() (generation + all surfaces + section stores) for BASELINE.json configs C1-C5 at their full
ray counts on one GPU.  Prints one line per config: ms per trace, ray-surface-intersections/s, algorithmic GB/s."""
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import numpy as np
import torch

import optrace_amd as ot
import scenes


c3, c4 = scenes.c3_arizona_eye_rgb, scenes.c4_image_render  # (the builders live with the other scenes)


CONFIGS = {
    "C1 single lens": (lambda o: scenes.c1_single_lens(o, seed=1), 100_000),
    "C1 single lens (1e7)": (lambda o: scenes.c1_single_lens(o, seed=1), 10_000_000),
    "C2 double gauss": (lambda o: scenes.double_gauss(o, seed=1), 10_000_000),
    "C2 double gauss no_pol": (lambda o: scenes.double_gauss(o, seed=1, no_pol=True), 10_000_000),
    "C3 arizona eye RGB": (c3, 50_000_000),
    "C4 image render no_pol": (c4, 200_000_000),
    "C5 HURB slit+lens": (lambda o: scenes.hurb_slit_lens(o, seed=51), 100_000_000),
    # numeric hit search: the C2 stack with aspheric lens fronts (7 of 15 surfaces), and the mixed test scene
    "A1 double gauss aspheric fronts": (lambda o: scenes.double_gauss(o, seed=1, aspheric=True), 10_000_000),
    "A1n double gauss aspheric no_pol": (lambda o: scenes.double_gauss(o, seed=1, aspheric=True, no_pol=True), 10_000_000),
    "A2 asphere test scene": (lambda o: scenes.asphere_scene(o, seed=3), 10_000_000),
    # the same with HURB at its slit: feature level 3 (asphere search + HURB in one kernel)
    "A3 asphere test scene, HURB on": (lambda o: scenes.asphere_scene(o, seed=3, use_hurb=True), 10_000_000),
    "freeform (spline surfaces)": (lambda o: scenes.freeform_scene(o, seed=5), 10_000_000),
    "F2 freeform without the ideal lens": (lambda o: scenes.freeform_scene(o, seed=5, ideal_lens=False), 10_000_000),
}

ONLY = [a for a in sys.argv[1:] if not a.startswith("-")]  # optional name prefixes, e.g. `bench_configs.py C4`
REPS = 2 if "--short" in sys.argv else 5

for name, (build, N) in CONFIGS.items():
    if ONLY and not any(name.startswith(o) for o in ONLY):
        continue
    with ot.global_options.no_warnings():
        RT = build(ot)
        for _ in range(12 if N <= 10_000_000 else 2):  # warm-up: scene compile, allocation, and the clocks settle
            RT.trace(N)                                 # under f64 load only after ~10 launches of this size
        torch.cuda.synchronize()
        ts = []
        for _ in range(REPS if N > 1_000_000 else 300):  # short calls: many repetitions, the host side dominates
            t0 = time.perf_counter()
            RT.trace(N)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
    t = min(ts)
    t_med = sorted(ts)[len(ts) // 2]
    nt = RT.rays.Nt
    M = nt - 2
    b = N * (nt * (36 if RT.no_pol else 48) + 28)
    print(f"{name:28s} N={N:>11,d} M={M:2d}  {1e3*t:8.2f} ms  {N*M/t:9.3e} ray-surf/s  {b/t/1e9:7.0f} GB/s algorithmic"
          f"  ({b/1e9:.1f} GB)  median {1e3*t_med:.3f} ms", flush=True)
    del RT
    torch.cuda.empty_cache()
