#!/usr/bin/env python3
"""Times the detector stage (Raytracer.detector_image = hit search + binning, detector_spectrum) and the chunked
iterative_render on BASELINE.json configs C3-C5 at their full ray counts on one GPU.  One line per measurement."""
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests"), str(ROOT / "tools")]
import numpy as np
import torch

import optrace_amd as ot
import scenes

sys.argv = [sys.argv[0], "NONE"]  # import the scene builders of bench_configs without running its timing loop
import bench_configs as bc


def timeit(f, n=3):
    f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3


def line(name, N, ms):
    print(f"{name:58s} N={N:11,d}  {ms:8.2f} ms  {N / ms * 1e3:.3e} rays/s  {N * 56 / ms / 1e6:6.0f} GB/s algorithmic", flush=True)


with ot.global_options.no_warnings():
    RT, N = bc.c4(ot), 200_000_000
    RT.trace(N)
    for ext in ([-8, 8, -8, 8], None):
        line(f"C4 detector_image extent={'user' if ext else 'auto'}", N, timeit(lambda: RT.detector_image(extent=ext, _keep_on_device=True)))
    line("C4 detector_image extent=user, two-step chain (hits, then binning)", N,
         timeit(lambda: RT.detector_image(extent=[-8, 8, -8, 8], _unfused=True)))
    line("C4 detector_spectrum", N, timeit(lambda: RT.detector_spectrum()))
    pos = [[0, 0, z] for z in (30, 32, 34, 36, 38, 39.5)]
    for ext in ([[-8, 8, -8, 8]] * 6, None):
        RT.iterative_render(N, pos=pos, extent=ext)  # warm-up at full size: allocator pools, clocks
        torch.cuda.synchronize()
        ts = []
        for _ in range(2):
            t0 = time.perf_counter()
            RT.iterative_render(N, pos=pos, extent=ext)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        line(f"C4 iterative_render, 6 detector positions, extent={'user' if ext else 'auto'}", N, min(ts))
    del RT
    torch.cuda.empty_cache()

    RT, N = bc.c3(ot), 50_000_000
    RT.trace(N)
    line("C3 detector_image (spherical detector, Equidistant)", N, timeit(lambda: RT.detector_image(_keep_on_device=True)))
    e3 = [float(v) for v in RT.detector_image()._extent0]
    line("C3 detector_image, that extent given", N, timeit(lambda: RT.detector_image(extent=e3)))
    del RT
    torch.cuda.empty_cache()

    RT, N = scenes.hurb_slit_lens(ot, seed=51), 100_000_000
    RT.trace(N)
    line("C5 detector_image extent=auto", N, timeit(lambda: RT.detector_image(_keep_on_device=True)))
    e5 = [float(v) for v in RT.detector_image()._extent0]
    line("C5 detector_image, that extent given", N, timeit(lambda: RT.detector_image(extent=e5)))
    del RT
    torch.cuda.empty_cache()

    RT, N = scenes.double_gauss(ot, seed=1), 10_000_000
    RT.trace(N)
    line("C2 detector_image extent=user (five spots)", N, timeit(lambda: RT.detector_image(extent=[-45., 45., -45., 45.], _keep_on_device=True)))
    line("C2 detector_image extent=auto", N, timeit(lambda: RT.detector_image(_keep_on_device=True)))
