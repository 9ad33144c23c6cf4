#!/usr/bin/env python3
"""Hit-list chain of `detector_image` (automatic extent) for a detector INSIDE the lens stack -- the case in which the pair of
kernels of `ot_detector_hits` (detector_last_kernel + detector_rest_kernel) finds every ray left over -- and for the detector
behind the stack."""
import os
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import torch

import optrace_amd as ot
import scenes


def timeit(f, n=7):
    f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[n // 2]


N = 20_000_000
with ot.global_options.no_warnings():
    RT = scenes.double_gauss(ot, seed=1)
    RT.trace(N)
    ot.Raytracer.AUTO_ONE_PASS_FROM = 1 << 60
    z_end = RT.detectors[0].pos[2]
    z_mid = 0.5 * (RT.lenses[2].pos[2] + RT.lenses[3].pos[2])
    for name, z in (("behind the stack", z_end), ("inside the stack", z_mid)):
        RT.detectors[0].move_to([0, 0, z])
        t = timeit(lambda: RT.detector_image(_keep_on_device=True))
        img = RT.detector_image()
        print(f"detector {name} (z = {z:.2f}): "
              f"{t:.3f} ms, power {img.power():.9f}", flush=True)
