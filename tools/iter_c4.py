#!/usr/bin/env python3
"""C4 (image_render_many_rays.py geometry, 2e8 rays, six detector positions): iterative_render with render-only chunks
against every chunk through the ray storage.  Wall time of the call, median of 5 after one untimed call."""
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import torch

import optrace_amd as ot
import scenes

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000_000
if len(sys.argv) > 2:
    ot.Raytracer.ITER_GROUP = int(sys.argv[2])  # detector positions per pass
pos = scenes.C4_POSITIONS
with ot.global_options.no_warnings():
    for mode in (True, False, True):
        ot.Raytracer.ITER_RENDER_ONLY = mode
        RT = scenes.c4_image_render(ot)
        for ext in ([[-8., 8., -8., 8.]] * 6, None):
            RT.iterative_render(N, pos=pos, extent=ext)
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                imgs = RT.iterative_render(N, pos=pos, extent=ext)
                torch.cuda.synchronize()
                ts.append(1e3 * (time.perf_counter() - t0))
            ts.sort()
            print(f"C4 iterative_render, 6 positions, N={N:,d}, render_only={mode}, extent={'user' if ext else 'auto'}: "
                  f"median {ts[2]:.2f} ms  min {ts[0]:.2f} ms  power {imgs[3].power():.6f}", flush=True)
        del RT
        torch.cuda.empty_cache()
