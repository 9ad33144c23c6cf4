#!/usr/bin/env python3
"""iterative_render end to end on the BASELINE scenes, render-only chunks against every chunk through the ray storage:
C2 (PSF-like image of five spots, 15 surfaces, polarisation), C3 (spherical detector with projection), C5 (HURB, feature level
1), C4 with ONE position, the asphere scene without / with HURB (levels 2 / 3), the spline-surface scene (level 5).  Median wall time of 5 calls after one untimed call."""
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import torch

import optrace_amd as ot
import scenes

CASES = [("C2 double Gauss (PSF-like image)", lambda: scenes.double_gauss(ot, seed=1), 20_000_000, [-45., 45., -45., 45.]),
         ("C2 double Gauss, 1e8 rays", lambda: scenes.double_gauss(ot, seed=1), 100_000_000, [-45., 45., -45., 45.]),
         ("C3 arizona eye (spherical detector, Equidistant)", lambda: scenes.c3_arizona_eye_rgb(ot), 50_000_000, None),
         ("C4 image render, one position", lambda: scenes.c4_image_render(ot), 200_000_000, [-8., 8., -8., 8.]),
         ("C5 HURB slit + lens", lambda: scenes.hurb_slit_lens(ot, seed=51), 100_000_000, None),
         ("A2 asphere test scene (feature level 2)", lambda: scenes.asphere_scene(ot, seed=3), 50_000_000, None),
         ("A3 asphere test scene, HURB (level 3)", lambda: scenes.asphere_scene(ot, seed=3, use_hurb=True), 50_000_000, None),
         ("freeform, spline surfaces (level 5)", lambda: scenes.freeform_scene(ot, seed=5), 50_000_000, None)]
with ot.global_options.no_warnings():
    for name, build, N, ext in CASES:
        for mode in (True, False):
            ot.Raytracer.ITER_RENDER_ONLY = mode
            RT = build()
            di = max(range(len(RT.detectors)), key=lambda i: RT.detectors[i].pos[2])  # the one behind the whole stack
            RT.iterative_render(N, extent=ext, detector_index=di)
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                img = RT.iterative_render(N, extent=ext, detector_index=di)[0]
                torch.cuda.synchronize()
                ts.append(1e3 * (time.perf_counter() - t0))
            ts.sort()
            print(f"{name:52s} N={N:>11,d} render_only={mode!s:5s} extent={'user' if ext else 'auto'}: median {ts[2]:7.2f} ms  "
                  f"min {ts[0]:7.2f} ms  {N / ts[2] * 1e3:.3e} rays/s  power {img.power():.6f}", flush=True)
            del RT
            torch.cuda.empty_cache()
