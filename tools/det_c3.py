#!/usr/bin/env python3
"""C3 (arizona eye, 5e7 rays, spherical retina detector): detector_image per sphere projection, automatic and given extent."""
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import torch

import optrace_amd as ot
import scenes


def timeit(f, n=5):
    f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[n // 2]


N = 50_000_000
with ot.global_options.no_warnings():
    RT = scenes.c3_arizona_eye_rgb(ot)
    RT.trace(N)
    for proj in ("Equidistant", "Orthographic", "Equal-Area", "Stereographic"):
        a = timeit(lambda: RT.detector_image(projection_method=proj, _keep_on_device=True))
        e = [float(v) for v in RT.detector_image(projection_method=proj)._extent0]
        u = timeit(lambda: RT.detector_image(projection_method=proj, extent=e, _keep_on_device=True))
        old = ot.Raytracer.AUTO_ONE_PASS_FROM
        ot.Raytracer.AUTO_ONE_PASS_FROM = 1 << 60
        c = timeit(lambda: RT.detector_image(projection_method=proj, _keep_on_device=True))
        ot.Raytracer.AUTO_ONE_PASS_FROM = old
        cu = timeit(lambda: RT.detector_image(projection_method=proj, extent=e, _unfused=True, _keep_on_device=True))
        print(f"C3 detector_image {proj:13s} N={N:,d}: automatic extent {a:.2f} ms (hit-list chain {c:.2f}), extent given {u:.2f} ms "
              f"(two-step chain {cu:.2f})", flush=True)
