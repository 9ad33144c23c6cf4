#!/usr/bin/env python3
"""Kernel time of Raytracer.trace on scenes with an early stop: the double Gauss (23 % of the rays end at its stop,
8 of 15 surfaces behind it) and a 30-surface relay whose stop takes about half of the rays before 28 surfaces.
Usage: OPTRACE_AMD_LIB=<lib> python tools/bench_compaction.py"""
import ctypes as C
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import numpy as np
import torch

import optrace_amd as ot
from optrace_amd import _capi
import scenes


def relay30(no_pol=False):
    RT = ot.Raytracer(outline=[-6, 6, -6, 6, -12, 80], no_pol=no_pol, seed=5)
    RT.add(ot.RaySource(ot.CircularSurface(r=2.0), pos=[0, 0, -10], divergence="Isotropic", div_angle=2.0,
                        spectrum=ot.LightSpectrum("Lines", lines=[486.1327, 589.2938, 656.272], line_vals=[1, 1, 1])))
    n = ot.RefractionIndex("Abbe", n=1.55, V=55)
    RT.add(ot.Lens(ot.SphericalSurface(r=4, R=60), ot.SphericalSurface(r=4, R=-60), n=n, pos=[0, 0, -6], d=1.0))
    RT.add(ot.Aperture(ot.RingSurface(r=4, ri=1.5), pos=[0, 0, -3]))  # the stop: about half of the beam passes
    for k in range(14):
        R = 90.0 if k % 2 == 0 else -90.0
        RT.add(ot.Lens(ot.SphericalSurface(r=4, R=R), ot.SphericalSurface(r=4, R=-R), n=n, pos=[0, 0, 5.0 * k], d=0.8))
    return RT


lib = _capi.load_library()
ms = C.c_double()
for name, build, N in (("C2 double gauss", lambda: scenes.double_gauss(ot, seed=1), 10_000_000),
                       ("relay, 31 surfaces, early stop", relay30, 5_000_000),
                       ("relay no_pol", lambda: relay30(True), 5_000_000)):
    with ot.global_options.no_warnings():
        RT = build()
        RT.trace(100_000)
        _capi.check(lib.ot_scene_set_timing(RT._scene_handle, 1))
        ts = []
        for i in range(40):
            RT.trace(N)
            _capi.check(lib.ot_scene_last_trace_ms(RT._scene_handle, C.byref(ms)))
            if i >= 20:
                ts.append(ms.value)
    w = RT.rays._dev["w"].view(RT.rays.Nt, N)
    alive = [(float((w[k] > 0).float().mean())) for k in (0, 3, RT.rays.Nt - 2)]
    print(f"{name:34s} M={RT.rays.Nt-2:3d} kernel {np.mean(ts):.4f} ms (min {np.min(ts):.4f})  alive at sections 0/3/last-1: "
          f"{alive[0]:.2f} {alive[1]:.2f} {alive[2]:.2f}", flush=True)
