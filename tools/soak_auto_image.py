#!/usr/bin/env python3
"""Random detector images with an automatic extent: the one-pass form (`Raytracer._auto_image_one_pass`) against the chain
hit list -> binning.  Random source image sides, divergence, detector position / size / kind, ray count, sample stride,
grid margins and resolution limit; every case must give the same extent (bit for bit), the same lit pixels and sums to 1e-11.
usage: tools/soak_auto_image.py [cases] [seed]"""
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import numpy as np

import optrace_amd as ot
import scenes

import os
N_CHOICES = [int(v) for v in os.environ.get("SOAK_N", "30000,200000,700001,2000000").split(",")]  # ray counts to draw from
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
applied = declined = bad = 0
for case in range(cases):
    sides = [float(rng.uniform(0.5, 5.0)), float(rng.uniform(0.5, 5.0))]
    z_det = float(rng.uniform(20, 39))
    n = int(rng.choice(N_CHOICES))
    kind = rng.choice(["rect", "circle", "sphere"])
    limit = None if rng.random() < 0.7 else float(rng.uniform(1, 20))
    stride = int(rng.choice([1, 16, 128, 2048]))
    margins = [((0.5, 1024), (0.3, 1024), (0.15, 1024), (0.3, 2048), (0.15, 2048), (0.05, 2048)),
               ((0.0, 2048),), ((0.02, 600), (0.0, 2048)), ((1.0, 2048),)][int(rng.integers(4))]
    with ot.global_options.no_warnings():
        RT = ot.Raytracer(outline=[-10, 10, -10, 10, 0, 40], no_pol=True, seed=int(rng.integers(1 << 30)))
        RT.add(ot.RaySource(ot.RGBImage(scenes.synthetic_rgb_image(), sides), divergence="Isotropic",
                            div_angle=float(rng.uniform(2, 16)), s=[0, 0, 1], pos=[0, 0, 0]))
        RT.add(ot.Lens(ot.SphericalSurface(r=4, R=float(rng.uniform(7, 14))), ot.SphericalSurface(r=4, R=-float(rng.uniform(7, 14))),
                       de=0.1, pos=[0, 0, 12], n=ot.RefractionIndex("Abbe", n=1.5, V=40)))
        if kind == "rect":
            surf = ot.RectangularSurface(dim=[float(rng.uniform(4, 19)), float(rng.uniform(4, 19))])
        elif kind == "circle":
            surf = ot.CircularSurface(r=float(rng.uniform(2, 9)))
        else:
            surf = ot.SphericalSurface(r=float(rng.uniform(2, 8)), R=-float(rng.uniform(12, 40)))
        RT.add(ot.Detector(surf, pos=[float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1)), z_det]))
        RT.trace(n)
        kw = dict(limit=limit, projection_method="Orthographic" if kind == "sphere" else "Equidistant")
        old = (ot.Raytracer.AUTO_ONE_PASS_FROM, ot.Raytracer.AUTO_SAMPLE_STRIDE, ot.Raytracer.AUTO_MARGINS)
        try:
            ot.Raytracer.AUTO_SAMPLE_STRIDE, ot.Raytracer.AUTO_MARGINS = stride, margins
            ot.Raytracer.AUTO_ONE_PASS_FROM = 1
            spec = dict(detector_index=0, source_index=None, extent=None, projection_method=kw["projection_method"])
            one = RT._auto_image_one_pass(spec, limit, _dont_filter=True)
            ot.Raytracer.AUTO_ONE_PASS_FROM = 1 << 60
            chain = RT.detector_image(_dont_filter=True, **kw)
        finally:
            ot.Raytracer.AUTO_ONE_PASS_FROM, ot.Raytracer.AUTO_SAMPLE_STRIDE, ot.Raytracer.AUTO_MARGINS = old
    if one is None:
        declined += 1
        continue
    applied += 1
    A, B = one._data, chain._data
    ok = (A.shape == B.shape and np.array_equal(one.extent, chain.extent)
          and np.array_equal(A[..., 3] != 0, B[..., 3] != 0) and np.abs(A - B).max() <= 1e-11 * np.abs(B).max())
    if not ok:
        bad += 1
        print(f"MISMATCH case {case}: sides={sides} z={z_det:.2f} n={n} {kind} limit={limit} stride={stride} margins={margins[0]} "
              f"shapes {A.shape} {B.shape} extents {one.extent} {chain.extent}", flush=True)
    if case % 25 == 24:
        print(f"... {case + 1} cases: {applied} one-pass images compared, {declined} declined, {bad} mismatches", flush=True)
print(f"{cases} random cases: {applied} one-pass images compared with the chain, {declined} declined (not applicable), "
      f"{bad} mismatches")
sys.exit(1 if bad else 0)
