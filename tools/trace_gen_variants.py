#!/usr/bin/env python3
"""Ray generation alone (no surfaces) for variants of C4's source: which part of create_rays costs what (counter runs).
Usage: trace_gen_variants.py <variant> [N]   variants: rgb_full, rgb_plain (no divergence, constant orientation),
gray_full, rect_full (rectangle with a D65 spectrum, isotropic + converging), rect_plain, point_plain"""
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests"), str(ROOT / "tools")]
which = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
import numpy as np
import torch

import optrace_amd as ot
import scenes

full = dict(divergence="Isotropic", div_angle=np.rad2deg(np.arctan(3 / 12) * 1.2), orientation="Converging", conv_pos=[0, 0, 12])
plain = dict(divergence="None", s=[0, 0, 1])
rgb = scenes.synthetic_rgb_image()
with ot.global_options.no_warnings():
    RT = ot.Raytracer(outline=[-8, 8, -8, 8, 0, 40], no_pol=True, seed=41)
    if which == "rgb_full":
        RT.add(ot.RaySource(ot.RGBImage(rgb, [4, 3]), pos=[0, 0, 0], **full))
    elif which == "rgb_plain":
        RT.add(ot.RaySource(ot.RGBImage(rgb, [4, 3]), pos=[0, 0, 0], **plain))
    elif which == "gray_full":
        RT.add(ot.RaySource(ot.GrayscaleImage(rgb.mean(axis=2), [4, 3]), pos=[0, 0, 0], **full))
    elif which == "rect_full":
        RT.add(ot.RaySource(ot.RectangularSurface(dim=[4, 3]), pos=[0, 0, 0], **full))
    elif which == "rect_plain":
        RT.add(ot.RaySource(ot.RectangularSurface(dim=[4, 3]), pos=[0, 0, 0], **plain))
    elif which == "point_plain":
        RT.add(ot.RaySource(ot.Point(), pos=[0, 0, 0], spectrum=ot.LightSpectrum("Monochromatic", wl=550.), **plain))
    elif which == "point_d65_plain":
        RT.add(ot.RaySource(ot.Point(), pos=[0, 0, 0], **plain))
    elif which == "point_lines_plain":
        RT.add(ot.RaySource(ot.Point(), pos=[0, 0, 0], spectrum=ot.presets.light_spectrum.FDC, **plain))
    elif which == "rect_mono_plain":
        RT.add(ot.RaySource(ot.RectangularSurface(dim=[4, 3]), pos=[0, 0, 0], spectrum=ot.LightSpectrum("Monochromatic", wl=550.), **plain))
    elif which == "point_mono_iso":
        RT.add(ot.RaySource(ot.Point(), pos=[0, 0, 0], spectrum=ot.LightSpectrum("Monochromatic", wl=550.),
                            divergence="Isotropic", div_angle=16., s=[0, 0, 1]))
    elif which == "rect_mono_conv":
        RT.add(ot.RaySource(ot.RectangularSurface(dim=[4, 3]), pos=[0, 0, 0], spectrum=ot.LightSpectrum("Monochromatic", wl=550.),
                            divergence="None", orientation="Converging", conv_pos=[0, 0, 12]))
    elif which == "rect_mono_full":
        RT.add(ot.RaySource(ot.RectangularSurface(dim=[4, 3]), pos=[0, 0, 0], spectrum=ot.LightSpectrum("Monochromatic", wl=550.), **full))
    elif which == "point_const_plain":
        RT.add(ot.RaySource(ot.Point(), pos=[0, 0, 0], spectrum=ot.LightSpectrum("Constant"), **plain))
    elif which == "point_data_plain":
        wls = np.linspace(400., 700., 101)
        RT.add(ot.RaySource(ot.Point(), pos=[0, 0, 0], spectrum=ot.LightSpectrum("Data", wls=wls, vals=1 + 0.5 * np.sin(wls / 30)), **plain))
    elif which == "point_gauss_plain":
        RT.add(ot.RaySource(ot.Point(), pos=[0, 0, 0], spectrum=ot.LightSpectrum("Gaussian", mu=550., sig=30.), **plain))
    elif which in ("point_mono_lamb", "point_mono_iso2d", "point_mono_iso_pol"):
        kw = dict(divergence="Lambertian" if which.endswith("lamb") else "Isotropic", div_angle=16., s=[0, 0, 1])
        if which.endswith("2d"):
            kw.update(div_2d=True, div_axis_angle=20.)
        RT.add(ot.RaySource(ot.Point(), pos=[0, 0, 0], spectrum=ot.LightSpectrum("Monochromatic", wl=550.), **kw))
    elif which == "ring_mono_plain":
        RT.add(ot.RaySource(ot.CircularSurface(r=2), pos=[0, 0, 0], spectrum=ot.LightSpectrum("Monochromatic", wl=550.), **plain))
    elif which in ("eye_point_lines", "eye_point_cont", "eye_none"):
        # the Arizona eye of C3 (conic surfaces, k != 0, polarisation on) behind a cheap source: cost of its surface steps
        RT = ot.Raytracer(outline=[-10, 10, -10, 10, -610, 28], seed=31)
        spec = ot.presets.light_spectrum.FDC if which == "eye_point_lines" else ot.LightSpectrum("Constant")
        RT.add(ot.RaySource(ot.Point(), divergence="Isotropic", div_angle=0.25, s=[0, 0, 1], pos=[0, 0, -600], spectrum=spec))
        if which != "eye_none":
            RT.add(ot.presets.geometry.arizona_eye(adaptation=1 / 0.6, pupil=4))
        N = 50_000_000
    elif which in ("inject_cont", "inject_lines"):
        RT.add(ot.RaySource(ot.Point(), pos=[0, 0, 0], spectrum=ot.LightSpectrum("Constant") if which == "inject_cont"
                            else ot.presets.light_spectrum.FDC, **plain))
    else:
        raise SystemExit("unknown variant")
    if which.startswith("inject"):   # rays handed in: the kernel without generation (GEN = false)
        N = 20_000_000
        rng = np.random.default_rng(1)
        p0 = np.zeros((N, 3))
        s0 = np.tile([0., 0., 1.], (N, 1))
        wl = rng.uniform(400, 700, N).astype(np.float32) if which == "inject_cont" else \
            rng.choice(np.float32([486.1327, 589.2938, 656.272]), N)
        init = (p0, s0, None, np.full(N, 1. / N, dtype=np.float32), wl)
        for _ in range(3):
            RT.trace(N, _initial_rays=init, _N_list=np.array([N]))
        torch.cuda.synchronize()
        print(which, N, RT.rays.Nt)
        raise SystemExit(0)
    for _ in range(3):
        RT.trace(N)
torch.cuda.synchronize()
print(which, N, RT.rays.Nt)
