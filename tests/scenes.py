"""Scene builders shared by the golden-vector generator (run with the reference package) and the tests
(run with optrace_amd).  Every builder takes the package as `ot`, so the same script text drives both --
which is also the drop-in check of the Python API.

Geometry numbers come from the reference's example / test scenes (cited per function); they are data.
"""
from __future__ import annotations

import numpy as np


def c1_single_lens(ot, **rt_args):
    """BASELINE.json configs[0] / SURVEY 8d C1: biconvex spherical lens, monochromatic point source."""
    RT = ot.Raytracer(outline=[-10, 10, -10, 10, -25, 60], **rt_args)
    RT.add(ot.RaySource(ot.Point(), divergence="Isotropic", div_angle=5, pos=[0, 0, -20],
                        spectrum=ot.LightSpectrum("Monochromatic", wl=550.)))
    RT.add(ot.Lens(ot.SphericalSurface(r=3, R=8), ot.SphericalSurface(r=3, R=-8), de=0.1,
                   n=ot.RefractionIndex("Constant", n=1.5), pos=[0, 0, 0]))
    RT.add(ot.Detector(ot.RectangularSurface(dim=[4, 4]), pos=[0, 0, 20]))
    return RT


def double_gauss(ot, spectrum=None, aspheric=False, **rt_args):
    """BASELINE.json configs[1] / C2: Nikkor-Wakamiya 100 mm f/1.4 double Gauss, geometry of
    examples/double_gauss.py:34-102 (US patent 4448497), 5 point sources, FDC line spectrum.
    aspheric=True: every lens FRONT becomes an AsphericSurface of the same vertex radius (k = -0.3, small even
    polynomial) -- the synthetic asphere scene SURVEY 8(a6) asks for: 7 of the 15 surfaces go through the numeric
    (Illinois) hit search, the rest stay closed-form."""
    RT = ot.Raytracer(outline=[-2000, 2000, -22000, 2000, -50000, 180], **rt_args)
    g = 50000
    spectrum = spectrum or ot.LightSpectrum("Lines", lines=[486.1327, 589.2938, 656.272], line_vals=[1, 1, 1])
    for deg in [0, 5, 10, 15, 20]:
        xp = g * np.tan(deg / 180 * np.pi)
        RT.add(ot.RaySource(ot.Point(), divergence="Isotropic", orientation="Converging", conv_pos=[0, 0, 0],
                            div_angle=0.03, pos=[0, -xp, -g], spectrum=spectrum, desc=f"{deg} deg"))

    def lens(r1, R1, r2, R2, n, V, z, d2):
        if aspheric:
            with ot.global_options.no_warnings():
                front = ot.AsphericSurface(r=r1, R=R1, k=-0.3, coeff=[-1e-6 * np.sign(R1), 2e-10 * np.sign(R1)])
        else:
            front = ot.SphericalSurface(r=r1, R=R1)
        L = ot.Lens(front, ot.SphericalSurface(r=r2, R=R2),
                    n=ot.RefractionIndex("Abbe", n=n, V=V), pos=[0, 0, z], d1=0, d2=d2)
        RT.add(L)
        return L

    L0 = lens(76 / 2, 78.36, 76 / 2, 469.5, 1.797, 45.3, 0, 9.8837)
    L1 = lens(64 / 2, 50.3, 62 / 2, 74.38, 1.773, 49.4, L0.back.pos[2] + 0.1938, 9.1085)
    L2 = lens(59 / 2, 138.1, 51 / 2, 34.33, 1.673, 32.20, L1.back.pos[2] + 2.9457, 2.3256)
    RT.add(ot.Aperture(ot.RingSurface(ri=49.6 / 2, r=76 / 2), pos=[0, 0, L2.back.pos[2] + 16.07]))
    L3 = lens(48.8 / 2, -34.41, 57 / 2, -2907, 1.740, 28.30, L2.back.pos[2] + 16.07 + 13, 1.938)
    L4 = lens(57 / 2, -2907, 60 / 2, -59.05, 1.773, 49.40, L3.back.pos[2] + 1e-6, 12.403)
    L5 = lens(66.8 / 2, -150.9, 67.8 / 2, -57.89, 1.788, 47.50, L4.back.pos[2] + 0.3876, 8.333)
    L6 = lens(66 / 2, 284.6, 66 / 2, -253.2, 1.788, 47.50, L5.back.pos[2] + 0.1938, 5.0388)
    RT.add(ot.Detector(ot.RectangularSurface(dim=[86.53, 86.53]), pos=[0, 0, L6.back.pos[2] + 73.839]))
    return RT


def mixed_geometry(ot, **rt_args):
    """Scene of tests/tracing_geometry.py:9-92 without its plot-only markers/volumes: two area sources
    (line + tabulated spectrum), flat lens, conic lenses (k != 0), ring aperture, Function-index lens with a
    different medium behind, Function filter, ideal lens, flat and spherical detectors."""
    RT = ot.Raytracer(outline=[-5, 5, -5, 5, -5, 60], **rt_args)
    RT.add(ot.RaySource(ot.CircularSurface(r=1), divergence="None", spectrum=ot.presets.light_spectrum.FDC,
                        pos=[0, 0, 0], s=[0, 0, 1], polarization="y"))
    RT.add(ot.RaySource(ot.CircularSurface(r=1), divergence="None", s=[0, 0, 1],
                        spectrum=ot.presets.light_spectrum.d65, pos=[0, 1, -3], polarization="Constant",
                        pol_angle=25, power=2))
    RT.add(ot.Lens(ot.CircularSurface(r=3), ot.CircularSurface(r=3), de=0.1, pos=[0, 0, 2],
                   n=ot.RefractionIndex("Constant", n=1.8)))
    RT.add(ot.Lens(ot.ConicSurface(r=3, R=10, k=-0.444), ot.ConicSurface(r=3, R=-10, k=-7.25), de=0.1,
                   pos=[0, 0, 10], n=ot.RefractionIndex("Cauchy", coeff=[1.49, 0.00354, 0, 0])))
    RT.add(ot.Lens(ot.ConicSurface(r=3, R=5, k=-0.31), ot.ConicSurface(r=3, R=-5, k=-3.04), de=0.6,
                   pos=[0, 0, 25], n=ot.RefractionIndex("Constant", n=1.8)))
    RT.add(ot.Aperture(ot.RingSurface(r=1, ri=0.01), pos=[0, 0, 20.3]))
    nL3 = ot.RefractionIndex("Function", func=lambda l: 1.8 - 0.007 * (l - 380) / 400)
    RT.add(ot.Lens(ot.SphericalSurface(r=1, R=2.2), ot.SphericalSurface(r=1, R=-5), de=0.1, pos=[0, 0, 47],
                   n=nL3, n2=ot.RefractionIndex("Constant", n=1.1)))
    fspec = ot.TransmissionSpectrum("Function", func=lambda l: np.exp(-0.5 * (l - 460) ** 2 / 20 ** 2))
    RT.add(ot.Filter(ot.CircularSurface(r=1), pos=[0, 0, 45.2], spectrum=fspec))
    RT.add(ot.Detector(ot.RectangularSurface(dim=[2, 2]), pos=[0, 0, 0]))
    RT.add(ot.Detector(ot.SphericalSurface(R=-1.1, r=1), pos=[0, 0, 40]))
    RT.add(ot.IdealLens(r=3, D=1, pos=[0, 0, RT.outline[5] - 1]))
    return RT


def arizona_eye_scene(ot, **rt_args):
    """BASELINE.json configs[2] / C3 geometry: presets.geometry.arizona_eye(adaptation=1/0.6, pupil=4)
    (examples/arizona_eye_model.py:15-57) lit by a rectangular area source converging onto the eye."""
    RT = ot.Raytracer(outline=[-10, 10, -10, 10, -610, 28], **rt_args)
    RT.add(ot.RaySource(ot.RectangularSurface(dim=[8.39, 8.39]), divergence="Isotropic", div_angle=0.25,
                        orientation="Converging", conv_pos=[0, 0, 0], pos=[0, 0, -600],
                        spectrum=ot.LightSpectrum("Rectangle", wl0=420., wl1=680.)))
    RT.add(ot.presets.geometry.arizona_eye(adaptation=1 / 0.6, pupil=4))
    return RT


def hurb_slit_lens(ot, **rt_args):
    """BASELINE.json configs[4] / C5: slit aperture (examples/hurb_apertures.py:30-36) in water followed by a
    biconvex lens, HURB edge bending on, polarisation tracked."""
    RT = ot.Raytracer(outline=[-3, 3, -3, 3, -5, 40], n0=ot.RefractionIndex("Constant", n=1.33),
                      use_hurb=True, **rt_args)
    RT.add(ot.RaySource(ot.RectangularSurface(dim=[0.05, 2]), divergence="None", s=[0, 0, 1], pos=[0, 0, -4],
                        spectrum=ot.LightSpectrum("Monochromatic", wl=550.)))
    RT.add(ot.Aperture(ot.SlitSurface(dim=[2.5, 2.5], dimi=[0.05, 2]), pos=[0, 0, 0]))
    RT.add(ot.Lens(ot.SphericalSurface(r=1.2, R=18), ot.SphericalSurface(r=1.2, R=-18), de=0.1,
                   n=ot.RefractionIndex("Constant", n=1.5), pos=[0, 0, 12]))
    RT.add(ot.Detector(ot.RectangularSurface(dim=[1.5, 1.5]), pos=[0, 0, 30]))
    return RT


def hurb_ring_ideal(ot, **rt_args):
    """Ring aperture in front of an ideal lens with HURB (tests/hurb_geometry.py:10-82, n=1.2, ri=0.5, 580 nm)."""
    zd = 100.
    RT = ot.Raytracer(outline=[-15, 15, -15, 15, -6, zd + 10], use_hurb=True,
                      n0=ot.RefractionIndex("Constant", n=1.2), **rt_args)
    RT.add(ot.RaySource(ot.CircularSurface(r=0.5), s=[0, 0, 1], pos=[0, 0, -5],
                        spectrum=ot.LightSpectrum("Monochromatic", wl=580.)))
    RT.add(ot.Aperture(ot.RingSurface(r=1.5, ri=0.5), pos=[0, 0, -0.001]))
    RT.add(ot.IdealLens(1.5, 1 / zd * 1000, pos=[0, 0, 0]))
    RT.add(ot.Detector(ot.RectangularSurface(dim=[0.5, 0.5]), pos=[0, 0, zd]))
    return RT


def asphere_scene(ot, **rt_args):
    """Synthetic scene for the numeric (Illinois) hit search: an aspheric singlet, a conic (k != 0) lens, a
    rectangular filter and a rotated rectangular aperture; tilted Lambertian disc source so that some rays
    miss, hit the outline or are totally reflected."""
    RT = ot.Raytracer(outline=[-8, 8, -8, 8, -12, 50], **rt_args)
    RT.add(ot.RaySource(ot.CircularSurface(r=2.0), divergence="Lambertian", div_angle=14, pos=[0.3, -0.2, -10],
                        s=[0.02, 0.05, 1], spectrum=ot.LightSpectrum("Gaussian", mu=540., sig=40.),
                        polarization="Uniform"))
    with ot.global_options.no_warnings():
        front = ot.AsphericSurface(r=4, R=12, k=-0.8, coeff=[2e-3, -4e-5, 3e-7])
        back = ot.AsphericSurface(r=4, R=-15, k=0.3, coeff=[-1e-3, 2e-5])
    RT.add(ot.Lens(front, back, de=0.4, pos=[0, 0, 0],
                   n=ot.RefractionIndex("Sellmeier1", coeff=[1.03961212, 0.00600069867, 0.231792344,
                                                             0.0200179144, 1.01046945, 103.560653])))
    filt = ot.RectangularSurface(dim=[5, 4])
    RT.add(ot.Filter(filt, pos=[0, 0, 6], spectrum=ot.TransmissionSpectrum("Gaussian", mu=550., sig=60., val=0.9)))
    RT.add(ot.Lens(ot.ConicSurface(r=3.5, R=9, k=-2.1), ot.ConicSurface(r=3.5, R=-30, k=1.5), de=0.3,
                   pos=[0, 0.1, 12], n=ot.RefractionIndex("Schott", coeff=[2.7, -0.01, 0.03, 1e-3, -1e-4, 1e-5]),
                   n2=ot.RefractionIndex("Constant", n=1.2)))
    ap = ot.RectangularSurface(dim=[6, 6])
    ap.rotate(20)
    slit = ot.SlitSurface(dim=[7, 7], dimi=[3, 2])
    slit.rotate(-12)
    RT.add(ot.Aperture(slit, pos=[0, 0, 20]))
    RT.add(ot.Filter(ap, pos=[0, 0, 24], spectrum=ot.TransmissionSpectrum("Rectangle", wl0=450., wl1=640.,
                                                                            val=0.8, inverse=True)))
    RT.add(ot.Detector(ot.SphericalSurface(r=5, R=-20), pos=[0, 0, 35]))
    RT.add(ot.Detector(ot.RectangularSurface(dim=[8, 8]), pos=[0, 0, 30]))
    return RT


def prism_scene(ot, **rt_args):
    """Geometry of examples/prism.py:16-36: thin D65 beam through a prism made of two tilted circles
    (N-LAK8 as Sellmeier1 coefficients), flat detector."""
    n = ot.RefractionIndex("Sellmeier1", coeff=[1.33183167, 0.00620023871, 0.546623206, 0.0216465439,
                                                1.19084015, 82.5827736])
    RT = ot.Raytracer(outline=[-5, 5, -5, 5, -5, 25], **rt_args)
    RT.add(ot.RaySource(ot.CircularSurface(r=0.05), divergence="None", spectrum=ot.presets.light_spectrum.d65,
                        pos=[0, -2.5, 0], s=[0, 0.3, 0.7]))
    front = ot.TiltedSurface(r=3, normal=[0, -0.45, np.sqrt(1 - 0.45 ** 2)])
    back = front.copy()
    back.rotate(180)
    RT.add(ot.Lens(front, back, de=0.5, pos=[0, 0, 10], n=n))
    RT.add(ot.Detector(ot.RectangularSurface(dim=[10, 10]), pos=[0, 0, 20]))
    return RT


def _cosine_surface(x, y):
    return 0.1 * np.cos(2 * np.pi * x / 2)


def freeform_scene(ot, ideal_lens=True, **rt_args):
    """Function, data and tilted surfaces in one beam path: the cosine-modulated lens of
    examples/cosine_surfaces.py:20-32 (FunctionSurface2D, back side flipped and rotated by 90 deg), a lens
    from a 1-D data profile against a 2-D data grid, a tilted window and an ideal lens."""
    RT = ot.Raytracer(outline=[-15, 15, -15, 15, 0, 80], **rt_args)
    RT.add(ot.RaySource(ot.CircularSurface(r=3), divergence="None", s=[0, 0, 1], pos=[0, 0, 0],
                        spectrum=ot.LightSpectrum("Rectangle", wl0=450., wl1=650.)))
    front = ot.FunctionSurface2D(func=_cosine_surface, r=5)
    back = front.copy()
    back.flip()
    back.rotate(90)
    nL1 = ot.RefractionIndex("Sellmeier1", coeff=[1.52481889, 0.011254756, 0.187085527, 0.0588995392,
                                                  1.42729015, 129.141675])
    RT.add(ot.Lens(front, back, de=2, pos=[0, 0, 12], n=nL1))
    r = np.linspace(0, 6.0, 400)
    d1 = ot.DataSurface1D(r=6.0, data=25 - np.sqrt(625 - r ** 2))
    xy = np.linspace(-6.0, 6.0, 240)
    X, Y = np.meshgrid(xy, xy)
    d2 = ot.DataSurface2D(r=6.0, data=-(X ** 2 / 60 + Y ** 2 / 45) + 0.01 * np.sin(X) * np.cos(0.7 * Y))
    d2.rotate(30)
    RT.add(ot.Lens(d1, d2, de=0.4, pos=[0, 0, 20], n=ot.RefractionIndex("Constant", n=1.55)))
    t1 = ot.TiltedSurface(r=7, normal_sph=[8., 35.])
    t2 = ot.TiltedSurface(r=7, normal_sph=[5., 200.])
    RT.add(ot.Lens(t1, t2, de=0.3, pos=[0, 0, 30], n=ot.RefractionIndex("Cauchy", coeff=[1.49, 0.00354, 0, 0])))
    if ideal_lens:
        RT.add(ot.IdealLens(r=9, D=50, pos=[0, 0, 40]))
    RT.add(ot.Detector(ot.RectangularSurface(dim=[14, 14]), pos=[0, 0, 24.4]))
    RT.add(ot.Detector(ot.RectangularSurface(dim=[14, 14]), pos=[0, 0, 60]))
    return RT


# ---- function surfaces with a mask_func -------------------------------------------------------------------------------
def _paraboloid(x, y):
    return 0.02 * (x ** 2 + 0.6 * y ** 2)


def _window(x, y):
    """Rectangular window: its borders fall on cell borders of the mask bitmap (r = 5, a power-of-two cell count), where
    the bitmap classifies exactly like the callable."""
    return (np.abs(x) <= 2.5) & (np.abs(y) <= 1.875)


def _bowl(r):
    return -0.015 * r ** 2


def _inner_disc(r):
    return r <= 2.5


def _half_plane(x, y):
    return x >= -0.625


def masked_scene(ot, **rt_args):
    """FunctionSurface2D / FunctionSurface1D with mask_func (function_surface_2d.py:158-191): a lens whose front is
    defined on a rectangular window and whose back on an inner disc, then a plate whose front keeps a half plane and
    is flipped and rotated by 90 degrees.  Rays outside a mask do not hit the lens and are absorbed."""
    RT = ot.Raytracer(outline=[-12, 12, -12, 12, 0, 60], **rt_args)
    RT.add(ot.RaySource(ot.CircularSurface(r=3.2), divergence="Isotropic", div_angle=1.5, s=[0, 0, 1], pos=[0, 0, 0],
                        spectrum=ot.LightSpectrum("Rectangle", wl0=480., wl1=620.)))
    front = ot.FunctionSurface2D(func=_paraboloid, mask_func=_window, r=5)
    back = ot.FunctionSurface1D(func=_bowl, mask_func=_inner_disc, r=5)
    RT.add(ot.Lens(front, back, d=1.2, pos=[0, 0, 12], n=ot.RefractionIndex("Constant", n=1.52)))
    half = ot.FunctionSurface2D(func=_paraboloid, mask_func=_half_plane, r=5)
    half.flip()
    half.rotate(90)
    RT.add(ot.Lens(half, ot.CircularSurface(r=5), d=0.8, pos=[0, 0, 22], n=ot.RefractionIndex("Constant", n=1.6)))
    RT.add(ot.Detector(ot.RectangularSurface(dim=[12, 12]), pos=[0, 0, 40]))
    return RT


def _zeros(x, y):
    return np.zeros_like(x)


def _lower_part(x, y):
    return y <= -0.625


def _quadrant_gap(x, y):
    """everything but the quadrant x > 0.625, y > 0 (borders on cell borders of the bitmap)"""
    return ~((x > 0.625) & (y > 0))


def masked_flat_scene(ot, **rt_args):
    """FLAT function surfaces with a mask_func and nothing else numeric in the scene -- the custom apertures of the
    reference's documentation (docs/source/usage/surfaces.rst:360-369: FunctionSurface2D(func=zeros, mask_func=...)):
    as Aperture (the part y <= -0.625 blocks), as Filter (window attenuates) and as the front face of a plate (rays outside the
    mask miss the lens and are absorbed).  Such a scene needs no numeric hit search, yet every hit test has to consult
    the mask bitmap."""
    RT = ot.Raytracer(outline=[-12, 12, -12, 12, 0, 60], **rt_args)
    RT.add(ot.RaySource(ot.CircularSurface(r=3.2), divergence="Isotropic", div_angle=1.5, s=[0, 0, 1], pos=[0, 0, 0],
                        spectrum=ot.LightSpectrum("Rectangle", wl0=480., wl1=620.)))
    ap = ot.FunctionSurface2D(func=_zeros, mask_func=_lower_part, r=5)
    RT.add(ot.Aperture(ap, pos=[0, 0, 8]))
    win = ot.FunctionSurface2D(func=_zeros, mask_func=_window, r=5)
    win.rotate(90)
    RT.add(ot.Filter(win, pos=[0, 0, 14], spectrum=ot.TransmissionSpectrum("Constant", val=0.5)))
    face = ot.FunctionSurface2D(func=_zeros, mask_func=_quadrant_gap, r=5)
    RT.add(ot.Lens(face, ot.SphericalSurface(r=5, R=-30), d=1.0, pos=[0, 0, 22], n=ot.RefractionIndex("Constant", n=1.55)))
    RT.add(ot.Detector(ot.RectangularSurface(dim=[12, 12]), pos=[0, 0, 40]))
    return RT


def double_gauss_aspheric(ot, **rt_args):
    return double_gauss(ot, aspheric=True, **rt_args)


SCENES3 = {"masked": (masked_scene, 3000), "masked_flat": (masked_flat_scene, 3000),
           "double_gauss_aspheric": (double_gauss_aspheric, 2000)}


# ---- random scenes (tests/test_gpu_random_scenes.py, golden trace_random*.npz) -------------------------------------
def random_surface(ot, rng, r):
    kind = rng.integers(0, 5)
    if kind == 0:
        return ot.CircularSurface(r=r)
    if kind == 1:
        R = rng.choice([-1, 1]) * rng.uniform(2.5 * r, 12 * r)
        return ot.SphericalSurface(r=r, R=R)
    if kind == 2:
        R = rng.choice([-1, 1]) * rng.uniform(3 * r, 12 * r)
        return ot.ConicSurface(r=r, R=R, k=rng.uniform(-3, 0.8))
    if kind == 3:
        R = rng.choice([-1, 1]) * rng.uniform(4 * r, 12 * r)
        return ot.AsphericSurface(r=r, R=R, k=rng.uniform(-1, 0.3),
                                  coeff=[rng.uniform(-2e-3, 2e-3) / r, rng.uniform(-2e-4, 2e-4) / r ** 3])
    th, ph = rng.uniform(0, 12), rng.uniform(0, 360)
    return ot.TiltedSurface(r=r, normal_sph=[th, ph])


def random_medium(ot, rng):
    kind = rng.integers(0, 4)
    if kind == 0:
        return ot.RefractionIndex("Constant", n=rng.uniform(1.3, 1.9))
    if kind == 1:
        return ot.RefractionIndex("Abbe", n=rng.uniform(1.45, 1.8), V=rng.uniform(25, 70))
    if kind == 2:
        return ot.RefractionIndex("Cauchy", coeff=[rng.uniform(1.4, 1.7), rng.uniform(0.002, 0.01), 0, 0])
    return ot.RefractionIndex("Sellmeier1", coeff=[1.03961212, 0.00600069867, 0.231792344, 0.0200179144, 1.01046945,
                                                   103.560653])


def random_scene(ot, scene_seed, **rt_args):
    """Seeded random sequential system: 2-5 elements of random type (lenses with flat / spherical / conic / aspheric /
    tilted surfaces and random media, ring and slit apertures, filters, ideal lenses), one random source."""
    rng = np.random.default_rng(scene_seed)
    RT = ot.Raytracer(outline=[-12, 12, -12, 12, -30, 120], no_pol=bool(rng.integers(0, 2)),
                      n0=ot.RefractionIndex("Constant", n=rng.choice([1.0, 1.0, 1.33])), **rt_args)
    spec = [ot.LightSpectrum("Monochromatic", wl=float(rng.uniform(420, 680))), ot.presets.light_spectrum.d65,
            ot.LightSpectrum("Lines", lines=[450., 550., 650.], line_vals=[1., 2., 1.]),
            ot.LightSpectrum("Rectangle", wl0=450., wl1=650.)][rng.integers(0, 4)]
    if rng.integers(0, 2):
        RT.add(ot.RaySource(ot.CircularSurface(r=rng.uniform(0.5, 2.5)), divergence="Lambertian",
                            div_angle=rng.uniform(1, 6), pos=[0, 0, -20], spectrum=spec,
                            polarization=["x", "y", "Uniform"][rng.integers(0, 3)]))
    else:
        RT.add(ot.RaySource(ot.Point(), divergence="Isotropic", div_angle=rng.uniform(3, 9), pos=[rng.uniform(-1, 1), 0, -25],
                            spectrum=spec))
    z = 0.0
    for _ in range(rng.integers(2, 6)):
        what = rng.integers(0, 10)
        r = rng.uniform(2.5, 5.0)
        if what < 6:
            RT.add(ot.Lens(random_surface(ot, rng, r), random_surface(ot, rng, r), de=rng.uniform(0.3, 1.0), pos=[0, 0, z],
                           n=random_medium(ot, rng), n2=random_medium(ot, rng) if rng.integers(0, 4) == 0 else None))
            z += rng.uniform(9, 16)
        elif what == 6:
            RT.add(ot.Aperture(ot.RingSurface(r=r + 1, ri=rng.uniform(0.8, 2.5)), pos=[0, 0, z]))
            z += rng.uniform(3, 6)
        elif what == 7:
            RT.add(ot.Filter(ot.CircularSurface(r=r), pos=[0, 0, z],
                             spectrum=ot.TransmissionSpectrum("Rectangle", wl0=430., wl1=640., val=0.8)))
            z += rng.uniform(3, 6)
        elif what == 8:
            RT.add(ot.IdealLens(r=r, D=float(rng.choice([-1, 1]) * rng.uniform(15, 60)), pos=[0, 0, z]))
            z += rng.uniform(6, 12)
        else:
            RT.add(ot.Aperture(ot.SlitSurface(dim=[2 * r, 2 * r], dimi=[rng.uniform(0.5, 2), rng.uniform(1, 3)]), pos=[0, 0, z]))
            z += rng.uniform(3, 6)
    return RT



SCENES2 = {  # SURVEY 8f rank 4 surfaces; kept apart from SCENES so that the seeds of the older fixtures stay put
    "prism": (prism_scene, 2000),
    "freeform": (freeform_scene, 2500),
    # six random systems picked for variety (apertures, filters, ideal lenses, aspheric and tilted surfaces, n2 media)
    **{f"random{k}": ((lambda k_: (lambda ot, **kw: random_scene(ot, k_, **kw)))(k), 1500)
       for k in (2014, 2016, 2026, 2036, 2053, 2058)},
}


SCENES = {
    "c1_single_lens": (c1_single_lens, 2000),
    "double_gauss": (double_gauss, 1500),
    "mixed_geometry": (mixed_geometry, 2500),
    "arizona_eye": (arizona_eye_scene, 2000),
    "hurb_slit_lens": (hurb_slit_lens, 2000),
    "hurb_ring_ideal": (hurb_ring_ideal, 2000),
    "asphere": (asphere_scene, 2500),
}


def surface_zoo(ot):
    """name -> surface, covering every flavour the device kernels implement."""
    z = {}
    z["circle"] = ot.CircularSurface(r=2.5)
    z["ring"] = ot.RingSurface(r=3.0, ri=0.7)
    z["rect"] = ot.RectangularSurface(dim=[3.0, 2.0])
    r2 = ot.RectangularSurface(dim=[4.0, 1.5])
    r2.rotate(33.0)
    z["rect_rot"] = r2
    z["slit"] = ot.SlitSurface(dim=[4.0, 4.0], dimi=[0.2, 2.5])
    s2 = ot.SlitSurface(dim=[5.0, 3.0], dimi=[1.0, 0.3])
    s2.rotate(-20.0)
    z["slit_rot"] = s2
    z["sphere_pos"] = ot.SphericalSurface(r=3.0, R=8.0)
    z["sphere_neg"] = ot.SphericalSurface(r=2.0, R=-5.0)
    z["conic_m025"] = ot.ConicSurface(r=3.0, R=7.8, k=-0.25)
    z["conic_m75"] = ot.ConicSurface(r=3.0, R=12.0, k=-7.5)
    z["conic_p3"] = ot.ConicSurface(r=2.0, R=-9.0, k=3.0)
    z["conic_parab"] = ot.ConicSurface(r=3.0, R=6.0, k=-1.0)
    z["asphere_a"] = ot.AsphericSurface(r=4, R=12, k=-0.8, coeff=[2e-3, -4e-5, 3e-7])
    z["asphere_b"] = ot.AsphericSurface(r=3, R=-15, k=0.3, coeff=[-1e-3, 2e-5])
    for j, (name, s) in enumerate(z.items()):
        s.move_to([0.1 * j - 0.5, 0.3 - 0.07 * j, 2.0 + 0.5 * j])
    return z


def surface_zoo3(ot):
    """Aspheres of every coefficient count the device code distinguishes (the hit search is compiled per count for
    1 .. 4 coefficients and jumps into an unrolled chain for 5 .. 12, ot_device.hpp::AsphereSag), flipped and strongly
    conic ones; own fixture file so that the random streams of the older zoos stay put."""
    z = {}
    with ot.global_options.no_warnings():
        z["asph_c1"] = ot.AsphericSurface(r=3.0, R=10.0, k=-1.3, coeff=[1.5e-3])
        z["asph_c2_neg"] = ot.AsphericSurface(r=3.5, R=-9.0, k=0.6, coeff=[-2e-3, 6e-5])
        z["asph_c4"] = ot.AsphericSurface(r=4.0, R=14.0, k=-0.4, coeff=[1e-3, -3e-5, 4e-7, -2e-9])
        z["asph_c5"] = ot.AsphericSurface(r=4.0, R=-16.0, k=2.0, coeff=[-8e-4, 2e-5, -3e-7, 2e-9, -5e-12])
        z["asph_c8"] = ot.AsphericSurface(r=3.0, R=11.0, k=-2.5,
                                          coeff=[2e-3, -1e-4, 3e-6, -5e-8, 6e-10, -4e-12, 2e-14, -1e-16])
        z["asph_c12"] = ot.AsphericSurface(r=2.5, R=8.0, k=0.0,
                                           coeff=[1e-3, -2e-4, 3e-5, -4e-6, 5e-7, -6e-8, 7e-9, -8e-10, 9e-11, -1e-11,
                                                  1e-12, -1e-13])
        f = ot.AsphericSurface(r=3.0, R=12.0, k=-0.7, coeff=[1e-3, -2e-5, 1e-7])
        f.flip()
        z["asph_c3_flipped"] = f
    z["conic_oblate"] = ot.ConicSurface(r=2.5, R=6.0, k=1.8)
    z["conic_hyper_neg"] = ot.ConicSurface(r=3.0, R=-7.0, k=-3.2)
    for j, (name, s) in enumerate(z.items()):
        s.move_to([0.08 * j - 0.3, 0.2 - 0.05 * j, 1.5 + 0.4 * j])
    return z


def _func2d(x, y, a=10.):
    return (x ** 2 + y ** 2 / 3) / a + 0.05 * np.cos(x)


def _func2d_deriv(x, y, a=10.):
    return 2 * x / a - 0.05 * np.sin(x), 2 * y / 3 / a


def surface_zoo2(ot):
    """Tilted, data and function surfaces; same role as surface_zoo."""
    z = {}
    z["tilted"] = ot.TiltedSurface(r=3.0, normal=[0, -0.45, np.sqrt(1 - 0.45 ** 2)])
    t2 = ot.TiltedSurface(r=2.0, normal_sph=[20., 70.])
    t2.rotate(15.0)
    t2.flip()
    z["tilted_sph"] = t2
    r = np.linspace(0, 3.0, 220)
    z["data1d"] = ot.DataSurface1D(r=3.0, data=10 - np.sqrt(100 - r ** 2))
    d1f = ot.DataSurface1D(r=2.0, data=0.02 * np.linspace(0, 2.0, 300) ** 3 + 1.0)
    d1f.flip()
    z["data1d_flip"] = d1f
    xy = np.linspace(-2.5, 2.5, 120)
    X, Y = np.meshgrid(xy, xy)
    z["data2d"] = ot.DataSurface2D(r=2.5, data=X ** 2 / 20 + Y ** 2 / 14 + 0.02 * np.sin(2 * X))
    xy = np.linspace(-2.0, 2.0, 101)
    X, Y = np.meshgrid(xy, xy)
    d2 = ot.DataSurface2D(r=2.0, data=0.3 - (X - 0.3) ** 2 / 9 - Y ** 2 / 30 + 0.01 * X * Y)
    d2.rotate(25.0)
    d2.flip()
    z["data2d_rot_flip"] = d2
    z["func1d"] = ot.FunctionSurface1D(r=3.0, func=lambda r: r ** 2 / 16 + 0.001 * r ** 4,
                                       deriv_func=lambda r: r / 8 + 0.004 * r ** 3)
    z["func2d"] = ot.FunctionSurface2D(r=3.0, func=_func2d, deriv_func=_func2d_deriv)
    f2 = ot.FunctionSurface2D(r=2.5, func=_func2d, func_args=dict(a=7.), deriv_func=_func2d_deriv,
                              deriv_args=dict(a=7.), z_min=0.05, z_max=0.05 + 1.0)
    f2.rotate(-40.0)
    z["func2d_rot"] = f2
    z["func2d_noderiv"] = ot.FunctionSurface2D(r=2.0, func=lambda x, y: 0.1 * np.sin(x) * y + x ** 2 / 12)
    for j, (name, s) in enumerate(z.items()):
        s.move_to([0.1 * j - 0.5, 0.3 - 0.07 * j, 2.0 + 0.5 * j])
    return z


MEDIA = {
    "Constant": dict(n=1.5),
    "Abbe": dict(n=1.797, V=45.3),
    "Abbe_lines": dict(n=1.6, V=30.0, lines=[479.9914, 546.0740, 643.8469]),
    "Cauchy": dict(coeff=[1.49, 0.00354, 1e-5, 2e-7]),
    "Conrady": dict(coeff=[1.5, 0.01, 0.0005]),
    "Sellmeier1": dict(coeff=[1.03961212, 0.00600069867, 0.231792344, 0.0200179144, 1.01046945, 103.560653]),
    "Sellmeier2": dict(coeff=[1.2, 0.9, 0.1, 0.002, 0.12]),
    "Sellmeier3": dict(coeff=[0.7, 0.005, 0.4, 0.014, 0.9, 97.9, 0.01, 0.02]),
    "Sellmeier4": dict(coeff=[1.9, 0.4, 0.02, 1.0, 100.0]),
    "Sellmeier5": dict(coeff=[0.6, 0.004, 0.4, 0.013, 0.8, 95.0, 0.01, 0.02, 0.002, 0.03]),
    "Schott": dict(coeff=[2.27, -0.01, 0.01, 1e-4, -1e-6, 1e-7]),
    "Herzberger": dict(coeff=[1.5, 0.005, 1e-4, -0.002, 1e-4, -1e-5]),
    "Handbook of Optics 1": dict(coeff=[2.2, 0.01, 0.02, 0.01]),
    "Handbook of Optics 2": dict(coeff=[1.2, 1.0, 0.01, 0.01]),
    "Extended": dict(coeff=[2.25, -0.009, 0.012, 2e-4, -1e-5, 1e-6, -1e-8, 1e-9]),
    "Extended2": dict(coeff=[2.25, -0.009, 0.012, 2e-4, -1e-5, 1e-6, 1e-4, -1e-5]),
    "Extended3": dict(coeff=[2.25, -0.009, 1e-4, 0.012, 2e-4, -1e-5, 1e-6, -1e-7, 1e-9]),
    "Data": dict(wls=np.linspace(380., 780., 41), vals=1.5 + 0.1 * np.exp(-np.linspace(0, 3, 41))),
}


def transmission_zoo(ot):
    return {
        "Constant": ot.TransmissionSpectrum("Constant", val=0.6),
        "Constant_inv": ot.TransmissionSpectrum("Constant", val=0.6, inverse=True),
        "Rectangle": ot.TransmissionSpectrum("Rectangle", wl0=450., wl1=640., val=0.8),
        "Gaussian": ot.TransmissionSpectrum("Gaussian", mu=550., sig=60., val=0.9),
        "Gaussian_inv": ot.TransmissionSpectrum("Gaussian", mu=500., sig=30., val=1.0, inverse=True),
        "Data": ot.TransmissionSpectrum("Data", wls=np.linspace(400., 700., 31), vals=np.linspace(0.1, 0.9, 31) ** 2),
    }


def synthetic_rgb_image():
    """24 x 32 sRGB test card: red / green / blue / white column blocks with a vertical brightness ramp and a
    black stripe (pixels that must never emit)."""
    H, W = 24, 32
    img = np.zeros((H, W, 3))
    ramp = np.linspace(0.15, 1.0, H)[:, None]
    img[:, 0:8, 0] = ramp
    img[:, 8:16, 1] = ramp
    img[:, 16:24, 2] = ramp
    img[:, 24:32, :] = ramp[:, :, None]
    img[10:12, :, :] = 0.0
    return img


def synthetic_gray_image():
    H, W = 16, 16
    y, x = np.mgrid[0:H, 0:W]
    g = ((x + 2 * y) % 7) / 6.0
    g[4:6, 4:6] = 0.0
    return g


def c3_arizona_eye_rgb(ot, seed=31, **rt_args):
    """BASELINE.json configs[2] / C3 at full size: examples/arizona_eye_model.py:15-57 -- an RGB image source (a synthetic
    256 x 256 card here, no image files travel) converging onto the Arizona eye model (adaptation 1 / 0.6 D, pupil 4 mm);
    continuous spectra of the sRGB primaries, aspheric / conic surfaces, spherical retina detector."""
    rng = np.random.default_rng(7)
    yy, xx = np.mgrid[0:256, 0:256]
    rgb = np.stack([(xx // 32 + yy // 32) % 2 * 0.8 + 0.1, xx / 255., yy / 255.], axis=2) * rng.uniform(0.9, 1, (256, 256, 1))
    RT = ot.Raytracer(outline=[-10, 10, -10, 10, -610, 28], seed=seed, **rt_args)
    RT.add(ot.RaySource(ot.RGBImage(rgb, [8.39, 8.39]), divergence="Isotropic", div_angle=0.25,
                        orientation="Converging", conv_pos=[0, 0, 0], pos=[0, 0, -600]))
    RT.add(ot.presets.geometry.arizona_eye(adaptation=1 / 0.6, pupil=4))
    return RT


C4_POSITIONS = [[0, 0, z] for z in (30., 32., 34., 36., 38., 39.5)]
"""image_render_many_rays.py:39-41 renders its detector at several positions: a sweep through the image plane"""


def c4_image_render(ot, seed=41, **rt_args):
    """BASELINE.json configs[3] / C4: examples/image_render_many_rays.py:11-41 -- RGB image source -> biconvex lens ->
    square detector, no_pol."""
    RT = ot.Raytracer(outline=[-8, 8, -8, 8, 0, 40], no_pol=True, seed=seed, **rt_args)
    RT.add(ot.RaySource(ot.RGBImage(synthetic_rgb_image(), [4, 3]), divergence="Isotropic",
                        div_angle=np.rad2deg(np.arctan(3 / 12) * 1.2), s=[0, 0, 1], pos=[0, 0, 0],
                        orientation="Converging", conv_pos=[0, 0, 12]))
    RT.add(ot.Lens(ot.SphericalSurface(r=3, R=8), ot.SphericalSurface(r=3, R=-8), de=0.1, pos=[0, 0, 12],
                   n=ot.RefractionIndex("Abbe", n=1.5, V=40)))
    RT.add(ot.Detector(ot.RectangularSurface(dim=[16, 16]), pos=[0, 0, 36]))
    return RT
