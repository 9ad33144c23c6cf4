"""Edge diffraction (HURB) against diffraction theory -- the reference's own acceptance tests for this part of the path,
restated: tests/test_tracer_hurb.py:19-228 with the set-ups of tests/hurb_geometry.py (pupil in front of an ideal lens,
pinhole, slit, straight edge).  The reference marks the four profile tests "slow" (1-5 M rays each on the CPU); here they
run as they stand.  Same parameters, same statistics, same bounds: ratio of the RMS widths of simulated and theoretical
profile 0.95 +- 0.04 (Airy patterns), 1.11 +- 0.05 / 0.09 (slit, both axes), RMS deviations 0.02 / 0.015 (edge).

The theory curves are textbook formulas (Airy pattern (2 J1(x) / x)^2, sinc^2 of a slit, Fresnel integrals of a straight
edge); the tracer side is `Raytracer(use_hurb=True)` = raytracer.py:418-508 on the device (`ot_trace.hpp::hurb_step`)."""
import numpy as np
import pytest
import scipy.interpolate
import scipy.ndimage
import scipy.special

import optrace_amd as ot
import scenes

pytestmark = pytest.mark.gpu


# ---- theory ------------------------------------------------------------------------------------------------------------
def airy_pattern(r_mm, wl_nm, n, radius_mm, z_mm):
    """Fraunhofer pattern of a circular opening of `radius_mm` seen at distance `z_mm` in a medium of index n."""
    x = 2 * np.pi * n * radius_mm / z_mm * (r_mm * 1e-3) / (wl_nm * 1e-9)
    with np.errstate(invalid="ignore", divide="ignore"):
        out = (2 * scipy.special.j1(x) / x) ** 2
    return np.where(x == 0, 1.0, out)


def slit_pattern(r_mm, wl_nm, n, width_mm, z_mm):
    return np.sinc(width_mm * 1e-3 * n / (wl_nm * 1e-9) * r_mm / z_mm) ** 2


def edge_pattern(y_mm, wl_nm, n, z_mm):
    """Fresnel diffraction at a straight edge at y = 0 (shadow below)."""
    u = np.sqrt(2 * n / (wl_nm * 1e-9) / (z_mm * 1e-3)) * y_mm * 1e-3
    S, C = scipy.special.fresnel(u)
    return 0.5 * ((S + 0.5) ** 2 + (C + 0.5) ** 2)


def rms_width(r, profile):
    return np.sqrt(np.average(r ** 2, weights=profile))


# ---- set-ups (tests/hurb_geometry.py) --------------------------------------------------------------------------------------
def tracer(half, z_end, n, use_hurb, hurb_factor):
    RT = ot.Raytracer(outline=[-half, half, -half, half, -6, z_end + 10], use_hurb=use_hurb,
                      n0=ot.RefractionIndex("Constant", n), seed=1)
    if hurb_factor is not None:
        RT.HURB_FACTOR = hurb_factor
    return RT


def airy_detector_side(wl, n, ri, zd, dim_ext_fact):
    """Six first-zero radii of the Airy pattern times the factor."""
    return 1.22 / (2 * np.pi / (wl * 1e-9) * n * ri / zd / np.pi) * 1e3 * 6 * dim_ext_fact


def cross_profile(RT, N_px):
    """Irradiance along the two image axes, averaged and normalised; pixel centres."""
    irr = RT.detector_image().get("Irradiance", N_px)
    edges, along_y = irr.profile(x=0)
    edges, along_x = irr.profile(y=0)
    prof = 0.5 * (along_y[0] + along_x[0])
    return edges[:-1] + 0.5 * (edges[1] - edges[0]), prof / prof.max()


def circular_opening(n, ri, wl, zd, N, N_px, dim_ext_fact, lens, use_hurb=True, hurb_factor=None):
    """Collimated beam of radius ri through a ring aperture of inner radius ri; `lens`: the aperture sits 1 um in front of an
    ideal lens of focal length zd (the pattern in its focal plane), else free propagation over zd.  -> r, simulated, theory"""
    with ot.global_options.no_warnings():
        RT = tracer(15, zd, n, use_hurb, hurb_factor)
        RT.add(ot.RaySource(ot.CircularSurface(r=ri), s=[0, 0, 1], pos=[0, 0, -5],
                            spectrum=ot.LightSpectrum("Monochromatic", wl=wl)))
        if lens:
            RT.add(ot.Aperture(ot.RingSurface(r=ri + 1, ri=ri), pos=[0, 0, -0.001]))
            RT.add(ot.IdealLens(ri + 1, 1000 / zd, pos=[0, 0, 0]))
        else:
            RT.add(ot.Aperture(ot.RingSurface(r=ri + 5, ri=ri), pos=[0, 0, 0]))
        side = airy_detector_side(wl, n, ri, zd, dim_ext_fact)
        RT.add(ot.Detector(ot.RectangularSurface(dim=[side, side]), pos=[0, 0, zd]))
        RT.trace(N)
        r, prof = cross_profile(RT, N_px)
    return r, prof, airy_pattern(r, wl, n, ri, zd)


def slit(n, d1, d2, wl, zd, N, N_px, angle, dim_ext_fact, hurb_factor=None):
    """Rectangular opening d1 x d2, turned by `angle` in its plane; profiles along the two turned axes.
    -> r, simulated along d1, simulated along d2, theory along d1, theory along d2"""
    half = 5 / (min(d1, d2) * 1e-3 * n / (wl * 1e-9) / zd) * dim_ext_fact
    with ot.global_options.no_warnings():
        RT = tracer(half, zd, n, True, hurb_factor)
        RS = ot.RaySource(ot.RectangularSurface(dim=[d1, d2]), s=[0, 0, 1], pos=[0, 0, -5],
                          spectrum=ot.LightSpectrum("Monochromatic", wl=wl))
        RS.rotate(angle)
        RT.add(RS)
        ap = ot.Aperture(ot.SlitSurface(dim=[d1 + 2, d2 + 2], dimi=[d1, d2]), pos=[0, 0, 0])
        ap.rotate(angle)
        RT.add(ap)
        RT.add(ot.Detector(ot.RectangularSurface(dim=[half, half]), pos=[0, 0, zd]))
        RT.trace(N)
        img = RT.detector_image()
        irr = img.get("Irradiance", N_px)
    xs = np.linspace(irr.extent[0], irr.extent[1], N_px)
    ys = np.linspace(irr.extent[2], irr.extent[3], N_px)
    spline = scipy.interpolate.RectBivariateSpline(ys, xs, irr.data, kx=3, ky=3)
    r = np.linspace(img.extent[0], img.extent[1], N_px)
    a = np.deg2rad(angle)
    cut1 = spline(r * np.sin(a), r * np.cos(a), grid=False)
    cut2 = spline(r * np.sin(a + np.pi / 2), r * np.cos(a + np.pi / 2), grid=False)
    return r, cut1 / cut1.max(), cut2 / cut2.max(), slit_pattern(r, wl, n, d1, zd), slit_pattern(r, wl, n, d2, zd)


def straight_edge(n, wl, zd, N, N_px, dim_ext_fact):
    """One edge of a very large rectangular opening at y = 0, lit from above the edge.  -> y, simulated, theory"""
    side = dim_ext_fact
    with ot.global_options.no_warnings():
        RT = tracer(4 * side, zd, n, True, 1)
        RT.add(ot.RaySource(ot.RectangularSurface(dim=[side / 2, side / 2]), s=[0, 0, 1], pos=[0, side / 4, -1],
                            spectrum=ot.LightSpectrum("Monochromatic", wl=wl)))
        inner = 4 * side - 0.4
        RT.add(ot.Aperture(ot.SlitSurface(dim=[4 * side, 4 * side], dimi=[inner, inner]), pos=[0, inner / 2, 0]))
        RT.add(ot.Detector(ot.RectangularSurface(dim=[side, side]), pos=[0, 0, zd]))
        RT.trace(N)
        irr = RT.detector_image().get("Irradiance", N_px)
    prof = irr.data.mean(axis=1)
    prof = prof / prof[4 * (len(prof) // 5):].mean()  # the plateau far from the edge
    y = np.linspace(irr.extent[2], irr.extent[3], irr.shape[0])
    return y, prof, edge_pattern(y, wl, n, zd)


# ---- the reference's tests ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,ri,wl,zd", [(1, 0.02, 550, 20), (1.33, 0.012, 380, 30), (1.5, 0.005, 780, 23), (1.1, 0.01, 480, 20)])
def test_pinhole_profile(n, ri, wl, zd):
    """tests/test_tracer_hurb.py:52-67: index, diameter, wavelength and distance varied."""
    r, sim, theory = circular_opening(n, ri, wl, zd, N=2_000_000, N_px=315, dim_ext_fact=3, lens=False, hurb_factor=1)
    assert rms_width(r, sim) / rms_width(r, theory) == pytest.approx(0.95, abs=0.04)


@pytest.mark.parametrize("n,ri,wl,zd", [(1, 1, 550, 20), (1.33, 3, 380, 30), (1.5, 5, 780, 23), (1.1, 2.7, 480, 20)])
def test_pupil_in_front_of_an_ideal_lens(n, ri, wl, zd):
    """tests/test_tracer_hurb.py:115-130: the Airy pattern in the focal plane."""
    r, sim, theory = circular_opening(n, ri, wl, zd, N=1_000_000, N_px=315, dim_ext_fact=3, lens=True, hurb_factor=1)
    assert rms_width(r, sim) / rms_width(r, theory) == pytest.approx(0.95, abs=0.04)


@pytest.mark.parametrize("n,d1,d2,wl,zd,angle", [(1, 0.02, 0.1, 550, 20, 0), (1.33, 0.012, 0.05, 380, 30, 10),
                                                 (1.5, 0.005, 0.005, 780, 23., -30), (1.1, 0.01, 0.1, 480, 20, 45)])
def test_slit_profiles(n, d1, d2, wl, zd, angle):
    """tests/test_tracer_hurb.py:95-113: aspect ratios, orientations, media, wavelengths, distances."""
    r, s1, s2, t1, t2 = slit(n, d1, d2, wl, zd, N=5_000_000, N_px=945, angle=angle, dim_ext_fact=5, hurb_factor=1)
    assert rms_width(r, s1) / rms_width(r, t1) == pytest.approx(1.11, abs=0.05)
    assert rms_width(r, s2) / rms_width(r, t2) == pytest.approx(1.11, abs=0.09)


@pytest.mark.parametrize("n,wl,zd", [(1, 550, 20), (1.33, 380, 30), (1.5, 780, 23), (1.1, 480, 20)])
def test_edge_profile(n, wl, zd):
    """tests/test_tracer_hurb.py:69-93: the plateau against the theory curve without its fringes, the flank in amplitude."""
    y, sim, theory = straight_edge(n, wl, zd, N=3_000_000, N_px=945, dim_ext_fact=2.5)
    flank_end = int(np.argmax(theory > 1.2))
    smooth = scipy.ndimage.gaussian_filter1d(theory, sigma=10)
    top = smooth[flank_end:-2] - sim[flank_end:-2]
    assert np.sqrt(np.mean(top ** 2)) < 0.02
    assert np.sqrt(np.mean((np.sqrt(theory[:flank_end]) - np.sqrt(sim[:flank_end])) ** 2)) < 0.015


def test_hurb_switch():
    """tests/test_tracer_hurb.py:132-147: without HURB an ideal lens gives an ideal focus; off by default; type-checked."""
    r, sim, _ = circular_opening(1.1, 2, 550, 20, N=100_000, N_px=945, dim_ext_fact=3, lens=True, use_hurb=False)
    assert rms_width(r, sim) == pytest.approx(0.0, abs=1e-10)
    assert not ot.Raytracer([-1, 1, -1, 1, -1, 1]).use_hurb
    with pytest.raises(TypeError):
        ot.Raytracer([-1, 1, -1, 1, -1, 1], use_hurb=[2, 3])


def test_hurb_factor_scales_the_width():
    """tests/test_tracer_hurb.py:149-165: for a slit the RMS width grows with the square root of HURB_FACTOR."""
    factors = np.array([1, np.sqrt(2), 2, 3])
    widths = []
    for f in factors:
        r, s1, _, _, _ = slit(1.1, 0.05, 0.50, 550, 20, N=1_000_000, N_px=945, angle=0, dim_ext_fact=6, hurb_factor=float(f))
        widths.append(rms_width(r, s1))
    scaled = np.array(widths) / np.sqrt(factors)
    assert np.std(scaled / scaled.mean()) < 0.05


def test_hurb_masks():
    """tests/test_tracer_hurb.py:19-50: rays a filter has taken and rays outside the opening pass the HURB step untouched."""
    with ot.global_options.no_warnings():
        RT = ot.Raytracer(outline=[-15, 15, -15, 15, -6, 10], use_hurb=True)
        RT.add(ot.RaySource(ot.CircularSurface(r=3), s=[0, 0, 1], pos=[0, 0, -5], spectrum=ot.presets.light_spectrum.d65))
        RT.add(ot.Filter(ot.CircularSurface(r=5), pos=[0, 0, -1],
                         spectrum=ot.TransmissionSpectrum("Rectangle", wl0=500, wl1=650, val=1)))
        RT.add(ot.Aperture(ot.RingSurface(r=5, ri=2.9), pos=[0, 0, 0]))
        RT.trace(200_000)
    alive = [np.count_nonzero(RT.rays.w_list[:, k]) for k in range(3)]
    assert alive[0] > alive[1] > alive[2] > 0
    assert np.isfinite(RT.rays.p_list).all()


def test_hurb_in_snapshot():
    """tests/test_tracer_hurb.py:167-187: use_hurb and HURB_FACTOR are trace settings of the snapshot."""
    with ot.global_options.no_warnings():
        RT = scenes.mixed_geometry(ot)
    snap = RT.tracing_snapshot()
    RT.use_hurb = True
    snap2 = RT.tracing_snapshot()
    diff = RT.compare_property_snapshot(snap, snap2)
    assert diff["Any"] and diff["TraceSettings"]
    RT.HURB_FACTOR = 1.389
    diff2 = RT.compare_property_snapshot(snap2, RT.tracing_snapshot())
    assert diff2["Any"] and diff2["TraceSettings"]


def test_hurb_needs_a_flat_ring_or_slit_aperture():
    """tests/test_tracer_hurb.py:189-203: a spherical aperture surface under HURB: geometry error, nothing traced."""
    with ot.global_options.no_warnings():
        RT = ot.Raytracer([-5, 5, -5, 5, 0, 10], use_hurb=True)
        RT.add(ot.RaySource(ot.Point(), pos=[0, 0, 0], s=[0, 0, 1]))
        RT.add(ot.Aperture(ot.SphericalSurface(r=3, R=12), pos=[0, 0, 2]))
        RT.trace(10_000)
    assert RT.geometry_error


def test_polarisation_follows_the_bent_directions():
    """tests/test_tracer_hurb.py:205-228: behind a pinhole the polarisation vectors are orthogonal to the NEW directions."""
    with ot.global_options.no_warnings():
        RT = ot.Raytracer([-5, 5, -5, 5, 0, 10], use_hurb=True)
        RT.add(ot.RaySource(ot.Point(), pos=[0, 0, 0], s=[0, 0, 1], div_angle=0.001))
        RT.add(ot.Aperture(ot.RingSurface(r=3, ri=0.001), pos=[0, 0, 1]))
        RT.trace(1_000_000)
        s = RT.rays.direction_vectors()[:, 1]
        pol = RT.rays.pol_list[:, 1]
    alive = RT.rays.w_list[:, 1] > 0
    assert alive.sum() > 1000
    angle = np.rad2deg(np.arccos(np.clip((s[alive] * pol[alive]).sum(axis=1), -1, 1)))
    assert np.std(angle) < 1e-6
