"""Physics / property tests of the GPU path, after the reference's own test strategy for its unseeded RNG
(SURVEY.md section 4): counters equal N for forced events (tests/test_tracer_special.py:570-627), Fresnel /
Brewster transmission values (:629-699), focal lengths (tests/test_tracer.py:295-347), ideal-lens imaging,
iterative rendering, HURB statistics."""
import numpy as np
import pytest

import optrace_amd as ot
import scenes

pytestmark = pytest.mark.gpu

MONO = dict(spectrum=ot.LightSpectrum("Monochromatic", wl=555.))


def test_absorb_missing_counts_all_rays():
    RT = ot.Raytracer(outline=[-3, 3, -3, 3, -10, 50], seed=1)
    RT.add(ot.RaySource(ot.CircularSurface(r=2), divergence="None", pos=[0, 0, -3], **MONO))
    surf = ot.CircularSurface(r=1e-6)
    RT.add(ot.Lens(surf, surf, n=ot.RefractionIndex("Constant", n=1.5), pos=[0, 0, 0], d=0.1))
    N = 10000
    with ot.global_options.no_warnings():
        RT.trace(N)
    assert abs(1 - RT._msgs[RT.INFOS.ABSORB_MISSING, 1] / N) < 1e-3
    assert np.all(RT.rays.p_list[:, -1, 2] < RT.outline[5] - 1)


def test_total_internal_reflection_counts_all_rays():
    RT = ot.Raytracer(outline=[-10, 10, -10, 10, -10, 50], n0=ot.RefractionIndex("Constant", 100), seed=1)
    RT.add(ot.RaySource(ot.CircularSurface(r=2), divergence="None", pos=[0, 0, -3], s=[0, 0.1, 0.99], **MONO))
    RT.add(ot.Lens(ot.CircularSurface(r=10), ot.CircularSurface(r=10), n=ot.RefractionIndex("Constant", n=1.5),
                   pos=[0, 0, 0], d=0.1))
    N = 10000
    with ot.global_options.no_warnings():
        RT.trace(N)
    assert RT._msgs[RT.INFOS.TIR, 0] == N
    assert np.all(RT.rays.w_list[:, 1] == 0) and np.all(np.isnan(RT.rays.s0_list))  # s' is NaN like the reference


def test_outline_intersection_counts_almost_all_rays():
    RT = ot.Raytracer(outline=[-3, 3, -3, 3, -10, 5000], seed=1)
    RT.add(ot.RaySource(ot.Point(), divergence="Isotropic", div_angle=80, pos=[0, 0, -3], **MONO))
    N = 10000
    with ot.global_options.no_warnings():
        RT.trace(N)
    assert abs(1 - RT._msgs[RT.INFOS.OUTLINE_INTERSECTION, 0] / N) < 1e-3
    p = RT.rays.p_list[:, 1]
    on_wall = (np.abs(np.abs(p[:, 0]) - 3) < 1e-9) | (np.abs(np.abs(p[:, 1]) - 3) < 1e-9) | (np.abs(p[:, 2] - 5000) < 1e-6)
    assert np.all(on_wall)


def test_brewster_and_fresnel_transmission():
    RT = ot.Raytracer(outline=[-5, 5, -5, 5, -10, 10], seed=4)
    n = ot.RefractionIndex("Constant", n=1.55)
    b_ang = np.arctan(1.55 / 1)
    s = [0, np.sin(b_ang), np.cos(b_ang)]
    spectrum = ot.LightSpectrum("Monochromatic", wl=550.)
    RSS = ot.CircularSurface(r=0.05)
    RT.add(ot.RaySource(RSS, divergence="None", spectrum=spectrum, pos=[2, -2.5, -2], s=s, polarization="x", power=1))
    RT.add(ot.RaySource(RSS, divergence="None", spectrum=spectrum, pos=[0, -2.5, -2], s=s, polarization="y", power=1))
    RT.add(ot.RaySource(RSS, divergence="None", spectrum=spectrum, pos=[-2, -2.5, -2], s=s, polarization="Uniform", power=10))
    rect = ot.RectangularSurface(dim=[10, 10])
    RT.add(ot.Lens(rect, rect, de=0.5, pos=[0, 0, 0], n=n, n2=n))
    with ot.global_options.no_warnings():
        RT.trace(120000)
    B = RT.rays.B_list
    w, pol = RT.rays.w_list, RT.rays.pol_list
    w0 = w[B[2], 0]
    assert np.allclose(w[B[0]:B[1], 1] / w0, 0.8301, atol=1e-3)       # s-polarised at the Brewster angle
    assert np.allclose(w[B[1]:B[2], 1] / w0, 1.0000, atol=1e-3)       # p-polarised: no reflection
    assert abs(np.mean(w[B[2]:, 1]) / w0 - 0.915) < 1e-3              # unpolarised
    assert np.allclose(pol[B[0]:B[1], 0, 1] ** 2 + pol[B[0]:B[1], 0, 2] ** 2, 0, atol=1e-5)
    assert np.allclose(pol[B[1]:B[2], 1, 1] ** 2 + pol[B[1]:B[2], 1, 2] ** 2, 1, atol=1e-5)
    d = np.diff(RT.rays.p_list[:, :3], axis=1)
    sdir = d / np.linalg.norm(d, axis=2)[:, :, None]
    for sec in (0, 1):
        pp = pol[:, sec].astype(np.float64)
        assert np.allclose((pp ** 2).sum(axis=1), 1, atol=1e-4)
        assert np.allclose(np.abs((pp * sdir[:, sec]).sum(axis=1)), 0, atol=1e-4)  # pol perpendicular to s


@pytest.mark.parametrize("R1,R2,n", [(8., -8., 1.5), (12., -30., 1.7), (-15., 9.5, 1.45)])
def test_paraxial_focus_matches_lensmaker(R1, R2, n):
    d = 1.0
    RT = ot.Raytracer(outline=[-5, 5, -5, 5, -10, 400], seed=2, no_pol=True)
    RT.add(ot.RaySource(ot.CircularSurface(r=0.05), divergence="None", pos=[0, 0, -5], **MONO))
    L = ot.Lens(ot.SphericalSurface(r=2, R=R1), ot.SphericalSurface(r=2, R=R2), n=ot.RefractionIndex("Constant", n=n),
                pos=[0, 0, 0], d=d)
    RT.add(L)
    with ot.global_options.no_warnings():
        RT.trace(20000)
    assert not RT.geometry_error
    inv_f = (n - 1) * (1 / R1 - 1 / R2 + (n - 1) * d / (n * R1 * R2))
    f = 1 / inv_f
    bfl = f * (1 - (n - 1) * d / (n * R1))  # back focal distance from the back vertex
    z_focus = L.back.pos[2] + bfl
    p, s = RT.rays.p_list[:, 2], RT.rays.s0_list
    if f > 0:
        t = -(p[:, 0] * s[:, 0] + p[:, 1] * s[:, 1]) / (s[:, 0] ** 2 + s[:, 1] ** 2 + 1e-300)
        z_cross = p[:, 2] + s[:, 2] * t
        sel = np.hypot(p[:, 0], p[:, 1]) > 1e-3
        assert abs(np.median(z_cross[sel]) - z_focus) < 2e-3 * abs(f)
    else:  # diverging: virtual focus in front of the lens
        t = -(p[:, 0] * s[:, 0] + p[:, 1] * s[:, 1]) / (s[:, 0] ** 2 + s[:, 1] ** 2 + 1e-300)
        sel = np.hypot(p[:, 0], p[:, 1]) > 1e-3
        assert abs(np.median((p[:, 2] + s[:, 2] * t)[sel]) - z_focus) < 2e-3 * abs(f)


def test_ideal_lens_images_point_to_point():
    D = 20.0  # f = 50 mm
    RT = ot.Raytracer(outline=[-20, 20, -20, 20, -110, 120], seed=5)
    RT.add(ot.RaySource(ot.Point(), divergence="Isotropic", div_angle=4, pos=[1.0, -0.5, -100], **MONO))
    RT.add(ot.IdealLens(r=15, D=D, pos=[0, 0, 0]))
    RT.add(ot.Detector(ot.RectangularSurface(dim=[10, 10]), pos=[0, 0, 100]))  # 1/f = 1/g + 1/b -> b = 100
    with ot.global_options.no_warnings():
        RT.trace(50000)
        ph, hw, wl, ext, proj, ill = RT._hit_detector("x", 0, None, None, None)
    n = hw.shape[0]
    ph = ph.cpu().numpy().reshape(3, n).T[hw.cpu().numpy() > 0]
    assert ph.shape[0] == 50000
    assert np.allclose(ph[:, 0], -1.0, atol=1e-8) and np.allclose(ph[:, 1], 0.5, atol=1e-8)  # magnification -1


def test_iterative_render_accumulates_chunks():
    with ot.global_options.no_warnings():
        import scenes
        RT = scenes.c1_single_lens(ot, seed=9)
        RT.ITER_RAYS_STEP = 200000
        imgs = RT.iterative_render(650000, extent=[-2, 2, -2, 2])
        assert len(imgs) == 1 and imgs[0]._data.shape == (945, 945, 4)
        RT2 = scenes.c1_single_lens(ot, seed=9)
        RT2.trace(650000)
        ref = RT2.detector_image(extent=[-2, 2, -2, 2])
    # same source power, independent ray sets: total power equal within sampling noise of the edge losses
    assert abs(imgs[0].power() - ref.power()) < 2e-3 * ref.power()
    assert RT._msgs.shape == (5, 4) and RT._msgs.sum() > 0
    # two positions at once
    with ot.global_options.no_warnings():
        imgs = RT.iterative_render(400000, pos=[[0, 0, 15], [0, 0, 25]], extent=[-3, 3, -3, 3])
    assert len(imgs) == 2 and imgs[0].power() > 0 and imgs[1].power() > 0
    # equal-sized chunks are binned straight into one histogram and scaled once: same power as a single trace,
    # and the same at both planes (nothing is lost between them inside this extent)
    # at z = 15 the whole beam is inside the extent
    assert abs(imgs[0].power() - ref.power()) < 3e-3 * ref.power()
    # the chunks of a seeded tracer are different ray sets: two chunks do not just repeat the first one
    with ot.global_options.no_warnings():
        one = RT.iterative_render(200000, pos=[0, 0, 15], extent=[-3, 3, -3, 3])[0]
        two = RT.iterative_render(400000, pos=[0, 0, 15], extent=[-3, 3, -3, 3])[0]
    assert np.abs(two._data[:, :, 3] - one._data[:, :, 3]).sum() > 0.05 * one._data[:, :, 3].sum()
    assert abs(one.power() - two.power()) < 3e-3 * two.power()


def test_hurb_spread_matches_uncertainty_formula():
    """Rays through the middle of a slit are bent by tan(theta) ~ N(0, HURB_FACTOR / (2 * dist * k))
    (raytracer.py:463-469); device Philox normals must reproduce that spread."""
    wl, half = 550., 0.025
    RT = ot.Raytracer(outline=[-3, 3, -3, 3, -5, 40], use_hurb=True, seed=21, no_pol=True)
    RT.add(ot.RaySource(ot.RectangularSurface(dim=[1e-6, 1e-6]), divergence="None", s=[0, 0, 1], pos=[0, 0, -4],
                        spectrum=ot.LightSpectrum("Monochromatic", wl=wl)))
    RT.add(ot.Aperture(ot.SlitSurface(dim=[2.5, 2.5], dimi=[2 * half, 2.0]), pos=[0, 0, 0]))
    N = 200000
    with ot.global_options.no_warnings():
        RT.trace(N)
    s = RT.rays.s0_list
    k = 2 * np.pi * 1.0 / (np.float32(wl) * np.float32(1e-9))
    sig_b = RT.HURB_FACTOR / (2 * half * 1e-3 * k)   # across the slit (x)
    sig_a = RT.HURB_FACTOR / (2 * 1.0 * 1e-3 * k)    # along the slit (y)
    tx, ty = s[:, 0] / s[:, 2], s[:, 1] / s[:, 2]
    assert abs(np.std(tx) / sig_b - 1) < 0.01 and abs(np.std(ty) / sig_a - 1) < 0.01
    assert abs(np.mean(tx)) < 0.01 * sig_b
    # normality: kurtosis of a Gaussian
    assert abs(np.mean(tx ** 4) / np.std(tx) ** 4 - 3) < 0.05
    assert RT._msgs[RT.INFOS.HURB_NEG_DIR].sum() == 0


def test_sharded_trace_single_process_equals_plain():
    """optrace_amd.distributed with world size 1 = plain trace + detector image."""
    from optrace_amd import distributed as D
    import scenes
    with ot.global_options.no_warnings():
        RT = scenes.c1_single_lens(ot)
        img = D.sharded_detector_image(RT, 100000, extent=[-2, 2, -2, 2], base_seed=3)
        RT2 = scenes.c1_single_lens(ot, seed=3)
        RT2.trace(100000)
        ref = RT2.detector_image(extent=[-2, 2, -2, 2])
    assert np.allclose(img._data, ref._data, rtol=1e-12, atol=1e-18)


def test_refraction_index_below_one_raises_at_trace():
    RT = ot.Raytracer(outline=[-5, 5, -5, 5, -5, 20], seed=1)
    RT.add(ot.RaySource(ot.Point(), pos=[0, 0, 0], spectrum=ot.LightSpectrum("Rectangle", wl0=400., wl1=700.)))
    bad = ot.RefractionIndex("Cauchy", coeff=[0.9, 0.01, 0, 0])  # n ~ 0.93 < 1 in the visible range
    RT.add(ot.Lens(ot.SphericalSurface(r=2, R=8), ot.SphericalSurface(r=2, R=-8), n=bad, pos=[0, 0, 5], d=1.0))
    with pytest.raises(RuntimeError, match="Refraction index below 1"):
        RT.trace(1000)


def test_scene_is_recompiled_only_when_it_changes():
    import scenes
    with ot.global_options.no_warnings():
        RT = scenes.c1_single_lens(ot, seed=2)
        RT.trace(10000)
        h1, k1 = RT._scene_handle.value, RT._scene_key
        RT.trace(20000)
        assert RT._scene_handle.value == h1 and RT._scene_key == k1
        RT.lenses[0].move_to([0, 0, 1.0])
        RT.trace(10000)
        assert RT._scene_key != k1
        img1 = RT.detector_image()
        RT.detectors[0].move_to([0, 0, 25])     # detectors are not part of the traced scene
        img2 = RT.detector_image()
    assert img1.power() > 0 and img2.power() > 0
    RT.no_pol = True
    with pytest.raises(RuntimeError, match="retrace"):
        RT.detector_image()


def test_polarisation_stays_transverse_and_normalised():
    """After the reference's test_polarization (tests/test_tracer.py:1183-1196): in a system with flat, conic and
    spherical lenses, a filter, an aperture and an ideal lens the stored polarisation of every living section is
    perpendicular to the section's direction and has unit length."""
    with ot.global_options.no_warnings():
        RT = scenes.mixed_geometry(ot, seed=12)
        RT.trace(200_000)
    _, s, pol, _, _, _, _ = RT.rays.rays_by_mask(ret=[0, 1, 1, 0, 0, 0, 0])
    scal = np.sum(s * pol, axis=2)
    scal = scal[~np.isnan(scal)]
    assert scal.size > 1_000_000
    assert np.ptp(scal) < 4e-7                    # float32 storage of pol: 6e-8 per component
    polpol = np.sum(pol.astype(np.float64) ** 2, axis=2)
    polpol = polpol[~np.isnan(polpol)]
    assert np.allclose(polpol, 1, rtol=2e-5)


@pytest.mark.parametrize("spectrum", ["d65", "mono", "led_b1"])
def test_grayscale_image_source_is_emitted_linearly(spectrum):
    """After the reference's test_grayscale_image_source (tests/test_tracer.py:1198-1237): the irradiance a detector
    right behind a GrayscaleImage source sees is the (linearised) image, whatever the spectrum."""
    spec = {"d65": ot.presets.light_spectrum.d65, "mono": ot.LightSpectrum("Monochromatic", wl=540),
            "led_b1": ot.presets.light_spectrum.led_b1}[spectrum]
    X, Y = np.mgrid[-1:1:63j, -1:1:63j]
    img = ot.GrayscaleImage(np.sin(5 * X + Y) ** 2, [1, 1])
    with ot.global_options.no_warnings():
        RT = ot.Raytracer([-10, 10, -10, 10, -10, 200], seed=3)
        RT.add(ot.RaySource(img, divergence="None", pos=[0, 0, 0], spectrum=spec))
        RT.add(ot.Detector(ot.RectangularSurface(img.s), pos=[0, 0, 1]))
        RT.trace(5_000_000)
        got = RT.detector_image().get("Irradiance", 63)
    a = got.data / np.max(got.data)
    v = img.data
    lin = np.where(v <= 0.04045, v / 12.92, ((v + 0.055) / 1.055) ** 2.4)  # sRGB -> linear (color/srgb.py:30-47)
    lin = lin / np.max(lin)
    assert np.mean(np.abs(a - lin)) < 0.003


def test_hurb_backward_directions_are_absorbed():
    """After the reference's test_hurb_negative_sz (tests/test_tracer_hurb.py:230-250): steep rays bent at a pinhole can
    come out with s_z < 0; they are absorbed and counted, no stored direction points backwards."""
    with ot.global_options.no_warnings():
        RT = ot.Raytracer([-5, 5, -5, 5, 0, 10], use_hurb=True, seed=8)
        RT.add(ot.RaySource(ot.Point(), pos=[0, -5, 0], s=[0, 0, 1], orientation="Converging", conv_pos=[0, 0, 0.01]))
        RT.add(ot.Aperture(ot.RingSurface(r=3, ri=0.001), pos=[0, 0, 0.01]))
        RT.trace(100_000)
    assert RT._msgs[RT.INFOS.HURB_NEG_DIR, 1] > 0
    s = RT.rays.direction_vectors()
    assert not np.any(s[:, :, 2] < 0)
    # the absorbed rays carry no power behind the aperture, the others keep theirs
    w = RT.rays.w_list
    lost = RT._msgs[RT.INFOS.HURB_NEG_DIR, 1] + RT._msgs[RT.INFOS.ABSORB_MISSING, 1]
    assert np.count_nonzero(w[:, 1] == 0) == lost + np.count_nonzero(w[:, 0] == 0)


def test_hurb_tilted_beam_sees_the_projected_slit():
    """After the reference's test_hurb_aperture_projection (tests/test_tracer_hurb.py:252-289): a beam that meets a slit
    at an angle sees it narrower by cos(angle), so the spread of the bending angle grows with 1 / cos(angle)."""
    products = []
    for ang0 in (5, 40, 60, 80):
        ang = np.deg2rad(ang0)
        za = 5 / np.tan(ang)
        with ot.global_options.no_warnings():
            RT = ot.Raytracer([-5, 5, -5, 5, 0, 10 * za], use_hurb=True, seed=21)
            RT.add(ot.RaySource(ot.Point(), pos=[-5, 0, 0], orientation="Converging", conv_pos=[0, 0, za]))
            RT.add(ot.Aperture(ot.SlitSurface(dim=[2, 2], dimi=[0.03, 1.9]), pos=[0, 0, za]))
            RT.trace(1_000_000)
        s = RT.rays.direction_vectors(normalize=True)
        angs = np.arccos(np.clip(np.sum(s[:, 0] * s[:, 1], axis=1), -1, 1))
        products.append(np.rad2deg(np.nanstd(angs)) * np.cos(ang))
    products = np.array(products)
    assert np.std(products / products[0]) < 0.002


def test_sphere_projections_keep_their_promises():
    """After the reference's test_sphere_projections (tests/test_tracer.py:636-740): on a hemispherical detector around
    an isotropic point source the Equal-Area image is flat; parallel pencils 30 degrees apart land equally spaced in
    the Equidistant image; small cones stay round in the Stereographic image wherever they hit; and for parallel light
    the Orthographic image of the sphere equals the image on a disc."""
    R = 90
    with ot.global_options.no_warnings():
        RT = ot.Raytracer(outline=[-100, 100, -100, 100, -10, 100], seed=5)
        RT.add(ot.Detector(ot.SphericalSurface(r=(1 - 1e-10) * R, R=-R), pos=[0, 0, R]))
        RS0 = ot.RaySource(ot.Point(), divergence="Isotropic", div_2d=False, pos=[0, 0, 0], s=[0, 0, 1], div_angle=89)
        RT.add(RS0)
        RT.trace(2_000_000)
        Z = RT.detector_image(projection_method="Equal-Area").get("Irradiance", 63).data
        Y, X = np.mgrid[-32:32:63j, -32:32:63j]
        Z = Z[np.sqrt(X ** 2 + Y ** 2) < 29]
        assert np.std(Z / np.max(Z)) < 0.015
        RT.remove(RS0)

        small = ot.RectangularSurface(dim=[0.0001, 0.0001])
        for theta in (-89.99, -60, -30, 0.001, 30, 60, 89.99):
            RT.add(ot.RaySource(small, divergence="None", div_2d=False, pos=[0, 0, 0], s_sph=[theta, 0]))
        RT.trace(20_000)
        z = RT.detector_image(projection_method="Equidistant").get("Irradiance", 256).profile(y=0)[1][0]
        fact = z.shape[0] / 6
        zp = (z > 0).nonzero()[0]
        assert zp.shape[0] >= 7
        assert np.all(np.abs(zp / fact - np.round(zp / fact)) < 1 / fact * 1.1)
        for rs in RT.ray_sources.copy():
            RT.remove(rs)

        for phi in np.linspace(0, 360, 5):
            for theta in np.linspace(0, 1, 4) * 87:
                rs = ot.RaySource(ot.Point(), divergence="Isotropic", div_2d=False, pos=[0, 0, 0], s_sph=[theta, phi], div_angle=2)
                RT.add(rs)
                RT.trace(40_000)
                x0, x1, y0, y1 = RT.detector_image(projection_method="Stereographic").extent
                RT.remove(rs)
                assert abs((x1 - x0) / (y1 - y0) - 1) < 0.004

        RT.clear()
        RT.add(ot.RaySource(ot.CircularSurface(r=3), pos=[0, 0, 0], divergence="None", s=[0, 0, 1]))
        RT.add(ot.Detector(ot.CircularSurface(r=3.01), pos=[0, 0, 10]))
        RT.add(ot.Detector(ot.SphericalSurface(r=3.01, R=-10), pos=[0, 0, 10]))
        RT.trace(200_000)
        a = RT.detector_image(detector_index=0, extent=[-3, 3, -3, 3])
        b = RT.detector_image(detector_index=1, extent=[-3, 3, -3, 3], projection_method="Orthographic")
        np.testing.assert_allclose(b._data, a._data, rtol=0, atol=1e-12 * a._data.max())


def test_ideal_lenses_keep_polarisation_transverse():
    """After the reference's test_ideal_lens_polarization (tests/test_tracer.py:1163-1181): behind a converging and a
    diverging ideal lens the polarisation is still perpendicular to the direction and of unit length."""
    with ot.global_options.no_warnings():
        RT = ot.Raytracer(outline=[-10, 10, -10, 10, 0, 40], seed=14)
        RT.add(ot.RaySource(ot.RectangularSurface(dim=[4, 4]), divergence="None", s=[0, 0, 1], pos=[0, 0, 0]))
        RT.add(ot.IdealLens(r=5, D=120, pos=[0, 0, 12]))
        RT.add(ot.IdealLens(r=6, D=-50, pos=[0, 0, 24]))
        RT.trace(200_000)
    _, s, pol, _, _, _, _ = RT.rays.rays_by_mask(ret=[0, 1, 1, 0, 0, 0, 0])
    scal = np.sum(s * pol, axis=2)[:, :-1]
    assert np.ptp(scal) < 4e-7
    assert np.allclose(np.sum(pol.astype(np.float64) ** 2, axis=2)[:, :-1], 1, rtol=2e-5)
    # and the two lenses do what their powers say: the beam converges (1 / 120 mm), then diverges again
    assert np.all(s[:, 0, 2] == 1) and np.mean(s[:, 1, 2]) < 1 and np.mean(s[:, 2, 2]) < 1
