"""Edge cases of the hot path on the GPU: empty and ragged inputs (sizes around the 64-lane wavefront and the
256-thread workgroup), rays that are dead or degenerate on entry, detectors nothing reaches."""
import numpy as np
import pytest

import optrace_amd as ot
from optrace_amd.scene import CompiledScene

import oracle_bridge as ob
import scenes

pytestmark = pytest.mark.gpu


def test_leaf_operators_accept_empty_arrays():
    with ot.global_options.no_warnings():
        zoo = {**scenes.surface_zoo(ot), **scenes.surface_zoo2(ot)}
    e3, e1 = np.zeros((0, 3)), np.zeros(0)
    for name in ("circle", "sphere_pos", "asphere_a", "tilted", "data2d"):
        sf = zoo[name]
        ph, hit, ill = sf.find_hit(e3, e3)
        assert ph.shape == (0, 3) and hit.shape == (0,)
        assert sf.normals(e1, e1).shape == (0, 3) and sf.mask(e1, e1).shape == (0,) and sf.values(e1, e1).shape == (0,)
    assert ot.RefractionIndex("Abbe", n=1.6, V=40.)(np.zeros(0, dtype=np.float32)).shape == (0,)


@pytest.mark.parametrize("N", [1, 2, 63, 64, 65, 255, 256, 257, 1000])
def test_ragged_sizes_match_oracle(N):
    """On-device generation for N rays, then the same rays through the oracle: every lane of a partial wave /
    partial workgroup must do exactly what a full one does."""
    with ot.global_options.no_warnings():
        RT = scenes.double_gauss(ot, seed=100 + N)
        RT.trace(N)
    r = RT.rays
    assert r.p_list.shape == (N, 17, 3) and r.N_list.sum() == N
    sc = CompiledScene(RT)
    rays = ob.HostRays(N, sc.nt, False)
    p0 = r.p_list[:, 0]
    d = r.p_list[:, 1] - p0
    s0 = d / np.linalg.norm(d, axis=1)[:, None]
    rays.set_initial(p0, s0, r.pol_list[:, 0], r.w_list[:, 0], r.wl_list)
    msgs, st = ob.trace(sc.desc, rays, None)
    assert st == 0
    # directions are regenerated from stored positions (1e-16 off): masks and counters still have to agree
    assert np.array_equal(rays.w_list > 0, r.w_list > 0)
    assert np.array_equal(msgs, RT._msgs)
    assert np.allclose(rays.p_list, r.p_list, rtol=1e-9, atol=1e-7)  # sources sit 50 m away: 1e-16 in s is 1e-11 mm here


def test_rays_dead_or_degenerate_on_entry():
    """Injected bundle with zero-weight rays, rays parallel to the axis far outside every lens, and rays that
    start behind the first surface: nothing may crash, dead rays stay where they are."""
    with ot.global_options.no_warnings():
        RT = scenes.c1_single_lens(ot)
        N = 300
        rng = np.random.default_rng(2)
        p0 = np.zeros((N, 3))
        p0[:, :2] = rng.uniform(-0.5, 0.5, (N, 2))
        p0[:, 2] = -20.
        s0 = np.tile([0., 0., 1.], (N, 1))
        w0 = np.full(N, 1.0 / N, dtype=np.float32)
        w0[:50] = 0.0                        # dead on entry
        p0[50:100, 0] = 9.0                  # pass far outside the lens (r = 3), inside the outline
        p0[100:150, 2] = 5.0                 # start behind the lens
        pol0 = np.tile([1., 0., 0.], (N, 1))
        wl = np.full(N, 550., dtype=np.float32)
        RT.trace(N, _initial_rays=(p0, s0, pol0, w0, wl))
        r = RT.rays
        assert np.all(np.isfinite(r.p_list))
        assert np.all(r.w_list[:50] == 0) and np.all(r.p_list[:50] == p0[:50, None, :])
        # rays outside the lens are absorbed at the first surface they miss (raytracer.py:347-357)
        # (so are the rays that start behind it: find_hit reports no hit for them, surface.py:460-476)
        assert np.all(r.w_list[50:150, 1] == 0) and RT._msgs[RT.INFOS.ABSORB_MISSING, 1] == 100
        # same bundle through the oracle
        sc = CompiledScene(RT)
        rays = ob.HostRays(N, sc.nt, False)
        rays.set_initial(p0, s0, pol0, w0, wl)
        msgs, st = ob.trace(sc.desc, rays, None)
        assert st == 0 and np.array_equal(msgs, RT._msgs)
        assert np.array_equal(rays.w_list > 0, r.w_list > 0)
        assert np.allclose(rays.p_list, r.p_list, rtol=1e-12, atol=1e-12)


def test_detector_nothing_reaches():
    """A detector beside the beam: empty image, zero power, the detector's own position as degenerate extent."""
    with ot.global_options.no_warnings():
        RT = scenes.c1_single_lens(ot, seed=3)
        RT.add(ot.Detector(ot.RectangularSurface(dim=[0.5, 0.5]), pos=[8.5, 8.5, 30]))
        RT.trace(5000)
        img = RT.detector_image(detector_index=1)
        assert img.power() == 0.0 and not np.any(img._data)
        spec = RT.detector_spectrum(detector_index=1)
        assert not np.any(spec._vals)
        # selecting a source range that hits and one user extent that excludes everything
        img2 = RT.detector_image(detector_index=0, extent=[3.5, 3.9, 3.5, 3.9])
        assert img2.power() == 0.0


def test_many_sources_and_tiny_ranges():
    """40 sources, 1000 rays: 25 rays per range, several ranges inside every wavefront (per-lane generation path)."""
    with ot.global_options.no_warnings():
        RT = ot.Raytracer(outline=[-10, 10, -10, 10, -25, 60], seed=8)
        for i in range(40):
            RT.add(ot.RaySource(ot.Point(), divergence="Isotropic", div_angle=2 + 0.05 * i,
                                pos=[0.02 * (i - 20), 0, -20], power=1.0,
                                spectrum=ot.LightSpectrum("Monochromatic", wl=450. + 5 * i)))
        RT.add(ot.Lens(ot.SphericalSurface(r=3, R=8), ot.SphericalSurface(r=3, R=-8), de=0.1,
                       n=ot.RefractionIndex("Constant", n=1.5), pos=[0, 0, 0]))
        RT.trace(1000)
        r = RT.rays
        assert np.array_equal(r.N_list, np.full(40, 25))
        for i in range(40):
            sl = slice(r.B_list[i], r.B_list[i + 1])
            assert np.all(r.wl_list[sl] == np.float32(450. + 5 * i))
            assert np.allclose(r.p_list[sl, 0, 0], 0.02 * (i - 20), atol=1e-15)
        assert np.all(np.isfinite(r.p_list))


def test_more_than_64_sources():
    """100 sources: the range table moves from the kernel arguments to device memory (binary search per ray)."""
    with ot.global_options.no_warnings():
        RT = ot.Raytracer(outline=[-10, 10, -10, 10, -25, 60], seed=12)
        for i in range(100):
            RT.add(ot.RaySource(ot.Point(), divergence="Isotropic", div_angle=2, pos=[0.01 * (i - 50), 0, -20],
                                power=1.0 + (i % 3), spectrum=ot.LightSpectrum("Monochromatic", wl=400. + 3 * i)))
        RT.add(ot.Lens(ot.SphericalSurface(r=3, R=8), ot.SphericalSurface(r=3, R=-8), de=0.1,
                       n=ot.RefractionIndex("Constant", n=1.5), pos=[0, 0, 0]))
        RT.trace(100_000)
        r = RT.rays
        assert r.N_list.shape == (100,) and r.N_list.sum() == 100_000
        for i in (0, 1, 37, 63, 64, 65, 99):
            sl = slice(r.B_list[i], r.B_list[i + 1])
            assert np.all(r.wl_list[sl] == np.float32(400. + 3 * i))
            assert np.allclose(r.p_list[sl, 0, 0], 0.01 * (i - 50), atol=1e-15)
            assert abs(r.w_list[sl, 0].astype(np.float64).sum() - (1.0 + (i % 3)) / r.N_list[i] * r.N_list[i]) < 1e-4
        assert np.all(np.isfinite(r.p_list)) and np.all(r.w_list[:, -1] == 0)


def test_seeded_tracer_repeats_including_the_source_split():
    """Three sources of unequal power and a ray count that does not divide: the random remainder of the split
    (ray_storage.py:63-68) follows the tracer's seed as well."""
    def build(seed):
        RT = ot.Raytracer(outline=[-10, 10, -10, 10, -25, 60], seed=seed)
        for i, pw in enumerate((1.0, 2.5, 0.7)):
            RT.add(ot.RaySource(ot.CircularSurface(r=0.5), divergence="Lambertian", div_angle=3, pos=[i - 1, 0, -20],
                                power=pw, spectrum=ot.presets.light_spectrum.d65))
        RT.add(ot.Lens(ot.SphericalSurface(r=3, R=8), ot.SphericalSurface(r=3, R=-8), de=0.1,
                       n=ot.RefractionIndex("Constant", n=1.5), pos=[0, 0, 0]))
        return RT
    with ot.global_options.no_warnings():
        a, b, c = build(5), build(5), build(6)
        for RT in (a, b, c):
            RT.trace(10007)
    assert np.array_equal(a.rays.N_list, b.rays.N_list) and a.rays.N_list.sum() == 10007
    assert np.array_equal(a.rays.p_list, b.rays.p_list) and np.array_equal(a.rays.wl_list, b.rays.wl_list)
    assert not np.array_equal(a.rays.p_list[:100], c.rays.p_list[:100])


def test_backward_directions_raise_like_create_rays():
    """A cone wide enough to reach s_z <= 0 (ray_source.py:353): RuntimeError from trace() and from create_rays()."""
    with ot.global_options.no_warnings():
        RT = ot.Raytracer(outline=[-10, 10, -10, 10, -25, 60], seed=2)
        rs = ot.RaySource(ot.Point(), divergence="Isotropic", div_angle=80, pos=[0, 0, -20], s=[0.8, 0, 0.6])
        RT.add(rs)
        RT.add(ot.Lens(ot.SphericalSurface(r=3, R=8), ot.SphericalSurface(r=3, R=-8), de=0.1,
                       n=ot.RefractionIndex("Constant", n=1.5), pos=[0, 0, 0]))
        with pytest.raises(RuntimeError, match="positive z"):
            RT.trace(20000)
        with pytest.raises(RuntimeError, match="positive z"):
            rs.create_rays(20000)
        rs.div_angle = 20  # inside the forward half space again
        RT.trace(20000)
        assert RT._msgs[RT.INFOS.HURB_NEG_DIR, 0] == 0


def test_batched_detector_hits_equal_single_calls():
    """`ot_detector_hits_multi` (iterative_render's positions in one pass over the sections) gives, detector by
    detector, exactly what single calls give: positions inside the lens stack, behind it, before the source, a
    spherical detector with projection, a user extent and automatic extents."""
    import torch
    with ot.global_options.no_warnings():
        RT = scenes.asphere_scene(ot, seed=3)
        RT.trace(60_000)
        zs = [-5., 2., 11., 20., 30., 34., 39.]
        specs = [dict(detector_index=1, pos=[0.1 * k, -0.05 * k, z], extent=None if k % 2 else [-3, 3, -2.5, 2.5])
                 for k, z in enumerate(zs)]
        specs += [dict(detector_index=0, pos=[0, 0, 36.], projection_method=pm, extent=None)
                  for pm in ("Equidistant", "Orthographic")]
        single = [RT._hit_detectors("t", [sp])[0] for sp in specs]
        batch = RT._hit_detectors("t", specs)
        assert len(batch) == len(specs)
        hits = 0
        for a, b in zip(single, batch):
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])   # positions, weights
            np.testing.assert_array_equal(a[3], b[3])                     # extent
            assert a[4] == b[4] and a[5] == b[5] and a[6] == b[6]         # projection, ill count, description
            hits += int((a[1] > 0).sum())
        assert hits > 0
        # iterative_render through the batch equals detector images position by position (same seed, one chunk)
        RT2 = scenes.asphere_scene(ot, seed=3)
        pos = [[0, 0, z] for z in (20., 22., 24., 26., 28., 30., 31., 32., 33., 34.)]  # more than one group of 8
        imgs = RT2.iterative_render(60_000, detector_index=1, pos=pos, extent=[[-4, 4, -4, 4]] * len(pos))
        for p_, im in zip(pos, imgs):
            RT2.detectors[1].move_to(p_)
            ref = RT2.detector_image(detector_index=1, extent=[-4, 4, -4, 4])
            np.testing.assert_allclose(im._data, ref._data, rtol=1e-12, atol=1e-300)


def test_rays_by_mask_gathers_on_the_device():
    """RayStorage.rays_by_mask (ray_storage.py:235-293): the device gather returns exactly what indexing the host
    copies returns, for row masks, all sections or one section per ray, with and without polarisation."""
    for no_pol in (False, True):
        with ot.global_options.no_warnings():
            RT = scenes.double_gauss(ot, seed=9, no_pol=no_pol)
            RT.trace(20_000)
        r = RT.rays
        rng = np.random.default_rng(4)
        ch = rng.random(r.N) < 0.1
        ch2 = rng.integers(0, r.Nt, int(ch.sum()))
        assert not r._host  # nothing copied yet: the first calls take the device path
        dev_all = r.rays_by_mask(ch)
        dev_sec = r.rays_by_mask(ch, ch2, normalize=False)
        dev_len, dev_opt, dev_src = r.ray_lengths(ch, ch2), r.optical_lengths(ch), r.source_numbers()
        dev_ss = r.source_sections(1) + r.source_sections()
        assert set(r._host) <= {"pol"}  # still no host copy of the big lists (pol is a broadcast NaN with no_pol)
        for name in ("p", "s", "w", "n", "wl", "pol"):
            r._view(name)             # now the host copies exist and are used
        host_all = r.rays_by_mask(ch)
        host_sec = r.rays_by_mask(ch, ch2, normalize=False)
        for a, b in zip(dev_all + dev_sec, host_all + host_sec):
            assert (a is None) == (b is None)
            if a is not None:
                assert a.shape == b.shape and a.dtype == b.dtype
                np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(dev_len, r.ray_lengths(ch, ch2))
        np.testing.assert_array_equal(dev_opt, r.optical_lengths(ch))
        np.testing.assert_array_equal(dev_src, r.source_numbers())
        for a, b in zip(dev_ss, r.source_sections(1) + r.source_sections()):
            assert a.shape == b.shape and a.dtype == b.dtype
            np.testing.assert_array_equal(a, b)


def test_few_rays_every_entry_point():
    """Like the reference's test_few_rays_action (tests/test_tracer_special.py:419-457): 70, 3 and 1 rays, with and
    without everything being absorbed before the detector, through every consumer of a trace."""
    with ot.global_options.no_warnings():
        RT = scenes.mixed_geometry(ot, seed=2)
        ap = RT.apertures[0]
        for el in [*RT.apertures, *RT.filters]:
            RT.remove(el)
        for blocked in (False, True):
            for N in (70, 3, 1):
                RT.trace(N)
                assert RT.rays.N == N and not RT.geometry_error
                assert RT.source_image().power() >= 0
                img = RT.detector_image()
                assert np.all(np.isfinite(img._data)) and img.power() >= 0
                img1 = RT.detector_image(detector_index=1, projection_method="Stereographic")
                if blocked:  # detector 1 stands behind the stop (detector 0 in the source plane, before it)
                    assert img1.power() == 0
                for sp in (RT.source_spectrum(), RT.detector_spectrum()):
                    assert np.all(np.isfinite(sp._vals))
                assert len(RT.iterative_render(N)) == 1
                for fm in RT.focus_search_methods:
                    res, info = RT.focus_search(fm, z_start=30)
                    if info["N"] > 1:
                        assert RT.outline[4] <= res.x <= RT.outline[5]
                    else:  # placeholder result like the reference's (raytracer.py:1553-1556)
                        assert "x" not in res and np.all(np.isnan(info["z"]))
                r = RT.rays
                assert sum(a.shape[0] for a in (r.source_sections(0)[0], r.source_sections(1)[0])) == N
                assert r.source_sections()[0].shape == (N, 3)
                none = r.rays_by_mask(np.zeros(N, dtype=bool))
                assert none[0].shape[0] == 0 and none[4].shape[0] == 0
                every = r.rays_by_mask(np.ones(N, dtype=bool))
                assert every[0].shape == (N, r.Nt, 3) and every[3].shape == (N, r.Nt)
            # a stop right behind the sources: nothing reaches the detectors or the focus-search region
            RT.add(ap)
            RT.apertures[0].move_to(ap.pos + [0.2, 0.2, 0])
            RT.add(ot.Aperture(ot.CircularSurface(r=4.5), pos=[0, 0, 1]))


@pytest.mark.parametrize("pos0", [(5.789, 0.123, -45.6), (0, 16546.789789, -4654), (1e5, -1e6, 15)])
def test_offset_system_equality(pos0):
    """The same system shifted by a vector images the same way (reference: tests/test_tracer_special.py:61-124, whose
    tolerances these are): extent relative to the shift, side lengths and power of the detector image."""
    def build(p0):
        p0 = np.array(p0, dtype=float)
        x0, y0, z0 = p0
        RT = ot.Raytracer(outline=[-5 + x0, 5 + x0, -5 + y0, 5 + y0, -10 + z0, 50 + z0], seed=17)
        RT.add(ot.RaySource(ot.CircularSurface(r=0.2), spectrum=ot.LightSpectrum("Monochromatic", wl=555), divergence="None",
                            pos=p0 + [0, 0, -3]))
        glass = ot.RefractionIndex("Sellmeier1", coeff=[1.62153902, 0.0122241457, 0.256287842, 0.0595736775, 1.64447552, 147.468793])
        RT.add(ot.Lens(ot.SphericalSurface(r=3, R=50), ot.ConicSurface(r=3, R=-50, k=-1.5), n=glass, pos=p0 + [0, 0.01, 0]))
        RT.add(ot.Lens(ot.FunctionSurface1D(r=3, func=lambda r: r ** 2 / 50 + r ** 2 / 5000, parax_roc=25),
                       ot.CircularSurface(r=2), n=glass, pos=p0 + [0, 0.01, 10]))
        Y, X = np.mgrid[-1:1:100j, -1:1:100j]
        RT.add(ot.Lens(ot.DataSurface2D(data=3 - (X ** 2 + Y ** 2), r=4), ot.TiltedSurface(r=4, normal=[0, 0.01, 1]),
                       n=ot.RefractionIndex("Cauchy", coeff=[1.5, 0.004, 0, 0]), pos=p0 + [0, 0, 20]))
        RT.add(ot.Detector(ot.RectangularSurface(dim=[10, 10]), pos=p0 + [0., 0., 45]))
        RT.trace(100_000)
        assert not RT.geometry_error
        return RT.detector_image()

    with ot.global_options.no_warnings():
        ref, im = build((0, 0, 0)), build(pos0)
    shift = np.array(pos0[:2], dtype=float).repeat(2)
    assert ref.power() > 0.5
    np.testing.assert_allclose(ref.extent - im.extent + shift, 0, atol=0.001, rtol=0)
    assert abs((ref.extent[1] - ref.extent[0]) - (im.extent[1] - im.extent[0])) < 0.001
    assert abs((ref.extent[3] - ref.extent[2]) - (im.extent[3] - im.extent[2])) < 0.001
    assert abs(ref.power() - im.power()) < 5e-5


def test_render_entry_points_reject_what_the_reference_rejects():
    """After the reference's test_image_render_parameter (tests/test_tracer.py:919-953): error behaviour of the image
    and spectrum calls -- nothing traced, indices out of range, malformed extents, no detectors, no sources."""
    with ot.global_options.no_warnings():
        RT = scenes.mixed_geometry(ot, seed=1)
        for f in (RT.detector_image, RT.detector_spectrum, RT.source_image, RT.source_spectrum):
            with pytest.raises(RuntimeError):
                f()
        RT.trace(10_000)
        for bad in (3, -3):
            for f in (RT.detector_image, RT.detector_spectrum):
                with pytest.raises(IndexError):
                    f(detector_index=bad)
                with pytest.raises(IndexError):
                    f(source_index=bad)
            for f in (RT.source_image, RT.source_spectrum):
                with pytest.raises(IndexError):
                    f(source_index=bad)
        for f in (RT.detector_image, RT.detector_spectrum):
            with pytest.raises(ValueError):
                f(extent="abc")
            with pytest.raises(ValueError):
                f(extent=[1, 2, 1, np.inf])
        RT.detectors = []
        for f in (RT.detector_image, RT.detector_spectrum):
            with pytest.raises(RuntimeError):
                f()
        RT.ray_sources = []
        for f in (RT.source_image, RT.source_spectrum):
            with pytest.raises(RuntimeError):
                f()


def test_iterative_render_arguments():
    """After the reference's test_iterative_render (tests/test_tracer.py:956-1075): single values and per-position lists
    of positions, detector indices, extents, projections and limits; mismatched lists, missing detectors / sources and
    colliding geometry raise; chunks add up, also with an odd-sized last chunk; counters add up over the chunks."""
    with ot.global_options.no_warnings():
        RT = scenes.mixed_geometry(ot, seed=7)
        RT.ITER_RAYS_STEP = 40_000
        N = 4000
        dim = RT.iterative_render(N)
        assert len(dim) == 1 and dim[0].limit is None
        RT.iterative_render(N, pos=[0, 0, 13.3])
        ext2 = [0, *RT.detectors[0].extent[1:4]]
        assert np.all(RT.iterative_render(N, extent=ext2)[0].extent == ext2)
        RT.iterative_render(N, detector_index=1)
        assert RT.iterative_render(N, detector_index=1, projection_method="Stereographic")[0].projection == "Stereographic"
        assert RT.iterative_render(N, detector_index=0, limit=5)[0].limit == 5
        assert len(RT.iterative_render(N, pos=[[0, 0, 0], [0, 0, 5]])) == 2
        dim = RT.iterative_render(N, pos=[[0, 0, 0], [0, 0, 5]], detector_index=[0, 1])
        assert len(dim) == 2 and dim[0].projection != dim[1].projection
        dim = RT.iterative_render(10_000, pos=[[0, 0, 0], [0, 0, 5]], detector_index=1,
                                  projection_method=["Equidistant", "Equal-Area"])
        assert dim[0].projection != dim[1].projection
        dim = RT.iterative_render(10_000, pos=[[0, 0, 0], [0, 0, 5]], detector_index=0, limit=[10, 12])
        assert dim[0].limit != dim[1].limit

        for bad in (0, -10):
            with pytest.raises(ValueError):
                RT.iterative_render(bad)
        for kw in (dict(extent=[None, None]), dict(projection_method=["Equidistant", "Equal-Area"]),
                   dict(detector_index=[0, 1]), dict(detector_index=[0, 1], pos=[0, 0, 0]), dict(limit=[4, 1], pos=[0, 0, 0])):
            with pytest.raises(ValueError):
                RT.iterative_render(10_000, **kw)

        # whole chunks and an odd-sized last one: the power of the image does not depend on the chunking
        RT.detectors[0].move_to([0, 0, 0])
        one = RT.iterative_render(80_000)[0].power()
        odd = RT.iterative_render(80_100)[0].power()
        RT.ITER_RAYS_STEP = None
        whole = RT.iterative_render(80_000)[0].power()
        assert abs(one - whole) < 2e-2 * whole and abs(odd - whole) < 2e-2 * whole

        det_backup = RT.detectors.copy()
        RT.remove(RT.detectors)
        with pytest.raises(RuntimeError):
            RT.iterative_render(N)
        RT.add(det_backup)
        RT.remove(RT.ray_sources)
        with pytest.raises(RuntimeError):
            RT.iterative_render(N)

        RT = scenes.mixed_geometry(ot, seed=7)
        RT.lenses[0].move_to(RT.lenses[1].pos)  # collision
        with pytest.raises(RuntimeError):
            RT.iterative_render(1000)

        # counters are summed over the chunks: every ray misses the hair-thin lens in every chunk
        RT = ot.Raytracer(outline=[-3, 3, -3, 3, -10, 50], seed=7)
        RT.add(ot.RaySource(ot.CircularSurface(r=2), spectrum=ot.LightSpectrum("Monochromatic", wl=555), divergence="None",
                            pos=[0, 0, -3]))
        RT.add(ot.Lens(ot.CircularSurface(r=1e-6), ot.CircularSurface(r=1e-6), n=ot.RefractionIndex("Constant", n=1.5),
                       pos=[0, 0, 0], d=0.1))
        RT.add(ot.Detector(ot.RectangularSurface(dim=[2, 2]), pos=[0, 0, 10]))
        RT.ITER_RAYS_STEP = 10_000
        RT.iterative_render(30_000)
        assert RT._msgs[RT.INFOS.ABSORB_MISSING, 1] >= 30_000 - 3


def test_stale_rays_are_refused():
    """After the reference's test_source_detector_image_spectrum (tests/test_tracer.py:1067-1080): changing the
    geometry without retracing makes every consumer of the rays raise."""
    with ot.global_options.no_warnings():
        RT = scenes.mixed_geometry(ot, seed=1)
        RT.trace(10_000)
        RT.remove(RT.lenses[0])
        for f in (RT.detector_image, RT.detector_spectrum, RT.source_image, RT.source_spectrum):
            with pytest.raises(RuntimeError):
                f()
        with pytest.raises(RuntimeError):
            RT.focus_search(RT.focus_search_methods[0], z_start=30)
        RT.trace(10_000)
        RT.detector_image(limit=4, extent=[-0.01, 0.01, -0.01, 0.01])  # limit together with an extent only warns


def test_arrays_edited_in_place_between_traces_are_seen():
    """The unchanged-scene shortcut of `trace` counts assignments; the few small arrays that stay writeable (the outline,
    RaySource.s, conv_pos) can also be edited in place, which the reference picks up because it re-reads the object graph
    at every trace (raytracer.py:246-278).  Their bytes are therefore compared before every shortcut."""
    with ot.global_options.no_warnings():
        RT = ot.Raytracer(outline=[-5, 5, -5, 5, -10, 30], seed=3)
        RS = ot.RaySource(ot.Point(), divergence="None", s=[0, 0, 1], pos=[0, 0, -5],
                          spectrum=ot.LightSpectrum("Monochromatic", wl=550.))
        RT.add(RS)
        RT.add(ot.Detector(ot.RectangularSurface(dim=[4, 4]), pos=[0, 0, 20]))
        RT.trace(1000)
        RT.trace(1000)  # second call: the shortcut
        assert RT.rays.p_list[0, -1, 2] == 30 and abs(RT.rays.p_list[0, -1, 0]) < 1e-12
        o = RT.outline
        assert o.flags.writeable, "the case only exists while the outline can be edited in place"
        o[5] += 7.0
        RT.trace(1000)
        assert RT.rays.p_list[0, -1, 2] == 37, "the new outline ends the rays"
        RT.detector_image()  # rays are current: no 'retrace first'
        assert RS.s.flags.writeable
        RS.s[0], RS.s[2] = 0.1 / np.hypot(0.1, 1), 1 / np.hypot(0.1, 1)
        RT.trace(1000)
        x_end = RT.rays.p_list[0, -1, 0]
        assert abs(x_end - 0.1 * 42) < 1e-9, "the tilted direction is used"
        RT.detector_image()
        # and an edit after the last trace makes the rays stale, like any other change
        o[5] -= 1.0
        with pytest.raises(RuntimeError):
            RT.detector_image()


def test_meniscus_lenses_whose_surfaces_embrace_each_other():
    """After the reference's test_non_sequental_surface_extent (tests/test_tracer_special.py:366-417): near-hemispherical
    meniscus lenses, one surface reaching past the other in z -- no geometry error, a beam through the middle passes
    completely, a ring of rays aimed at the gap between the surfaces' edges is absorbed completely."""
    N = 100_000
    R1, R2 = 5, 10
    with ot.global_options.no_warnings():
        RT = ot.Raytracer(outline=[-15, 15, -15, 15, -20, 50], seed=10)
        RS = ot.RaySource(ot.CircularSurface(r=4), pos=[0, 0, -20])
        RT.add(RS)
        L = ot.Lens(ot.SphericalSurface(r=0.999 * R1, R=-R1), ot.SphericalSurface(r=0.999 * R2, R=-R2), pos=[0, 0, 0],
                    n=ot.RefractionIndex(), d=0.5)                                    # )): second embraces first
        RT.add(L)
        RT.add(ot.Filter(ot.CircularSurface(r=3), spectrum=ot.TransmissionSpectrum("Constant", val=1), pos=[0, 0, 20]))
        RT.trace(N)
        assert not RT.geometry_error and np.all(RT.rays.w_list[:, -2] > 0)

        RT.remove(L)
        L = ot.Lens(ot.SphericalSurface(r=0.999 * R2, R=R2), ot.SphericalSurface(r=0.999 * R1, R=R1), pos=[0, 0, 0],
                    n=ot.RefractionIndex(), d=0.5)                                    # ((: first embraces second
        RT.add(L)
        RT.trace(N)
        assert not RT.geometry_error and np.all(RT.rays.w_list[:, -2] > 0)

        RT.remove(RS)
        RT.add(ot.RaySource(ot.RingSurface(r=7, ri=6), pos=[0, 0, -20]))             # aimed at the edge gap
        RT.trace(N)
        assert not RT.geometry_error and np.all(RT.rays.w_list[:, -2] == 0)

        RT.remove(L)
        RT.add(ot.Lens(ot.SphericalSurface(r=0.999 * R1, R=-R1), ot.SphericalSurface(r=0.999 * R2, R=-R2), pos=[0, 0, 0],
                       n=ot.RefractionIndex(), d=0.5))
        RT.trace(N)
        assert not RT.geometry_error and np.all(RT.rays.w_list[:, -2] == 0)


def test_ray_storage_bookkeeping():
    """After the reference's test_ray_storage (tests/test_tracer.py:738-832): ray counts and powers per source for
    several power ratios, ray numbers and numbers of sources; source_sections and rays_by_mask shapes, omitted
    properties, unit directions; direction_vectors and source_numbers agree with rays_by_mask."""
    rng = np.random.default_rng(5)
    with ot.global_options.no_warnings():
        RT = scenes.mixed_geometry(ot, seed=9)
        RT.add(ot.RaySource(ot.Point(), spectrum=ot.LightSpectrum("Monochromatic", wl=550), pos=[0, 0, 0]))
        assert len(RT.ray_sources) == 3
        for _ in range(2):
            for powers in ((1, 1, 1), (2, 1, 1), (0.3456465, 4.57687168, np.pi / 2)):
                for N in (30000, 30001, 52657):
                    for rs, pw in zip(RT.ray_sources, powers):
                        rs.power = pw
                    RT.trace(N)
                    r = RT.rays
                    assert N == r.N
                    P_s = sum(rs.power for rs in RT.ray_sources)
                    assert abs(P_s - np.sum(r.w_list[:, 0].astype(np.float64))) < 10 / N
                    for Ni, rs in enumerate(RT.ray_sources):
                        assert r.source_sections(Ni)[0].shape[0] == r.N_list[Ni]
                        assert abs(r.N_list[Ni] / r.N - rs.power / P_s) < 10 / N
                    assert r.source_sections()[0].shape[0] == N
                    for t1, t2 in zip(r.rays_by_mask(), r.rays_by_mask(np.ones(N, dtype=bool))):
                        np.testing.assert_array_equal(t1, t2)
                    ch = rng.integers(0, 2, size=N).astype(bool)
                    N2 = int(np.count_nonzero(ch))
                    ch2 = rng.integers(0, r.Nt, size=N2)
                    for ch2li in (None, ch2):
                        for retli in (None, [1, 0, 1, 1, 1, 1, 1], [1, 1, 1, 1, 1, 0, 0]):
                            for normli in (False, True):
                                tup = r.rays_by_mask(ch, ch2li, retli, normli)
                                assert tup[3].shape[0] == N2 and tup[3].ndim == (1 if ch2li is not None else 2)
                                for k in (1, 5, 6):
                                    assert retli is None or retli[k] or tup[k] is None
                                if normli and (retli is None or retli[1]):
                                    mask = np.all(np.isfinite(tup[1]), axis=2 if ch2li is None else 1)
                                    assert np.allclose(np.sum(tup[1][mask] ** 2, axis=-1), 1, rtol=0, atol=1e-4)
            RT.remove(RT.ray_sources[-1])

        RT = scenes.mixed_geometry(ot, seed=9)
        RT.trace(100_000)
        for norm in (False, True):
            s1 = RT.rays.rays_by_mask(ret=[0, 1, 0, 0, 0, 0, 0], normalize=norm)[1]
            s2 = RT.rays.direction_vectors(norm)
            assert np.all((s1 == s2) | np.isnan(s1))
        assert np.all(RT.rays.rays_by_mask(ret=[0, 0, 0, 0, 0, 1, 0])[5] == RT.rays.source_numbers())


def test_optical_and_geometric_path_lengths():
    """After the reference's test_optical_lengths_ray_lengths (tests/test_tracer.py:835-886): axial rays through a slab,
    a clear filter and a stop: section lengths 1, 1.2, 2.8, 4 mm; optical lengths with n(lambda) inside the slab; masked
    calls pick the same numbers."""
    rng = np.random.default_rng(2)
    n = ot.RefractionIndex("Sellmeier1", coeff=[1.62153902, 0.0122241457, 0.256287842, 0.0595736775, 1.64447552, 147.468793])
    with ot.global_options.no_warnings():
        RT = ot.Raytracer(outline=[-10, 10, -10, 10, -2, 10], seed=4)
        RT.add(ot.RaySource(ot.Point(), pos=[0, 0, 0]))
        RT.add(ot.Lens(ot.CircularSurface(r=3), ot.CircularSurface(r=3), n=n, pos=[0, 0, 1], d1=0, d2=1.2))
        RT.add(ot.Filter(ot.CircularSurface(r=3), spectrum=ot.TransmissionSpectrum("Constant", val=1), pos=[0, 0, 5]))
        RT.add(ot.Aperture(ot.CircularSurface(r=3), pos=[0, 0, 9]))
        RT.trace(1000)
        r = RT.rays
        ol = r.optical_lengths()
        want = ol.copy()
        want[:, 0], want[:, 1], want[:, 2], want[:, 3] = 1, n(r.wl_list) * 1.2, 2.8, 4
        assert np.allclose(ol - want, 0, atol=1e-9, rtol=0)
        m1 = rng.choice(np.array([0, 1], dtype=bool), size=r.N)
        m2 = rng.choice(np.arange(r.Nt), size=r.N, replace=True)[m1]
        assert np.all(ol[m1, m2] == r.optical_lengths(m1, m2))
        ln = r.ray_lengths()
        want = ln.copy()
        want[:, 0], want[:, 1], want[:, 2], want[:, 3] = 1, 1.2, 2.8, 4
        assert np.allclose(ln - want, 0, atol=1e-9, rtol=0)
        assert np.all(ln[m1, m2] == r.ray_lengths(m1, m2))


def test_long_stack_with_discrete_spectrum_falls_back_to_formula_kernels():
    """A monochromatic source selects the kernel that keeps per-line tables of every step in LDS; with hundreds of
    surfaces those tables exceed what a workgroup gets, and the formula kernel traces the scene instead of a launch
    error (340 surfaces: 3 rows x 8 lines x 8 B per step = 65 KB)."""
    n = ot.RefractionIndex("Constant", n=1.5)
    with ot.global_options.no_warnings():
        RT = ot.Raytracer(outline=[-3, 3, -3, 3, -2, 175], seed=3)
        RT.add(ot.RaySource(ot.CircularSurface(r=0.5), pos=[0, 0, -1], spectrum=ot.LightSpectrum("Monochromatic", wl=550.)))
        for k in range(170):  # weak, alternating lenses: the bundle stays near the axis
            R = 400.0 if k % 2 == 0 else -400.0
            RT.add(ot.Lens(ot.SphericalSurface(r=2, R=R), ot.SphericalSurface(r=2, R=-R), n=n, pos=[0, 0, 1.0 * k], d=0.4))
        RT.trace(3000)
        assert not RT.geometry_error and RT.rays.Nt == 342
        w = RT.rays.w_list
        assert np.all(w[:, -1] == 0) and np.all(w[:, -2] > 0), "everything arrives at the end aperture"
        # Fresnel losses of 340 near-normal glass / air passages
        t = float(w[:, -2].astype(np.float64).sum() / w[:, 0].astype(np.float64).sum())
        assert abs(t - 0.96 ** 340) < 0.1 * 0.96 ** 340
        assert RT._msgs.shape == (5, 342) and RT._msgs[:, :-1].sum() == 0


def test_beam_along_a_face_normal_keeps_its_power():
    """A collimated beam whose direction equals the normal of a tilted face bitwise: the plane of incidence is
    undefined (n x s = 0).  The reference gets the normal-incidence transmission 4 n1 n2 / (n1 + n2)^2 out of its
    rounding-noise basis; a NaN weight here would silently absorb the rays."""
    nrm = np.array([0.06, -0.03, 1.0])
    nrm /= np.linalg.norm(nrm)
    n = ot.RefractionIndex("Constant", n=1.6)
    with ot.global_options.no_warnings():
        RT = ot.Raytracer(outline=[-6, 6, -6, 6, -5, 30], seed=9)
        RT.add(ot.RaySource(ot.CircularSurface(r=0.4), pos=[0, 0, -2], s=list(nrm), polarization="Uniform"))
        front = ot.TiltedSurface(r=3, normal=list(nrm))
        RT.add(ot.Lens(front, ot.CircularSurface(r=3), n=n, pos=[0, 0, 5], d=1.5))
        RT.trace(20000)
        assert not RT.geometry_error
        w, pol = RT.rays.w_list.astype(np.float64), RT.rays.pol_list.astype(np.float64)
        assert np.all(np.isfinite(w)) and np.all(np.isfinite(pol[:, :3]))
        assert np.all(w[:, 1] > 0), "no ray is lost at the tilted face"
        T = 4 * 1.6 / (1 + 1.6) ** 2
        np.testing.assert_allclose(w[:, 1] / w[:, 0], T, rtol=2e-6)
        # direction unchanged up to rounding, polarisation still transverse and of unit length
        np.testing.assert_allclose(np.linalg.norm(pol[:, 1], axis=1), 1.0, atol=1e-6)


def test_padded_plane_stride_equals_packed_storage(monkeypatch):
    """From RayStorage.PAD_FROM rays on the planes of the device buffers are a multiple of 128 elements apart (an odd ray
    count would put every plane off the 128-byte lines).  Everything a user can read is what the packed layout gives:
    host lists, counters, detector images and spectra, source images, ray selections, and the same for rays handed in
    (with HURB normals laid out with the stride)."""
    import optrace_amd.ray_storage as rs_mod
    N = (1 << 20) + 37

    def run(pad: bool):
        monkeypatch.setattr(rs_mod.RayStorage, "PAD_FROM", (1 << 20) if pad else (1 << 62))
        with ot.global_options.no_warnings():
            RT = scenes.double_gauss(ot, seed=17)
            RT.trace(N)
            assert (RT.rays._Np > N) == pad and RT.rays._Np % 128 == (0 if pad else N % 128)
            r = RT.rays
            out = dict(p=r.p_list.copy(), w=r.w_list.copy(), n=r.n_list.copy(), pol=r.pol_list.copy(), wl=r.wl_list.copy(),
                       s=r.s0_list.copy(), msgs=RT._msgs.copy())
            assert r.p_list.shape == (N, 17, 3) and r.p_list.flags.f_contiguous and r.s0_list.shape == (N, 3)
            assert r.pol_list.flags.f_contiguous and r.w_list.shape == (N, 17) and r.wl_list.shape == (N,)
            out["img_auto"] = RT.detector_image()._data.copy()
            out["img_user"] = RT.detector_image(extent=[-30., 30., -40., 10.])._data.copy()
            out["img_src"] = RT.detector_image(source_index=3)._data.copy()
            sp = RT.detector_spectrum()
            out["spec"] = np.array(sp._vals)
            out["src_img"] = RT.source_image(2)._data.copy()
            ch = np.zeros(N, dtype=bool)
            ch[[0, 5, N // 2, N - 1]] = True
            sel = r.rays_by_mask(ch)
            out["sel"] = [np.array(a) for a in sel if a is not None]
            out["sel2"] = [np.array(a) for a in r.rays_by_mask(ch, np.array([0, 3, 16, 8])) if a is not None]
            out["sec"] = [np.array(a) for a in r.source_sections(4)]
            # the same rays handed in again (ot_trace walks the whole stride: the padding must be dead)
            init = (out["p"][:, 0], out["s"] * 0 + _dirs(out["p"]), out["pol"][:, 0], out["w"][:, 0], out["wl"])
            RT.trace(N, _initial_rays=init, _N_list=r.N_list)
            out["msgs_inj"] = RT._msgs.copy()
            out["w_inj"] = RT.rays.w_list.copy()
            out["p_inj"] = RT.rays.p_list.copy()
            # HURB with injected normals
            RH = scenes.hurb_slit_lens(ot, seed=5)
            RH.trace(N)
            hn = np.random.default_rng(3).normal(size=(2, N))
            init = (RH.rays.p_list[:, 0].copy(), _dirs(RH.rays.p_list), RH.rays.pol_list[:, 0].copy(),
                    RH.rays.w_list[:, 0].copy(), RH.rays.wl_list.copy())
            RH.trace(N, _initial_rays=init, _hurb_normals=hn, _N_list=RH.rays.N_list)
            out["hurb_p"], out["hurb_w"], out["hurb_msgs"] = RH.rays.p_list.copy(), RH.rays.w_list.copy(), RH._msgs.copy()
            # chunked rendering with odd chunk sizes (two chunks of 1 100 003 and 1 899 998 rays), fused and hit-list paths
            RI = scenes.double_gauss(ot, seed=23)
            RI.ITER_RAYS_STEP = 1_100_003
            imgs = RI.iterative_render(3_000_001, pos=[[0, 0, 160.], [0, 0, 170.]], extent=[[-40., 40., -40., 40.], None])
            out["img_iter0"], out["img_iter1"] = imgs[0]._data.copy(), imgs[1]._data.copy()
        return out

    def _dirs(p):
        d = p[:, 1] - p[:, 0]
        return d / np.linalg.norm(d, axis=1)[:, None]

    a, b = run(True), run(False)
    for key in a:
        if isinstance(a[key], list):
            assert len(a[key]) == len(b[key])
            for x, y in zip(a[key], b[key]):
                assert np.array_equal(x, y, equal_nan=True), key
        elif key.startswith(("img_", "src_img", "spec")):  # sums of atomic adds: equal up to their order
            assert a[key].shape == b[key].shape and np.array_equal(a[key] != 0, b[key] != 0), key
            np.testing.assert_allclose(a[key], b[key], rtol=1e-11, atol=0, err_msg=key)
        else:
            assert np.array_equal(a[key], b[key], equal_nan=True), key
    assert a["msgs"].sum() > 0 and a["img_auto"][..., 3].sum() > 0 and a["hurb_msgs"].sum() > 0


def test_moving_a_detector_keeps_the_trace_shortcut_unless_it_shares_its_surface():
    """`iterative_render` moves its detector from position to position (raytracer.py:1244).  That changes nothing a trace
    depends on, so the unchanged-scene shortcut of `trace` survives it (one integer comparison instead of a snapshot per
    chunk) -- unless the detector's surface object also belongs to a tracing element: then the move is a change of the
    scene and the next trace sees it."""
    from optrace_amd import base as _base
    with ot.global_options.no_warnings():
        RT = scenes.c1_single_lens(ot, seed=3)
        RT.trace(50_000)
        assert RT._scene_unchanged()
        e0 = _base.mutation_epoch()
        a = RT.iterative_render(120_000, pos=[[0, 0, 18.], [0, 0, 22.]], extent=[[-2, 2, -2, 2]] * 2)
        assert _base.mutation_epoch() == e0 and RT._scene_unchanged(), "detector moves must not look like scene changes"
        assert a[0].power() > 0 and not np.array_equal(a[0]._data, a[1]._data)
        # a real change is still seen
        RT.lenses[0].move_to([0, 0, 1.0])
        assert not RT._scene_unchanged()
        RT.trace(50_000)
        # a detector that shares its surface with an aperture: moving it moves the aperture
        RT2 = ot.Raytracer(outline=[-5, 5, -5, 5, -5, 40], seed=1)
        RT2.add(ot.RaySource(ot.CircularSurface(r=1), divergence="None", s=[0, 0, 1], pos=[0, 0, 0]))
        RT2.add(ot.Aperture(ot.RingSurface(r=3, ri=0.5), pos=[0, 0, 10]))
        RT2.add(ot.Detector(ot.RectangularSurface(dim=[4, 4]), pos=[0, 0, 30]))
        RT2.trace(20_000)
        z_before = float(RT2.rays.p_list[:, 1, 2].max())
        # (elements copy the surfaces they are given, so this takes force: the detector now IS the aperture's surface object)
        RT2.detectors[0].__dict__["front"] = RT2.apertures[0].front
        assert RT2.detectors[0].surface is RT2.apertures[0].front
        RT2._detector_requests([dict(detector_index=0, pos=[0, 0, 14.0])])
        assert not RT2._scene_unchanged()
        RT2.trace(20_000)
        assert abs(float(RT2.rays.p_list[:, 1, 2].max()) - 14.0) < 1e-9 and abs(z_before - 10.0) < 1e-9
