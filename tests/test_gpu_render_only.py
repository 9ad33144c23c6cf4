"""Render-only chunks of `iterative_render` (`Raytracer.trace(_tail=...)`, `ot_generate_and_trace_tail`, TailStorage) against the
stored path.  The reference keeps the rays of its last chunk only (raytracer.py:1235-1267); every chunk before it exists to
be binned, and a detector behind the last surface reads of a ray its last section alone (raytracer.py:929-985).  A
render-only trace stores no section and writes that last section for the living rays into a compact two-section storage.

Checked here: the records are the stored path's, bit for bit (same seed: same generation, same arithmetic -- the stored
path is what the reference fixtures pin, tests/test_gpu_parity.py, test_gpu_parity_paths.py); the images of an iterative
render are the stored path's (same pixels lit, sums to 1e-11: the hits arrive in another order); counters are equal; the
last chunk's rays stay in the tracer; sources and detector positions the form does not serve go through the ray storage."""
import numpy as np
import pytest

import optrace_amd as ot
from optrace_amd.ray_storage import TailStorage
import scenes
from test_gpu_fused_detector import same_image

pytestmark = pytest.mark.gpu


class settings:
    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: getattr(ot.Raytracer, k) for k in self.kw}
        for k, v in self.kw.items():
            setattr(ot.Raytracer, k, v)

    def __exit__(self, *a):
        for k, v in self.old.items():
            setattr(ot.Raytracer, k, v)


def hurb_scene(ot_, **kw):
    return scenes.hurb_slit_lens(ot_, **kw)


def relay_with_stop(ot_, **kw):
    """Two lenses around a ring aperture that takes a third of the rays, Lambertian disc source, D65: absorbed rays at
    every surface, a continuous spectrum, polarisation."""
    RT = ot_.Raytracer(outline=[-10, 10, -10, 10, -5, 80], **kw)
    RT.add(ot_.RaySource(ot_.CircularSurface(r=1.5), divergence="Lambertian", div_angle=9, pos=[0, 0, 0],
                         spectrum=ot_.presets.light_spectrum.d65))
    RT.add(ot_.Lens(ot_.SphericalSurface(r=5, R=30), ot_.ConicSurface(r=5, R=-30, k=-1.5), de=0.2, pos=[0, 0, 20],
                    n=ot_.RefractionIndex("Abbe", n=1.6, V=40)))
    RT.add(ot_.Aperture(ot_.RingSurface(r=5, ri=2.4), pos=[0, 0, 30]))
    RT.add(ot_.Lens(ot_.SphericalSurface(r=5, R=25), ot_.SphericalSurface(r=5, R=-40), de=0.2, pos=[0, 0, 40],
                    n=ot_.RefractionIndex("Constant", n=1.5)))
    RT.add(ot_.Detector(ot_.RectangularSurface(dim=[12, 12]), pos=[0, 0, 70]))
    return RT


SCENES = {
    "image_no_pol": lambda **kw: scenes.c4_image_render(ot, **kw),
    "double_gauss_lines": lambda **kw: scenes.double_gauss(ot, **kw),
    "hurb_slit_lens": lambda **kw: hurb_scene(ot, **kw),
    "relay_with_stop": lambda **kw: relay_with_stop(ot, **kw),
    "c1_single_lens": lambda **kw: scenes.c1_single_lens(ot, **kw),
    # the numeric hit levels: aspheres (+ filters, a slit, with and without HURB: feature levels 2 and 3), spline surfaces
    "asphere": lambda **kw: scenes.asphere_scene(ot, **kw),
    "asphere_hurb": lambda **kw: scenes.asphere_scene(ot, use_hurb=True, **kw),
    "freeform": lambda **kw: scenes.freeform_scene(ot, **kw),
}


@pytest.mark.parametrize("name", list(SCENES))
@pytest.mark.parametrize("N", [1, 63, 65, 70_001, 1_300_000])
def test_tail_records_are_the_stored_sections_bit_for_bit(name, N):
    """Same seed through `trace(N)` and `trace(N, _tail=...)`: the set of (p[nt-2], p[nt-1], w[nt-2], wl) over the rays alive
    in the last section is identical; every slot in use beyond them carries weight 0; counters equal."""
    with ot.global_options.no_warnings():
        check_tail_records(SCENES[name](seed=17), N)


@pytest.mark.parametrize("hurb", [False, True])
@pytest.mark.parametrize("scene_seed", range(3000, 3016))
def test_tail_records_of_random_systems(scene_seed, hurb):
    """Random sequential systems (tests/scenes.py::random_scene: lenses with flat / spherical / conic / aspheric / tilted
    surfaces, ring and slit apertures, filters, ideal lenses, with and without polarisation; HURB on: the diffracting form
    of every aperture) -- whatever kernel variant the scene compiler picks, its render-only form writes the same records."""
    with ot.global_options.no_warnings():
        RT = scenes.random_scene(ot, scene_seed, seed=scene_seed, use_hurb=hurb)
        check_tail_records(RT, 50_003)


def check_tail_records(RT, N):
    if True:
        RT.trace(N)
        if RT.geometry_error:
            pytest.skip("random geometry collides")
        msgs = RT._msgs.copy()
        r = RT.rays
        nt = r.Nt
        w = r.w_list[:, nt - 2]
        sel = w > 0
        ref = np.concatenate([r.p_list[sel, nt - 2], r.p_list[sel, nt - 1], w[sel, None].astype(np.float64),
                              r.wl_list[sel, None].astype(np.float64)], axis=1)
        tail = TailStorage()
        RT.trace(N, _tail=tail)
        assert np.array_equal(RT._msgs, msgs)
        assert RT.rays.N == N and RT.rays.Nt == nt, "a render-only trace leaves the stored rays alone"
    assert tail.traced == N and tail.alive == int(sel.sum())
    assert tail.N % 65536 == 0 and tail.alive <= tail.N <= tail._cap
    cap, n = tail._cap, tail.N
    p = tail._dev["p"].view(3, 2, cap)[:, :, :n].cpu().numpy()      # [component, section, slot]
    tw = tail._dev["w"].view(2, cap)[:, :n].cpu().numpy()
    twl = tail._dev["wl"][:n].cpu().numpy()
    assert not tw[1].any(), "section 1: every ray ends absorbed"
    live = tw[0] > 0
    assert int(live.sum()) == tail.alive
    assert np.isfinite(p).all()
    got = np.concatenate([p[:, 0, live].T, p[:, 1, live].T, tw[0, live, None].astype(np.float64),
                          twl[live, None].astype(np.float64)], axis=1)
    order = lambda a: a[np.lexsort(a.T[::-1])]
    assert np.array_equal(order(got), order(ref)), "records differ from the stored sections"


def tail_records(tail):
    """(n, 8) array of the living records of a tail storage: p[nt-2], p[nt-1], w, wl (f64), any order."""
    cap, n = tail._cap, tail.N
    p = tail._dev["p"].view(3, 2, cap)[:, :, :n].cpu().numpy()
    tw = tail._dev["w"].view(2, cap)[:, :n].cpu().numpy()
    twl = tail._dev["wl"][:n].cpu().numpy()
    assert not tw[1].any() and np.isfinite(p).all()
    live = tw[0] > 0
    assert int(live.sum()) == tail.alive and tail.N % 65536 == 0
    return np.concatenate([p[:, 0, live].T, p[:, 1, live].T, tw[0, live, None].astype(np.float64),
                           twl[live, None].astype(np.float64)], axis=1)


@pytest.mark.parametrize("name", ["image_no_pol", "relay_with_stop", "asphere_hurb"])
@pytest.mark.parametrize("n_tail,n_stored", [(70_001, 5_003), (65_536 * 3, 64), (1, 1_000_000), (300_000, 1)])
def test_living_rays_of_a_stored_chunk_join_the_tail(name, n_tail, n_stored):
    """`TailStorage.append_living` (`ot_tail_append`): afterwards the tail holds its own records and, for every ray of the
    stored chunk that is alive in its last section, that section with the weight times the scale (f64 product, one rounding)."""
    order = lambda a: a[np.lexsort(a.T[::-1])]
    with ot.global_options.no_warnings():
        RT = SCENES[name](seed=23)
        alone = TailStorage()
        RT.trace(n_tail, _chunk=0, _tail=alone)
        own = tail_records(alone)
        RT.trace(n_stored, _chunk=1)
        r = RT.rays
        nt = r.Nt
        w = r.w_list[:, nt - 2]
        sel = w > 0
        scale = n_stored / n_tail
        scaled = (w[sel].astype(np.float64) * scale).astype(np.float32)
        joined = np.concatenate([r.p_list[sel, nt - 2], r.p_list[sel, nt - 1], scaled[:, None].astype(np.float64),
                                 r.wl_list[sel, None].astype(np.float64)], axis=1)
        keep = scaled > 0  # (a weight that underflows in the scaling is a ray without weight: a slot like any empty one)
        tail = TailStorage()
        RT.trace(n_tail, _chunk=0, _tail=tail, _tail_room=n_stored + 64)
        assert np.array_equal(order(tail_records(tail)), order(own)), "room for more rays changes nothing"
        tail.append_living(RT.rays, scale)
        assert RT.rays.N == n_stored
    assert tail.traced == n_tail + n_stored and tail.alive == len(own) + int(keep.sum())
    assert np.array_equal(order(tail_records(tail)), order(np.concatenate([own, joined[keep]])))


@pytest.mark.parametrize("merge", [False, True])
@pytest.mark.parametrize("name", list(SCENES))
def test_iterative_render_render_only_equals_stored_path(name, merge):
    """Three chunks, three detector positions, user extents and automatic ones: render-only chunks against the same chunks
    through the ray storage (same seeds).  merge: the stored last chunk is traced before the render-only chunk in front of it
    and binned with it (`TailStorage.append_living`: its rays' weights are rescaled to that chunk's rays in f64 and rounded to
    f32 once -- 6e-8 of a third of the rays); without, every chunk has its own pass and the sums agree to 1e-11."""
    with ot.global_options.no_warnings():
        n = 400_000
        out = {}
        for mode in (True, False):
            RT = SCENES[name](seed=5)
            di = int(np.argmax([d.pos[2] for d in RT.detectors]))  # the detector behind the last surface
            det = RT.detectors[di]
            z0 = float(det.pos[2])
            z_last = max(s.z_max for s in RT.tracing_surfaces)
            pos = [[0, 0, z] for z in (z0, 0.5 * (z0 + z_last) + 0.1, z0 - 0.3)]
            half = 0.5 * max(det.surface.dim) if hasattr(det.surface, "dim") else 4.
            exts = [[-half, half, -half, half], [-0.6 * half, 0.7 * half, -half, 0.2 * half], None]
            traced = []
            orig = RT.trace

            def spy(N, **kw):
                traced.append((N, kw.get("_tail") is not None))
                return orig(N, **kw)

            RT.trace = spy
            if merge and not mode:
                # (the merged form takes automatic extents from its stored LAST chunk, which it traces first; every other form
                # from the first chunk: the stored form gets the merged form's extents to bin into)
                exts = [e if e is not None else list(out[True][0][k]._extent0) for k, e in enumerate(exts)]
            with settings(ITER_RAYS_STEP=n, ITER_RENDER_ONLY=mode, ITER_EXTENT_RAYS=1 << 60, ITER_MERGE_LAST=merge):
                imgs = RT.iterative_render(3 * n + 77, detector_index=di, pos=pos, extent=exts)
            del RT.trace
            if mode and merge:
                assert traced == [(n + 77, False), (n, True), (n, True)], "the last chunk goes through the storage, first"
            else:
                assert traced == [(n, mode), (n, mode), (n + 77, False)], "the last chunk always goes through the storage"
            assert RT.rays.N == n + 77, "... and its rays stay in the tracer"
            out[mode] = (imgs, RT._msgs.copy())
    (a, ma), (b, mb) = out[True], out[False]
    assert np.array_equal(ma, mb)
    for x, y in zip(a, b):
        same_image(x, y, tol=1e-7 if merge else 1e-11)
        assert abs(x.power() - y.power()) <= (3e-8 if merge else 1e-12) * y.power()


def test_chunk_plan_and_speed_path_of_a_long_render():
    """Without ITER_RAYS_STEP: render-only chunks as large as the tail storage allows, then one stored chunk of
    ITER_LAST_RAYS; the result has the power of the plain chunked render to the ray statistics."""
    with ot.global_options.no_warnings():
        RT = scenes.c4_image_render(ot, seed=3)
        traced = []
        orig = RT.trace

        def spy(N, **kw):
            traced.append((N, kw.get("_tail") is not None))
            return orig(N, **kw)

        RT.trace = spy
        N = 3_000_000
        with settings(ITER_LAST_RAYS=1 << 19, ITER_STORAGE_BYTES=60 * (1 << 20)):
            imgs = RT.iterative_render(N, pos=scenes.C4_POSITIONS[:2], extent=[[-8., 8., -8., 8.]] * 2)
        del RT.trace
        last = 1 << 19
        # (the stored last chunk is traced first and binned with the final render-only chunk, ITER_MERGE_LAST)
        assert traced[0] == (last, False) and all(t for _, t in traced[1:])
        assert sum(n for n, _ in traced) == N and all(n <= 1 << 20 for n, t in traced if t)
        assert len(traced) == 4
        with settings(ITER_RENDER_ONLY=False):
            ref = RT.iterative_render(N, pos=scenes.C4_POSITIONS[:2], extent=[[-8., 8., -8., 8.]] * 2)
    for x, y in zip(imgs, ref):
        assert abs(x.power() - y.power()) < 3e-3 * y.power()  # (other rays: statistics)
        assert x._data.shape == y._data.shape


def test_automatic_extents_of_a_render_only_first_chunk():
    """The extents of the first chunk (raytracer.py:1262) come from a sample of the living rays with the stride the generated
    rays would have had; given back as user extents they reproduce the images."""
    with ot.global_options.no_warnings():
        RT = scenes.c4_image_render(ot, seed=9)
        with settings(ITER_RAYS_STEP=600_000, ITER_EXTENT_RAYS=50_000):
            auto = RT.iterative_render(1_800_000, pos=scenes.C4_POSITIONS[:3])
            given = RT.iterative_render(1_800_000, pos=scenes.C4_POSITIONS[:3],
                                        extent=[[float(v) for v in im._extent0] for im in auto])
        with settings(ITER_RAYS_STEP=600_000, ITER_EXTENT_RAYS=1 << 60, ITER_RENDER_ONLY=False):
            full = RT.iterative_render(1_800_000, pos=scenes.C4_POSITIONS[:3])
    for a, g, f in zip(auto, given, full):
        same_image(a, g)
        ea, ef = a._extent0, f._extent0
        assert ef[0] <= ea[0] and ea[1] <= ef[1] and ef[2] <= ea[2] and ea[3] <= ef[3], "a sample's extent lies inside"
        assert 0.995 * f.power() < a.power() <= f.power() * (1 + 1e-12)


def test_scenes_and_positions_the_form_does_not_serve_take_the_storage():
    with ot.global_options.no_warnings():
        # a detector position inside the stack
        RT = scenes.double_gauss(ot, seed=2)
        k = [0]
        z_mid = 0.5 * (RT.lenses[2].back.pos[2] + RT.apertures[0].pos[2])
        assert not RT._render_only_applies([0, 0], [RT.detectors[0].pos, [0, 0, z_mid]])
        assert RT._render_only_applies([0], [RT.detectors[0].pos])
        # a source whose orientation is a Python callable: its generation needs a position pre-pass on the host
        RA = ot.Raytracer(outline=[-10, 10, -10, 10, -25, 60], seed=2)
        RA.add(ot.RaySource(ot.RectangularSurface(dim=[2, 3]), divergence="None", orientation="Function",
                            or_func=lambda x, y: np.tile([0., 0., 1.], (x.shape[0], 1)), pos=[0, 0, -20],
                            spectrum=ot.LightSpectrum("Monochromatic", wl=550.)))
        RA.add(ot.Lens(ot.SphericalSurface(r=3, R=8), ot.SphericalSurface(r=3, R=-8), de=0.1,
                       n=ot.RefractionIndex("Constant", n=1.5), pos=[0, 0, 0]))
        RA.add(ot.Detector(ot.RectangularSurface(dim=[4, 4]), pos=[0, 0, 20]))
        assert not RA._render_only_applies([0], [RA.detectors[0].pos])
        tails = []
        orig = RA.trace
        RA.trace = lambda N, **kw: (tails.append(kw.get("_tail") is not None), orig(N, **kw))[1]
        with settings(ITER_RAYS_STEP=50_000):
            imgs = RA.iterative_render(150_000)
        del RA.trace
        assert tails == [False] * 3 and imgs[0].power() > 0
        with pytest.raises(ValueError):
            RT.trace(1000, _tail=TailStorage(), _initial_rays=(None,) * 5)


def test_a_chunk_without_survivors():
    """Every ray absorbed before the last section (an aperture that blocks all): the render-only chunk contributes nothing
    and nothing breaks."""
    with ot.global_options.no_warnings():
        RT = ot.Raytracer(outline=[-5, 5, -5, 5, -5, 40], seed=4)
        RT.add(ot.RaySource(ot.CircularSurface(r=1), divergence="None", s=[0, 0, 1], pos=[0, 0, 0]))
        RT.add(ot.Aperture(ot.CircularSurface(r=3), pos=[0, 0, 10]))
        RT.add(ot.Detector(ot.RectangularSurface(dim=[4, 4]), pos=[0, 0, 30]))
        with settings(ITER_RAYS_STEP=100_000):
            imgs = RT.iterative_render(300_000, extent=[-2, 2, -2, 2])
        assert imgs[0].power() == 0.0


@pytest.mark.parametrize("render_only", [True, False])
def test_first_chunk_is_cropped_to_the_sampled_extent_like_the_later_ones(render_only):
    """A line-like image: `RenderImage._fix_extent` widens its thin side to a band (render_image.py:240-262).  The extents of an
    iterative render come from a sample of the first chunk (ITER_EXTENT_RAYS); hits of that chunk outside the sample's
    extent but inside the band must be dropped as they are for every later chunk (raytracer.py:1262: cropped to
    `_extent0`) -- the image equals the one a caller gets who passes that extent himself, chunk by chunk."""
    with ot.global_options.no_warnings():
        RT = ot.Raytracer(outline=[-8, 8, -8, 8, 0, 40], seed=21)
        RT.add(ot.RaySource(ot.Line(r=2.0), divergence="Isotropic", div_angle=3, s=[0, 0, 1], pos=[0, 0, 0],
                            spectrum=ot.LightSpectrum("Monochromatic", wl=550.)))
        RT.add(ot.Lens(ot.SphericalSurface(r=3, R=8), ot.SphericalSurface(r=3, R=-8), de=0.1, pos=[0, 0, 12],
                       n=ot.RefractionIndex("Constant", n=1.5)))
        RT.add(ot.Detector(ot.RectangularSurface(dim=[16, 16]), pos=[0, 0, 36]))
        with settings(ITER_RAYS_STEP=200_000, ITER_EXTENT_RAYS=5_000, ITER_RENDER_ONLY=render_only):
            auto = RT.iterative_render(400_000)[0]
            e0 = [float(v) for v in auto._extent0]
            given = RT.iterative_render(400_000, extent=e0)[0]
        with settings(ITER_RAYS_STEP=200_000, ITER_EXTENT_RAYS=1 << 60, ITER_RENDER_ONLY=render_only):
            full = RT.iterative_render(400_000)[0]
    assert (e0[1] - e0[0]) > 20 * (e0[3] - e0[2]) or (e0[3] - e0[2]) > 20 * (e0[1] - e0[0]), "a line-like image"
    same_image(auto, given)
    assert auto.power() < full.power(), "the sample's extent is smaller: some hits of every chunk fall outside"


@pytest.mark.parametrize("proj", ["Orthographic", "Equidistant"])
def test_render_only_with_a_spherical_detector(proj):
    """Curved detectors behind the last surface: without a sphere projection (Orthographic) the two-section storage goes
    through the pair-only tile kernel (`detector_hit_pair`), with one (Equidistant) through the hit-list chain -- both must
    give the stored path's images, also where the ray ends inside the detector's z range."""
    with ot.global_options.no_warnings():
        out = {}
        for mode in (True, False):
            RT = scenes.c4_image_render(ot, seed=8)
            RT.add(ot.Detector(ot.SphericalSurface(r=7.5, R=-30), pos=[0, 0, 34]))
            pos = [[0, 0, 33.], [0, 0, 35.5], [0, 0, 39.2]]  # the last one: the outline (z = 40) cuts through the sphere's sag
            exts = [[-6., 6., -6., 6.], None, None] if proj == "Orthographic" else [None] * 3
            if not mode:  # (automatic extents: those the render-only form took from its stored last chunk)
                exts = [e if e is not None else list(out[True][k]._extent0) for k, e in enumerate(exts)]
            with settings(ITER_RAYS_STEP=500_000, ITER_RENDER_ONLY=mode, ITER_EXTENT_RAYS=1 << 60):
                out[mode] = RT.iterative_render(1_500_000, detector_index=1, pos=pos, projection_method=proj, extent=exts)
    for x, y in zip(out[True], out[False]):
        same_image(x, y, tol=1e-7)
