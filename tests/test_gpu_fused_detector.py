"""Detector images with a known extent: hit search and binning in one pass (`ot_detector_images`,
csrc/ot_detector_fused.hpp) against the two-step chain `ot_detector_hits_multi` + `ot_render_accumulate`, which the
reference fixtures pin (tests/test_gpu_parity.py).  Same hits, same pixels; f64 sums in another order."""
import os

import numpy as np
import pytest
import torch

import optrace_amd as ot
import scenes

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["line buffers", "plain tile kernel"], autouse=True)
def tile_kernel(request, monkeypatch):
    """Every case with both forms of the tile kernel's first pass: `fuse_tiles_lb_kernel` (one detector, images of up to
    361 tiles: records leave the CU as whole 384-byte segments) and `fuse_tiles_kernel` (OT_TILE_LINEBUF=0)."""
    if request.param == "plain tile kernel":
        monkeypatch.setenv("OT_TILE_LINEBUF", "0")
    else:
        monkeypatch.delenv("OT_TILE_LINEBUF", raising=False)


class pinned:
    """OT_RENDER_PATH = direct | tiles for the calls inside (None: the probe decides)."""

    def __init__(self, path):
        self.path = path

    def __enter__(self):
        self.old = os.environ.pop("OT_RENDER_PATH", None)
        if self.path:
            os.environ["OT_RENDER_PATH"] = self.path

    def __exit__(self, *a):
        os.environ.pop("OT_RENDER_PATH", None)
        if self.old is not None:
            os.environ["OT_RENDER_PATH"] = self.old


def same_image(a, b, tol=1e-11):
    A, B = a._data, b._data
    assert A.shape == B.shape
    np.testing.assert_allclose(a.extent, b.extent, rtol=0, atol=0)
    assert A[..., 3].sum() > 0
    assert np.array_equal(A[..., 3] != 0, B[..., 3] != 0), "same pixels lit"
    assert np.abs(A - B).max() <= tol * np.abs(A).max()


def image_scene(N=400_000, seed=3, no_pol=True):
    """Extended RGB image source through a lens onto a square detector (geometry of examples/image_render_many_rays.py)."""
    RT = ot.Raytracer(outline=[-8, 8, -8, 8, 0, 40], no_pol=no_pol, seed=seed)
    RT.add(ot.RaySource(ot.RGBImage(scenes.synthetic_rgb_image(), [4, 3]), divergence="Isotropic",
                        div_angle=np.rad2deg(np.arctan(3 / 12) * 1.2), s=[0, 0, 1], pos=[0, 0, 0],
                        orientation="Converging", conv_pos=[0, 0, 12]))
    RT.add(ot.Lens(ot.SphericalSurface(r=3, R=8), ot.SphericalSurface(r=3, R=-8), de=0.1, pos=[0, 0, 12],
                   n=ot.RefractionIndex("Abbe", n=1.5, V=40)))
    RT.add(ot.Detector(ot.RectangularSurface(dim=[16, 16]), pos=[0, 0, 36]))
    with ot.global_options.no_warnings():
        RT.trace(N)
    return RT


@pytest.mark.parametrize("path", ["direct", "tiles", None])
def test_fused_equals_two_step_extended_image(path):
    RT = image_scene()
    for extent in ([-8., 8., -8., 8.], [-2., 1., -1.5, 0.5], [-9., 9., -3., 3.]):  # whole, crop, ratio 3 image
        with pinned(path):
            fused = RT.detector_image(extent=extent)
            two = RT.detector_image(extent=extent, _unfused=True)
        same_image(fused, two)
    # power inside the crop is what the hits inside carry
    full = RT.detector_image(extent=[-8., 8., -8., 8.])
    part = RT.detector_image(extent=[-2., 1., -1.5, 0.5])
    assert 0 < part.power() < full.power()


@pytest.mark.parametrize("path", ["direct", "tiles"])
def test_fused_point_images_and_source_selection(path):
    """Five PSF-like spots (double Gauss): everything lands in a few pixels; one source alone selects its ray range."""
    with ot.global_options.no_warnings():
        RT = scenes.double_gauss(ot, seed=5)
        RT.trace(300_000)
        e4 = [float(v) for v in RT.detector_image(source_index=4)._extent0]  # where the spot of the last source lies
        e0 = [float(v) for v in RT.detector_image(source_index=0)._extent0]
        for kw in (dict(extent=[-45., 45., -45., 45.]), dict(extent=e4, source_index=4),
                   dict(extent=e0, source_index=0, limit=5.)):
            with pinned(path):
                fused = RT.detector_image(**kw)
                two = RT.detector_image(**kw, _unfused=True)
            same_image(fused, two, tol=1e-9 if "limit" in kw else 1e-11)
            assert fused.long_desc == two.long_desc


@pytest.mark.parametrize("projection", ["Equidistant", "Orthographic", "Equal-Area", "Stereographic"])
def test_fused_spherical_detector_projections(projection):
    with ot.global_options.no_warnings():
        RT = ot.Raytracer(outline=[-10, 10, -10, 10, -10, 40], seed=9)
        RT.add(ot.RaySource(ot.CircularSurface(r=1.5), divergence="Lambertian", div_angle=25, pos=[0, 0, 0]))
        RT.add(ot.Detector(ot.SphericalSurface(r=7.9, R=-8), pos=[0, 0, 12]))
        RT.trace(200_000)
        auto = RT.detector_image(projection_method=projection)
        ext = [float(v) for v in auto._extent0]
        for path in ("direct", "tiles"):
            with pinned(path):
                fused = RT.detector_image(extent=ext, projection_method=projection)
                two = RT.detector_image(extent=ext, projection_method=projection, _unfused=True)
            same_image(fused, two)
            assert fused.projection == projection


def test_fused_numeric_detector_and_no_hits():
    """A tilted (numerically intersected) detector goes through the NUMERIC kernels; an extent without hits gives an
    empty image instead of an error."""
    with ot.global_options.no_warnings():
        RT = ot.Raytracer(outline=[-10, 10, -10, 10, -10, 40], seed=11)
        RT.add(ot.RaySource(ot.RectangularSurface(dim=[2, 1]), divergence="Isotropic", div_angle=8, pos=[0, 0, 0]))
        RT.add(ot.Detector(ot.TiltedSurface(r=6, normal=[0.2, 0.1, 1]), pos=[0, 0, 20]))
        RT.trace(150_000)
        for path in ("direct", "tiles"):
            with pinned(path):
                fused = RT.detector_image(extent=[-6., 6., -6., 6.])
                two = RT.detector_image(extent=[-6., 6., -6., 6.], _unfused=True)
            same_image(fused, two)
        empty = RT.detector_image(extent=[5.5, 5.9, 5.5, 5.9])
        assert empty.power() == 0.0 and empty._data.shape == (945, 945, 4)


@pytest.mark.parametrize("path", ["direct", "tiles", None])
def test_iterative_render_positions_equal_single_images(path):
    """Chunks after the first are binned by the fused kernels, up to 8 detector positions per pass: the result equals
    images rendered position by position from the same chunks."""
    N, step = 600_000, 200_000
    pos = [[0, 0, 30.], [0, 0, 33.], [0, 0, 36.], [0.5, 0, 38.]]
    ext = [[-8., 8., -8., 8.]] * 3 + [None]
    with ot.global_options.no_warnings(), pinned(path):
        RT = image_scene(N=1000, seed=21)
        RT.ITER_RAYS_STEP = step
        imgs = RT.iterative_render(N, pos=pos, extent=ext)
        # the same chunks by hand: trace(chunk i), one detector_image per position, scaled sums (raytracer.py:1247-1267)
        # (automatic extents: those of the LAST chunk, the stored one, which the render traces first; its rays are binned with
        # the render-only chunk before it after one more rounding of their weights: 1e-7 instead of 1e-11)
        ref = [None] * len(pos)
        extents = list(ext)
        n_chunks = N // step
        for i in [n_chunks - 1] + list(range(n_chunks - 1)):
            RT.trace(step, _chunk=i)
            for j, p in enumerate(pos):
                RT.detectors[0].move_to(p)
                im = RT.detector_image(extent=extents[j], _unfused=True)
                if ref[j] is None:
                    extents[j] = [float(v) for v in im._extent0]
                    ref[j] = im._data * (step / N)
                else:
                    ref[j] = ref[j] + im._data * (step / N)
    for j in range(len(pos)):
        assert imgs[j]._data.shape == ref[j].shape
        assert np.array_equal(imgs[j]._data[..., 3] != 0, ref[j][..., 3] != 0), "same pixels lit"
        assert np.abs(imgs[j]._data - ref[j]).max() <= 1e-7 * np.abs(ref[j]).max()
        assert abs(imgs[j].power() - ref[j][..., 3].sum()) <= 3e-8 * ref[j][..., 3].sum()


@pytest.mark.parametrize("path", ["direct", "tiles"])
def test_fused_detector_not_behind_the_last_surface(path):
    """The fused kernels settle a flat detector behind the last surface from the last two sections of a ray
    (`detector_hit_last`); everywhere else -- before the lens, inside it, in a lens stack -- a wave falls back to the
    section search.  Both give the two-step chain's image, also when the two cases meet in one wave (rays that die at
    the stop next to rays that reach the detector)."""
    with ot.global_options.no_warnings():
        RT = image_scene(N=300_000, seed=11)
        for z in (6.0, 12.0, 12.9, 20.0):          # before the lens, at its centre plane, behind its back vertex, far behind
            RT.detectors[0].move_to([0, 0, z])
            with pinned(path):
                same_image(RT.detector_image(extent=[-6., 6., -6., 6.]),
                           RT.detector_image(extent=[-6., 6., -6., 6.], _unfused=True))
        RT2 = scenes.double_gauss(ot, seed=4)
        RT2.trace(200_000)
        z_last = max(s.z_max for s in RT2.tracing_surfaces[:-1])
        stop_z = [el for el in RT2.apertures][0].pos[2]
        for z in (stop_z + 2.0, z_last + 1.0, z_last + 60.0):   # inside the stack (behind the stop), just behind, far behind
            RT2.detectors[0].move_to([0, 0, z])
            with pinned(path):
                same_image(RT2.detector_image(extent=[-40., 40., -40., 40.]),
                           RT2.detector_image(extent=[-40., 40., -40., 40.], _unfused=True))


# ---- compact hit lists: automatic extents on long bundles (ot_detector_req.fill, ot_render_accumulate_compact) -------
@pytest.mark.parametrize("path", ["direct", "tiles", None])
@pytest.mark.parametrize("N", [3000, 400_001, 2_500_000])
def test_compact_hit_list_equals_dense_list(path, N):
    """`detector_image(extent=None)` through a hit list that holds the valid hits only, gathered at the front of its 1024
    pieces, against the dense list (one entry per ray, weight 0 for no hit): same extent, same pixels, sums to 1e-11 --
    with both binning paths, ray counts that leave pieces empty, ragged and (2.5 M rays: 3 workgroups per piece) shared."""
    RT = image_scene(N=N)
    old = RT.COMPACT_HITS_FROM
    try:
        with pinned(path), ot.global_options.no_warnings():
            type(RT).COMPACT_HITS_FROM = 1 << 60
            dense = RT.detector_image()
            type(RT).COMPACT_HITS_FROM = 1
            compact = RT.detector_image()
            one_src = RT.detector_image(source_index=0)
    finally:
        type(RT).COMPACT_HITS_FROM = old
    same_image(dense, compact)
    same_image(dense, one_src)
    assert abs(dense.power() - compact.power()) <= 1e-12 * dense.power()


def test_compact_hit_list_on_a_point_image_and_without_hits():
    """PSF-like image (the direct binning path by the probe's verdict) and a detector no ray reaches."""
    with ot.global_options.no_warnings():
        RT = scenes.double_gauss(ot, seed=5)
        RT.add(ot.Detector(ot.RectangularSurface(dim=[1, 1]), pos=[1500, 0, 150]))
        RT.trace(2_300_000)
        old = RT.COMPACT_HITS_FROM
        try:
            type(RT).COMPACT_HITS_FROM = 1 << 60
            dense, dense_empty = RT.detector_image(), RT.detector_image(detector_index=1)
            type(RT).COMPACT_HITS_FROM = 1
            compact, compact_empty = RT.detector_image(), RT.detector_image(detector_index=1)
        finally:
            type(RT).COMPACT_HITS_FROM = old
    same_image(dense, compact)
    assert compact_empty.power() == 0.0 and dense_empty.power() == 0.0
    np.testing.assert_array_equal(compact_empty.extent, dense_empty.extent)


def test_kept_scratch_grows_and_can_be_handed_back():
    """The binning paths keep their scratch per stream between calls (csrc/ot_api.hip::workspace): a larger bundle after a
    smaller one reallocates, `ot_scratch_trim` returns everything to the driver, and the images do not depend on any of it."""
    from optrace_amd import _capi
    lib = _capi.load_library()
    with pinned("tiles"), ot.global_options.no_warnings():
        small = image_scene(N=300_000)
        a1 = small.detector_image(extent=[-8., 8., -8., 8.])
        big = image_scene(N=2_400_000)
        b1, b1a = big.detector_image(extent=[-8., 8., -8., 8.]), big.detector_image()
        _capi.check(lib.ot_scratch_trim())
        a2 = small.detector_image(extent=[-8., 8., -8., 8.])
        b2, b2a = big.detector_image(extent=[-8., 8., -8., 8.]), big.detector_image()
    same_image(a1, a2, tol=1e-12)
    same_image(b1, b2, tol=1e-12)
    same_image(b1a, b2a, tol=1e-12)


def test_iterative_render_extents_come_from_a_sample_like_the_first_iteration():
    """Automatic extents of `iterative_render` are those of the reference's first iteration of 1 M rays, everything else is
    cropped to them (raytracer.py:1212, 1262).  A chunk here is much larger: an evenly spread sample of ITER_EXTENT_RAYS of its
    rays fixes the extents.  The sample's extent lies inside the whole chunk's, nearly all the power stays, and the image is
    the one a caller gets who passes that extent himself."""
    RT = image_scene(N=1000)
    old = ot.Raytracer.ITER_EXTENT_RAYS
    try:
        with ot.global_options.no_warnings():
            ot.Raytracer.ITER_EXTENT_RAYS = 1 << 60  # every ray of the first chunk
            full = RT.iterative_render(600_000)[0]
            ot.Raytracer.ITER_EXTENT_RAYS = 20_000
            samp = RT.iterative_render(600_000)[0]
            given = RT.iterative_render(600_000, extent=[float(v) for v in samp._extent0])[0]
    finally:
        ot.Raytracer.ITER_EXTENT_RAYS = old
    ef, es = full._extent0, samp._extent0
    assert ef[0] <= es[0] and es[1] <= ef[1] and ef[2] <= es[2] and es[3] <= ef[3]
    assert not np.array_equal(ef, es), "30 rays in 600 000 reach further than the 20 000 sampled ones"
    assert 0.99 * full.power() < samp.power() <= full.power()
    same_image(samp, given)
