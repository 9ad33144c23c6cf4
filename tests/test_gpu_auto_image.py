"""Detector images with an AUTOMATIC extent in one pass over the ray sections (`Raytracer._auto_image_one_pass`,
`ot_detector_image_auto_*`, csrc/ot_detector_fused.hpp last section) against the chain hit list -> binning
(`ot_detector_hits_multi` + `ot_render_accumulate`), which the reference fixtures pin (tests/test_gpu_parity.py):
same extent (bit for bit), same pixels lit, f64 sums in another order (raytracer.py:1042-1049, 1053-1098)."""
import os

import numpy as np
import pytest
import torch

import optrace_amd as ot
from optrace_amd import _capi, detector as _detector
import scenes
from test_gpu_fused_detector import same_image, image_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["line buffers", "plain tile kernel"], autouse=True)
def tile_kernel(request, monkeypatch):
    """Every case with both forms of the tile kernel's first pass: `fuse_tiles_lb_kernel` (one detector, images of up to
    361 tiles: records leave the CU as whole 384-byte segments) and `fuse_tiles_kernel` (OT_TILE_LINEBUF=0)."""
    if request.param == "plain tile kernel":
        monkeypatch.setenv("OT_TILE_LINEBUF", "0")
    else:
        monkeypatch.delenv("OT_TILE_LINEBUF", raising=False)


class one_pass_from:
    """Raytracer.AUTO_ONE_PASS_FROM (and optionally the sample stride) for the calls inside."""

    def __init__(self, n, stride=None):
        self.n, self.stride = n, stride

    def __enter__(self):
        self.old = (ot.Raytracer.AUTO_ONE_PASS_FROM, ot.Raytracer.AUTO_SAMPLE_STRIDE)
        ot.Raytracer.AUTO_ONE_PASS_FROM = self.n
        if self.stride:
            ot.Raytracer.AUTO_SAMPLE_STRIDE = self.stride

    def __exit__(self, *a):
        ot.Raytracer.AUTO_ONE_PASS_FROM, ot.Raytracer.AUTO_SAMPLE_STRIDE = self.old


def both(RT, **kw):
    """(one pass, chain) for the same request; the one-pass form must have applied."""
    calls = []
    orig = RT._auto_image_one_pass

    def spy(*a, **k):
        img = orig(*a, **k)
        calls.append(img is not None)
        return img

    RT._auto_image_one_pass = spy
    try:
        with one_pass_from(1), ot.global_options.no_warnings():
            one = RT.detector_image(**kw)
    finally:
        del RT._auto_image_one_pass
    assert calls == [True], "the one-pass form did not apply"
    with one_pass_from(1 << 60), ot.global_options.no_warnings():
        chain = RT.detector_image(**kw)
    return one, chain


@pytest.mark.parametrize("N", [3000, 400_001, 2_500_000])
def test_one_pass_equals_chain_extended_image(N):
    RT = image_scene(N=N)
    one, chain = both(RT)
    same_image(one, chain)
    assert one.long_desc == chain.long_desc and one.projection == chain.projection
    assert abs(one.power() - chain.power()) <= 1e-12 * chain.power()
    same_image(*both(RT, source_index=0))


def test_one_pass_with_resolution_limit_and_sparse_sample():
    """`limit` widens the extent (render_image.py:252-255) and filters; a sample of a few hundred rays (stride 4096) gives
    a much smaller box than the hits' extent: the margin tiles and the escape list take the rest."""
    RT = image_scene(N=1_200_000)
    same_image(*both(RT, limit=5), tol=1e-9)
    with one_pass_from(1, stride=4096):
        same_image(*both(RT))


@pytest.mark.parametrize("sides", [[4.0, 1.9], [1.0, 4.4], [4.2, 2.0], [3.0, 3.0]])
def test_one_pass_image_ratios(sides):
    """Side ratios around the snaps of RenderImage._pixel_counts (945 x 945 / 2835 / 4725 pixels)."""
    RT = ot.Raytracer(outline=[-8, 8, -8, 8, 0, 40], no_pol=True, seed=11)
    RT.add(ot.RaySource(ot.RGBImage(scenes.synthetic_rgb_image(), sides), divergence="Isotropic",
                        div_angle=np.rad2deg(np.arctan(3 / 12) * 1.2), s=[0, 0, 1], pos=[0, 0, 0],
                        orientation="Converging", conv_pos=[0, 0, 12]))
    RT.add(ot.Lens(ot.SphericalSurface(r=3, R=8), ot.SphericalSurface(r=3, R=-8), de=0.1, pos=[0, 0, 12],
                   n=ot.RefractionIndex("Abbe", n=1.5, V=40)))
    RT.add(ot.Detector(ot.RectangularSurface(dim=[16, 16]), pos=[0, 0, 36]))
    with ot.global_options.no_warnings():
        RT.trace(700_000)
    one, chain = both(RT)
    same_image(one, chain)


def test_one_pass_on_a_spherical_detector_without_projection_and_a_second_detector():
    with ot.global_options.no_warnings():
        RT = image_scene(N=600_000)
        RT.add(ot.Detector(ot.SphericalSurface(r=7.5, R=-30), pos=[0, 0, 34]))
    same_image(*both(RT, detector_index=1, projection_method="Orthographic"))
    same_image(*both(RT, detector_index=1, projection_method=None))


def test_not_applicable_cases_take_the_chain():
    """Point-like image, a detector no ray reaches, a sphere projection: `_auto_image_one_pass` declines (None) and the
    image is the chain's."""
    with ot.global_options.no_warnings():
        RT = scenes.double_gauss(ot, seed=5)
        RT.add(ot.Detector(ot.RectangularSurface(dim=[1, 1]), pos=[1500, 0, 150]))
        RT.add(ot.Detector(ot.SphericalSurface(r=7.5, R=-30), pos=[0, 0, 150]))
        RT.trace(300_000)
        with one_pass_from(1):
            assert RT._auto_image_one_pass(dict(detector_index=1, source_index=None, extent=None,
                                                projection_method="Equidistant"), None) is None
            assert RT._auto_image_one_pass(dict(detector_index=2, source_index=None, extent=None,
                                                projection_method="Equidistant"), None) is None
            a, a1 = RT.detector_image(), RT.detector_image(detector_index=1)
        with one_pass_from(1 << 60):
            b, b1 = RT.detector_image(), RT.detector_image(detector_index=1)
    same_image(a, b)
    np.testing.assert_array_equal(a1.extent, b1.extent)
    assert a1.power() == 0.0


def _rq(RT, k=0):
    return RT._detector_requests([dict(detector_index=k, source_index=None, extent=None, projection_method=None)])[0]


@pytest.mark.parametrize("grid_kind", ["coarse tiles", "half covered", "one tile"])
def test_any_grid_gives_the_same_image(grid_kind):
    """The provisional grid decides speed, not the result: tiles wider than a 64-pixel window (records outside their window
    go straight to the image), a grid over part of the hits (the rest escapes), a single tile."""
    RT = image_scene(N=500_000)
    with one_pass_from(1 << 60), ot.global_options.no_warnings():
        chain = RT.detector_image()
    rq = _rq(RT)
    e = chain._extent0 if hasattr(chain, "_extent0") else chain.extent
    sx, sy = e[1] - e[0], e[3] - e[2]
    grid = {"coarse tiles": (e[0] - 0.1 * sx, e[2] - 0.1 * sy, sx / 4, sy / 3, 5, 5),
            "half covered": (e[0] + 0.45 * sx, e[2] - 0.1 * sy, sx / 16, sy / 16, 12, 20),
            "one tile": (e[0] - sx, e[2] - sy, 3 * sx, 3 * sy, 1, 1)}[grid_kind]
    auto = _detector.AutoImage(RT.rays, rq["Ns"], rq["Ne"] - rq["Ns"], rq["surf_desc"], _capi.PROJECTIONS[None], grid)
    np.testing.assert_array_equal(auto.extent, np.asarray(e, dtype=np.float64))
    if grid_kind == "half covered":
        assert 0 < auto.escaped <= auto.escape_capacity
    else:
        assert auto.escaped == 0
    img = ot.RenderImage(extent=auto.extent.copy())
    img._limit = None
    img._fix_extent()
    Nx, Ny = img._pixel_counts()
    hist = torch.zeros(Ny * Nx * 4, dtype=torch.float64, device="cuda")
    auto.finish(img.extent, Nx, Ny, hist)
    img._dev, img._host = hist.view(Ny, Nx, 4), None
    same_image(img, chain)


def test_escape_list_overflow_is_reported_and_cancel_frees():
    RT = image_scene(N=2_000_000)
    rq = _rq(RT)
    grid = (100.0, 100.0, 1.0, 1.0, 2, 2)  # nowhere near the image: every hit escapes
    auto = _detector.AutoImage(RT.rays, rq["Ns"], rq["Ne"] - rq["Ns"], rq["surf_desc"], _capi.PROJECTIONS[None], grid)
    assert auto.escaped > auto.escape_capacity >= 1 << 18
    assert np.all(np.isfinite(auto.extent))
    auto.cancel()
    auto.cancel()  # idempotent
    with one_pass_from(1), ot.global_options.no_warnings():  # and the next image is unaffected
        same_image(RT.detector_image(), RT.detector_image(_unfused=True))


def test_argument_checks():
    RT = image_scene(N=5000)
    rq = _rq(RT)
    n = rq["Ne"] - rq["Ns"]
    for grid in [(0.0, 0.0, 0.0, 1.0, 4, 4), (0.0, 0.0, 1.0, 1.0, 0, 4), (np.nan, 0.0, 1.0, 1.0, 4, 4)]:
        with pytest.raises(_capi.BackendError):
            _detector.AutoImage(RT.rays, rq["Ns"], n, rq["surf_desc"], _capi.PROJECTIONS[None], grid)
    with pytest.raises(_capi.BackendError):  # more tiles than a workgroup keeps counters for
        _detector.AutoImage(RT.rays, rq["Ns"], n, rq["surf_desc"], _capi.PROJECTIONS[None], (0.0, 0.0, 1.0, 1.0, 64, 64))
    with pytest.raises(_capi.BackendError):  # sphere projection
        _detector.detector_extent_sample(RT.rays, rq["Ns"], n, rq["surf_desc"], _capi.PROJECTIONS["Equidistant"], 128)
    with pytest.raises(_capi.BackendError):
        _detector.detector_extent_sample(RT.rays, rq["Ns"], n + 1, rq["surf_desc"], _capi.PROJECTIONS[None], 128)
    e = _detector.detector_extent_sample(RT.rays, rq["Ns"], n, rq["surf_desc"], _capi.PROJECTIONS[None], 1)
    with one_pass_from(1 << 60), ot.global_options.no_warnings():
        np.testing.assert_array_equal(e, RT.detector_image()._extent0)  # stride 1: every ray
