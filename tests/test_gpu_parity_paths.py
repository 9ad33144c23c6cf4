"""The detector paths production sizes take, each compared DIRECTLY with the reference's fixture images.

`tests/test_gpu_parity.py::test_detector_image_matches_reference` runs the 1.5-2.5 k-ray fixtures through whatever
`detector_image` picks at that size: the dense hit-list chain for automatic extents, `fuse_direct` for user extents.  The
configurations of BASELINE.json (1e7 .. 2e8 rays) and `iterative_render` execute other kernels: compact hit lists, the
one-pass speculative grid (`_auto_image_one_pass`), the tile kernels with and without line buffers, the multi-detector tile
pass `fuse_tiles_kernel<false, 2 / 4 / 8, 1>`.  Here every one of them is forced onto the same fixtures (thresholds set to
1, `OT_RENDER_PATH`, `OT_TILE_LINEBUF`) and held to the same bar against the reference's own images
(`det*/<projection>/img`, `det*/user/img` of tests/golden/trace_*.npz): extent 1e-9, power 1e-6, image norm 1e-4
(BASELINE.json north_star).  Where a path does not apply to a fixture (point-like image, sphere projection) the test
asserts that it declined -- the image is then the chain's, which the other test pins.

Reference: raytracer.py:881-1098 (`_hit_detector`, `detector_image`), :1134-1279 (`iterative_render`),
image/render_image.py:361-421."""
import os

import numpy as np
import pytest

import optrace_amd as ot
from helpers import assert_close, sparse_to_dense, image_rel_l1
from test_gpu_parity import gpu_trace

pytestmark = pytest.mark.gpu

NAMES = ["c1_single_lens", "double_gauss", "mixed_geometry", "arizona_eye", "asphere", "hurb_slit_lens"]


class forced:
    """Thresholds of `Raytracer.detector_image` and environment switches of the library for the calls inside."""

    def __init__(self, one_pass=None, compact=None, render_path=None, linebuf=None):
        self.cls = dict(AUTO_ONE_PASS_FROM=one_pass, COMPACT_HITS_FROM=compact)
        self.env = dict(OT_RENDER_PATH=render_path, OT_TILE_LINEBUF=linebuf)

    def __enter__(self):
        self.old_cls = {k: getattr(ot.Raytracer, k) for k in self.cls}
        self.old_env = {k: os.environ.pop(k, None) for k in self.env}
        for k, v in self.cls.items():
            if v is not None:
                setattr(ot.Raytracer, k, v)
        for k, v in self.env.items():
            if v is not None:
                os.environ[k] = v
        return self

    def __exit__(self, *a):
        for k, v in self.old_cls.items():
            setattr(ot.Raytracer, k, v)
        for k, v in self.old_env.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def projections(det):
    return ["Equidistant", "Orthographic", "Equal-Area", "Stereographic"] if isinstance(det.surface, ot.SphericalSurface) \
        else [None]


def check_against_fixture(img, g, key, what):
    ref = sparse_to_dense(g, f"{key}/img")
    assert img._data.shape == ref.shape, what
    assert_close(img.extent, g[f"{key}/img/extent"], rtol=1e-9, atol=1e-11, what=f"{what}: image extent")
    pw = float(g[f"{key}/img/power"])
    assert abs(img.power() - pw) <= 1e-6 * max(pw, 1e-300), what
    if pw > 0:
        err = image_rel_l1(img._data, ref)
        assert np.all(err < 1e-4), (what, err)


def spy_one_pass(RT):
    """-> list that receives True / False per call of `_auto_image_one_pass` (applied / declined)."""
    calls = []
    orig = RT._auto_image_one_pass

    def spy(*a, **k):
        img = orig(*a, **k)
        calls.append(img is not None)
        return img

    RT._auto_image_one_pass = spy
    return calls


# (scene, detector, projection) where the one-pass form applies to the fixture's bundle; everywhere else it must decline
# (sphere projections with transcendentals, numeric detectors, point- / line-like or very elongated sample extents, a
# sample without a hit: detector 0 of the mixed scene sits at the sources, every 128th wave of its 2500 rays misses it)
ONE_PASS_APPLIES = {
    ("c1_single_lens", 0, None), ("double_gauss", 0, None), ("mixed_geometry", 1, "Orthographic"),
    ("arizona_eye", 0, "Orthographic"), ("asphere", 0, "Orthographic"), ("asphere", 1, None),
    ("hurb_slit_lens", 0, None),
}


@pytest.mark.parametrize("linebuf", ["1", "0"], ids=["line buffers", "plain tile kernel"])
@pytest.mark.parametrize("name", NAMES)
def test_one_pass_auto_extent_against_reference_images(name, linebuf):
    """`_auto_image_one_pass` (sample extent -> provisional tile grid -> 24-byte records -> exact binning) with both tile
    kernels, on the fixture's own rays, against the reference's image with an automatic extent."""
    g, RT = gpu_trace(name)
    calls = spy_one_pass(RT)
    applied = set()
    with forced(one_pass=1, linebuf=linebuf), ot.global_options.no_warnings():
        for di, det in enumerate(RT.detectors):
            for proj in projections(det):
                del calls[:]
                img = RT.detector_image(detector_index=di, projection_method=proj)
                assert len(calls) == 1
                if calls[0]:
                    applied.add((name, di, proj))
                check_against_fixture(img, g, f"det{di}/{proj}", f"{name} det{di} {proj} one pass")
    assert applied == {k for k in ONE_PASS_APPLIES if k[0] == name}, "where the one-pass form applies / declines"


@pytest.mark.parametrize("path", ["direct", "tiles"])
@pytest.mark.parametrize("name", NAMES)
def test_compact_hit_lists_against_reference_images(name, path):
    """Compact hit lists (valid hits only, 1024 interleaved pieces, `ot_detector_req.fill`) binned by the direct kernel and
    by the tile path (`tile_count / scatter / accum / reduce` walking every piece up to its fill)."""
    g, RT = gpu_trace(name)
    with forced(one_pass=1 << 60, compact=1, render_path=path), ot.global_options.no_warnings():
        for di, det in enumerate(RT.detectors):
            for proj in projections(det):
                img = RT.detector_image(detector_index=di, projection_method=proj)
                check_against_fixture(img, g, f"det{di}/{proj}", f"{name} det{di} {proj} compact {path}")


@pytest.mark.parametrize("path,linebuf", [("direct", "1"), ("tiles", "1"), ("tiles", "0")],
                         ids=["fuse_direct", "fuse_tiles_lb", "fuse_tiles"])
@pytest.mark.parametrize("name", NAMES)
def test_fused_user_extent_against_reference_images(name, path, linebuf):
    """`ot_detector_images` (hit search + binning in one pass) with the reference's user extent and single source:
    the LDS-hash direct kernel, the line-buffer tile kernel and the plain tile kernel."""
    g, RT = gpu_trace(name)
    with forced(render_path=path, linebuf=linebuf), ot.global_options.no_warnings():
        for di, det in enumerate(RT.detectors):
            key = f"det{di}/user"
            img = RT.detector_image(detector_index=di, extent=list(g[f"{key}/uext"]),
                                    projection_method=projections(det)[0], source_index=len(RT.ray_sources) - 1)
            check_against_fixture(img, g, key, f"{name} det{di} user extent {path} linebuf={linebuf}")


@pytest.mark.parametrize("K", [2, 3, 6])
@pytest.mark.parametrize("name", NAMES)
def test_multi_detector_tile_pass_against_reference_images(name, K):
    """The kernel `iterative_render` and the sharded forms run -- `fuse_tiles_kernel<false, 2 | 4 | 8, 1>`, K images per pass
    over the sections, then the `*_multi_kernel` second pass -- with automatic extents from the extent-only pass
    (`_auto_extents`): K copies of every (detector, projection) request in one call, each against the reference's image."""
    g, RT = gpu_trace(name)
    with forced(render_path="tiles"), ot.global_options.no_warnings():
        for di, det in enumerate(RT.detectors):
            for proj in projections(det):
                specs = [dict(detector_index=di, source_index=None, extent=None, projection_method=proj) for _ in range(K)]
                specs = RT._auto_extents(specs)
                imgs = RT._render_detectors(specs, [None] * K)
                assert len(imgs) == K
                for k, img in enumerate(imgs):
                    check_against_fixture(img, g, f"det{di}/{proj}", f"{name} det{di} {proj} image {k} of {K}")


@pytest.mark.parametrize("path", [None, "tiles"])
@pytest.mark.parametrize("name", NAMES)
def test_iterative_render_against_reference_images(name, path):
    """`iterative_render` (raytracer.py:1134-1279) with K = 3 positions and 3 chunks, every chunk tracing the fixture's
    recorded rays: the reference scales each chunk's image by rays_step / N and adds (:1257-1264), so three equal chunks of
    n rays with N = 3 n give the fixture's single image again -- per position.  Extents fixed by the first chunk."""
    g, RT = gpu_trace(name)
    n = int(g["N"])
    init = (g["p0"], g["s0"], g["pol0"] if not RT.no_pol else None, g["w0"], g["wl"])
    hn = g["hurb_normals"] if "hurb_normals" in g else None
    traced = []
    orig = RT.trace

    def inject(N, **kw):  # every chunk: the recorded bundle
        assert N == n
        traced.append(kw.get("_chunk"))
        return orig(N, _initial_rays=init, _hurb_normals=hn, _N_list=g["N_list"])

    RT.trace = inject
    old = ot.Raytracer.ITER_RAYS_STEP, ot.Raytracer.ITER_RENDER_ONLY
    ot.Raytracer.ITER_RAYS_STEP = n
    ot.Raytracer.ITER_RENDER_ONLY = False  # recorded rays go through the ray storage (a render-only trace generates its own:
    try:                                   # tests/test_gpu_render_only.py holds that form to this one, record by record)
        with forced(render_path=path), ot.global_options.no_warnings():
            for di, det in enumerate(RT.detectors):
                for proj in projections(det)[:2]:
                    pos = [list(det.pos)] * 3
                    del traced[:]
                    imgs = RT.iterative_render(3 * n, detector_index=di, pos=pos, projection_method=proj)
                    assert traced == [0, 1, 2], "one trace per chunk whatever the number of positions"
                    assert len(imgs) == 3
                    assert np.array_equal(RT._msgs, 3 * g["msgs"])
                    for k, img in enumerate(imgs):
                        check_against_fixture(img, g, f"det{di}/{proj}", f"{name} det{di} {proj} position {k}")
    finally:
        ot.Raytracer.ITER_RAYS_STEP, ot.Raytracer.ITER_RENDER_ONLY = old
        del RT.trace
