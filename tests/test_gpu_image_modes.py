"""RenderImage.get on the GPU (ot_image_convert) against the reference's RenderImage.get at full resolution
(tests/golden/image_modes.npz), plus the bin-joining down-scaling by its defining properties."""
import numpy as np
import pytest

import optrace_amd as ot
from helpers import load, assert_close

pytestmark = pytest.mark.gpu
MODES = ot.RenderImage.image_modes


@pytest.fixture(scope="module")
def gm():
    return load("image_modes.npz")


def make_image(gm, name):
    shape = tuple(gm[f"{name}/shape"])
    img = ot.RenderImage(extent=gm[f"{name}/extent"])
    data = np.zeros(shape)
    data[gm[f"{name}/iy"], gm[f"{name}/ix"]] = gm[f"{name}/xyzw"]
    img._data = data
    return img


@pytest.mark.parametrize("name", ["double_gauss", "mixed_geometry"])
@pytest.mark.parametrize("mode", MODES)
def test_get_matches_reference(gm, name, mode):
    img = make_image(gm, name)
    iy, ix = gm[f"{name}/iy"], gm[f"{name}/ix"]
    variants = [("", {})]
    if mode == "sRGB (Perceptual RI)":
        variants += [("|Lth", dict(L_th=0.02)), ("|cs", dict(chroma_scale=0.6))]
    for tag, kw in variants:
        res = img.get(mode, 945, **kw)
        assert type(res).__name__ == ("RGBImage" if mode.startswith("sRGB") else "ScalarImage")
        assert res.quantity == mode and np.array_equal(res.extent, img.extent)
        d = res._data
        ref, bg = gm[f"{name}/{mode}{tag}"], gm[f"{name}/{mode}{tag}/bg"]
        if mode == "Hue (CIELUV)":  # angle: compare on the circle, ignore achromatic pixels (hue undefined)
            chroma = img.get("Chroma (CIELUV)", 945)._data[iy, ix]
            sel = chroma > 1e-6
            diff = np.abs((d[iy, ix][sel] - ref[sel] + 180) % 360 - 180)
            assert diff.max() < 1e-6
        else:
            assert_close(d[iy, ix], ref, rtol=1e-9, atol=1e-12, what=f"{name} {mode}{tag}")
        mask = np.ones(d.shape[:2], dtype=bool)
        mask[iy, ix] = False
        assert np.allclose(d[mask], bg, atol=1e-12), "background pixels"


def test_bin_joining_conserves_power_and_shape(gm):
    img = make_image(gm, "mixed_geometry")
    full = img.get("Irradiance", 945)
    P = full._data.sum() * img.Apx
    assert abs(P - img.power()) < 1e-12 * img.power()
    for N, side in [(315, 315), (300, 315), (189, 189), (100, 105), (10, 9), (1, 1)]:
        small = img.get("Irradiance", N)
        assert small.shape[0] == side * (img.shape[0] // 945) and small.shape[1] == side * (img.shape[1] // 945)
        assert abs(small._data.sum() * small.Apx - img.power()) < 1e-10 * img.power()
        f = 945 // side
        ref = img._data[:, :, 3].reshape(small.shape[0], f, small.shape[1], f).mean(axis=(1, 3)) / img.Apx
        assert_close(small._data, ref, rtol=1e-12, atol=1e-18, what=f"N={N}")
    ill = img.get("Illuminance", 315)
    assert abs(ill._data.sum() * ill.Apx - img.luminous_power()) < 1e-10 * img.luminous_power()
    with pytest.raises(ValueError):
        img.get("Irradiance", 0)
    with pytest.raises(ValueError):
        img.get("nope")
    rgb = img.get("sRGB (Absolute RI)", 189)
    assert rgb.shape[2] == 3 and rgb._data.min() >= 0 and rgb._data.max() <= 1


@pytest.mark.parametrize("limit", [3.0, 12.0])
def test_rayleigh_filter_matches_reference(gm, limit):
    """limit= : the Airy-disc convolution (render_image.py:257-296).  The reference convolves by FFT, we sum
    directly: agreement to FFT round-off relative to the image maximum."""
    img = ot.RenderImage(extent=gm["filter/ext0"])
    img.render(gm["filter/ph"], gm["filter/w"], gm["filter/wl"], limit=limit)
    assert img.limit == limit
    assert_close(img.extent, gm[f"filter/{limit}/extent"], rtol=1e-12, what="extent incl. the 2.7*limit margin")
    assert img._data.shape == tuple(gm[f"filter/{limit}/shape"])
    mx = gm[f"filter/{limit}/max"]
    got = img._data[3::7, 2::7, :]
    ref = gm[f"filter/{limit}/grid7"]
    assert np.max(np.abs(got - ref) / mx) < 1e-11
    assert abs(img.power() - float(gm[f"filter/{limit}/power"])) < 1e-9 * img.power()
    assert img._data.min() >= 0
    # the filter spreads but conserves power away from the borders
    plain = ot.RenderImage(extent=gm["filter/ext0"])
    plain.render(gm["filter/ph"], gm["filter/w"], gm["filter/wl"])
    assert abs(plain.power() - img.power()) < 1e-6 * plain.power()
    assert img._data[..., 3].max() < plain._data[..., 3].max()


def test_large_resolution_kernels_go_through_the_fft():
    """A resolution limit of many pixels (a tiny extent) needs a kernel beyond the direct convolution's 137 x 137 taps:
    the FFT path gives what the direct kernel gives where both apply, and handles the reference's own case
    (tests/test_tracer.py:1083: limit=4 with a 0.02 mm extent)."""
    import torch
    import optrace_amd as ot
    from optrace_amd.render_image import RenderImage
    rng = np.random.default_rng(3)
    img = RenderImage(extent=[-1, 1, -1, 1])
    img._data = np.zeros((945, 945, 4))
    d = rng.random((945, 945, 4)) * (rng.random((945, 945, 1)) < 0.02)
    img._data = d
    img._limit = 2000 * 2 / 945 * 20          # first zero 20 pixels out: ps = 54, the direct kernel
    a = RenderImage(extent=[-1, 1, -1, 1]); a._data = d.copy(); a._limit = img._limit
    a._apply_rayleigh_filter()
    b = RenderImage(extent=[-1, 1, -1, 1]); b._data = d.copy(); b._limit = img._limit
    old = RenderImage._DIRECT_PSF_MAX
    try:
        RenderImage._DIRECT_PSF_MAX = 0         # force the FFT
        b._apply_rayleigh_filter()
    finally:
        RenderImage._DIRECT_PSF_MAX = old
    assert np.abs(a._data - b._data).max() < 1e-11 * a._data.max()
    assert abs(b._data[..., 3].sum() / d[..., 3].sum() - 1) < 0.05  # power stays (edges lose a little)
    c = RenderImage(extent=[-1, 1, -1, 1]); c._data = d.copy(); c._limit = 2000 * 2 / 945 * 120   # ps = 324
    c._apply_rayleigh_filter()
    assert np.all(np.isfinite(c._data)) and c._data.min() >= 0 and c._data[..., 3].sum() > 0.5 * d[..., 3].sum()


@pytest.mark.parametrize("k", [1, 0.3, 5])
def test_resolution_filter_turns_a_point_into_an_airy_disc(k):
    """After the reference's test_r_image_filter (tests/test_image.py:216-253): a point rendered with a resolution limit
    becomes an Airy disc whose radial centroid is 1.10861 / 3.8317 of the limit, for square and elongated images and
    for limits of a few and of very many pixels (direct kernel and FFT)."""
    import optrace_amd as ot
    p = np.zeros((1000, 3))
    w = np.ones(1000)
    wl = np.full(1000, 500)
    r0 = 1e-4  # smaller than the resolution limit
    img = ot.RenderImage([-r0, r0, -k / 2 * r0, k / 2 * r0])
    for limit in (0.3, 20):
        img.render(p, w, wl, limit=limit)
        irr = img._data[:, :, 3]
        ny, nx = irr.shape[:2]
        x = np.linspace(0, img.extent[1], nx // 2 + 1)
        cut = irr[ny // 2 + 1, nx // 2:]
        centroid = 3.8317 * np.sum(x * cut) / np.sum(cut)
        assert abs(centroid / 1.10861 - limit / 1000) < 0.0002, (k, limit)
    with pytest.raises(RuntimeError):
        ot.RenderImage([-1, 1, -1, 1], projection="abc").render(limit=1)


def test_render_and_rescale_keep_power_and_shape():
    """After the reference's test_render_image_render_and_rescaling (tests/test_image.py:108-172): power and luminous
    power of rendered random hits, with and without the resolution filter and for four side ratios; every image size
    keeps the integrated power, the side ratio, and picks the nearest of SIZES; extents that are too small or too
    elongated are adjusted; impossible sizes raise."""
    import optrace_amd as ot
    rng = np.random.default_rng(8)
    for limit in (None, 20):
        for ratio in (1 / 6, 0.38, 1, 5):
            img = ot.RenderImage([-1, 1, -1 * ratio, 1 * ratio])
            N = 10_000
            p = np.zeros((N, 3))
            p[:, 0] = rng.uniform(img.extent[0], img.extent[1], N)
            p[:, 1] = rng.uniform(img.extent[2], img.extent[3], N)
            w = rng.uniform(1e-9, 1, N)
            wl = rng.uniform(380, 780, N)
            img.render(p, w, wl, limit=limit)
            P0, L0 = img.power(), img.luminous_power()
            tol = 1e-6 if limit is None else 2e-2   # the filter loses what it spreads over the image border
            assert abs(P0 / np.sum(w) - 1) < tol
            assert L0 > 0
            ratio_act = img.shape[1] / img.shape[0]
            for Npx in [*ot.RenderImage.SIZES, *rng.integers(1, ot.RenderImage.SIZES[-1], 3)]:
                for Q, mode in ((P0, "Irradiance"), (L0, "Illuminance")):
                    data = img.get(mode, int(Npx))
                    siz = ot.RenderImage.SIZES
                    near = siz[int(np.argmin(np.abs(Npx - np.array(siz))))]
                    assert near == (data.shape[1] if ratio_act < 1 else data.shape[0])
                    assert abs(Q / np.sum(data.data) / data.Apx - 1) < 1e-6
                    assert abs(ratio_act - data.shape[1] / data.shape[0]) < 1e-12
    img.render()
    for bad in (ot.RenderImage.MAX_IMAGE_SIDE * 1.2, -2, 0):
        with pytest.raises(ValueError):
            img.get("Irradiance", N=bad)
    for ext in ([0, ot.RenderImage.EPS / 2, 0, ot.RenderImage.EPS / 2], [0, 1, 0, 1.2 * ot.RenderImage.MAX_IMAGE_RATIO],
                [0, 1.2 * ot.RenderImage.MAX_IMAGE_RATIO, 0, 1]):
        im = ot.RenderImage(extent=ext)
        before = im.extent.copy()
        im.render()
        assert not np.all(im.extent == before)
