"""`convolve()` held to the behavioural tests of the reference (tests/test_convolve.py), restated with synthetic pictures and
point spread functions in place of the reference's preset files: a point PSF changes nothing, a point stays a point, white
stays white, zero images and zero PSFs, the sign and the value of the magnification, channel orthogonality, slicing,
padding, extents of shifted images and PSFs, flipped images behind a real (traced) PSF, mean values without normalisation.
(The value-level pins of the function are the fixtures of tests/test_gpu_convolve.py.)"""
import numpy as np
import pytest
import scipy.ndimage

import optrace_amd as ot
import scenes

pytestmark = pytest.mark.gpu

SRGB_TO_XYZ = np.array([[0.4124564, 0.3575761, 0.1804375], [0.2126729, 0.7151522, 0.0721750], [0.0193339, 0.1191920, 0.9503041]])


def to_linear(v):
    v = np.asarray(v, dtype=np.float64)
    return np.where(v <= 0.04045, v / 12.92, ((v + 0.055) / 1.055) ** 2.4)


def picture(s, n=240, seed=0):
    """A colourful test picture (blocks, a gradient, some noise), side lengths s."""
    rng = np.random.default_rng(seed)
    img = np.zeros((n, n, 3))
    yy, xx = np.mgrid[0:n, 0:n] / n
    img[..., 0] = 0.2 + 0.6 * xx
    img[..., 1] = 0.8 - 0.5 * yy
    img[..., 2] = 0.3 + 0.4 * ((xx * 6).astype(int) + (yy * 6).astype(int)) % 2
    img[n // 4:n // 2, n // 3:2 * n // 3] = [0.9, 0.1, 0.2]
    img += 0.05 * rng.random(img.shape)
    return ot.RGBImage(np.clip(img, 0, 1), s)


def disc_psf(d_um=60, n=121):
    """Uniform disc of diameter d (micrometres)."""
    half = 0.55 * d_um * 1e-3
    yy, xx = np.mgrid[-half:half:n * 1j, -half:half:n * 1j]
    return ot.GrayscaleImage((xx ** 2 + yy ** 2 <= (0.5e-3 * d_um) ** 2).astype(float), [2 * half, 2 * half])


def halo_psf(n=151, side=0.1):
    """A core with a ring around it."""
    yy, xx = np.mgrid[-1:1:n * 1j, -1:1:n * 1j]
    r = np.hypot(xx, yy)
    return ot.GrayscaleImage(np.exp(-(r / 0.12) ** 2) + 0.15 * np.exp(-((r - 0.6) / 0.06) ** 2), [side, side])


def edge_mean(a):
    return (a[0].mean() + a[-1].mean() + a[:, 0].mean() + a[:, -1].mean()) / 4


def test_point_psf_changes_nothing():
    """tests/test_convolve.py:131-154 (test_point_psf)."""
    image = picture([1, 1], seed=1)
    with ot.global_options.no_warnings():
        for padding in ["constant", "edge"]:
            for keep in [False, True]:
                for shape in [(200, 200), (201, 201)]:
                    res = ot.convolve(image, ot.GrayscaleImage(np.ones(shape), [1e-9, 1e-9]), keep_size=keep, padding_mode=padding)
                    d = res.data
                    dx, dy = (d.shape[1] - image.shape[1]) // 2, (d.shape[0] - image.shape[0]) // 2
                    inner = d if keep else d[dy:-dy, dx:-dx]
                    assert np.max(inner - image.data) ** 2.2 < 2e-5


def test_a_point_convolved_with_a_point_is_a_point():
    """tests/test_convolve.py:156-176 (test_point_image_point_psf)."""
    psf = np.zeros((301, 301))
    psf[151, 151] = 1
    img = np.repeat(psf[:, :, None], 3, axis=2)
    with ot.global_options.no_warnings():
        for padding in ["constant", "edge"]:
            for keep in [False, True]:
                d = ot.convolve(ot.RGBImage(img, [1, 1]), ot.GrayscaleImage(psf, [1e-9, 1e-9]), keep_size=keep,
                                padding_mode=padding).data
                assert d[d.shape[0] // 2 + 1, d.shape[1] // 2 + 1, 1] == pytest.approx(1)


@pytest.mark.parametrize("normalize", [True, False])
def test_white_stays_white(normalize):
    """tests/test_convolve.py:345-388 (test_white_balance): a colour PSF whose red part has high spatial frequencies and
    whose mean colour is white leaves the mean colour of a gray picture white -- also as one PSF per primary."""
    rimg = ot.RenderImage([-1e-6, 1e-6, -1e-6, 1e-6])
    rimg.render()
    rimg._data[:, :] = 0.5
    rimg._data[:2, :2, 2] = 1
    rimg._data[-2:, -2:, 2] = 0
    assert np.std(rimg._data[:, :, :3].mean(axis=(0, 1))) < 1e-6
    rimg._data[:, :, :3] = rimg._data[:, :, :3] @ SRGB_TO_XYZ.T
    gray = np.zeros((300, 300))
    gray[40:260:20] = 1
    gray[:, 50:250:25] = 1
    with ot.global_options.no_warnings():
        res = ot.convolve(ot.GrayscaleImage(gray, [1, 1]), rimg, cargs=dict(normalize=normalize))
        assert np.std(to_linear(res.data).mean(axis=(0, 1))) < 1e-6
        res = ot.convolve(picture([1, 1]), [rimg, rimg, rimg], cargs=dict(normalize=normalize), keep_size=True)
        assert np.std(to_linear(res.data).mean(axis=(0, 1))) < 1e-6


def test_zero_image_and_zero_psf():
    """tests/test_convolve.py:459-488."""
    with ot.global_options.no_warnings():
        assert ot.convolve(ot.RGBImage(np.zeros((200, 200, 3)), [5, 5]), halo_psf()).data.max() == 0
        assert ot.convolve(picture([5, 5]), ot.GrayscaleImage(np.zeros((200, 200)), [1, 1])).data.max() == 0
        empty = ot.RenderImage([-1, 1, -1, 1])
        empty.render()
        assert ot.convolve(ot.GrayscaleImage(picture([5, 5]).data[..., 0], [5, 5]), empty).data.max() == 0
        assert ot.convolve(picture([5, 5]), [empty, empty, empty]).data.max() == 0


def test_sign_and_value_of_the_magnification():
    """tests/test_convolve.py:490-513 (test_m_behavior): m = -1 flips the result, m = 2 equals a picture twice as large."""
    img = picture([5, 6])
    psf = halo_psf()
    psf.extent = [-0.05, 0.05, -0.15, 0.15]
    with ot.global_options.no_warnings():
        a, b = ot.convolve(img, psf, m=1), ot.convolve(img, psf, m=-1)
        assert np.all(np.array(a.s) == np.array(b.s))
        assert np.allclose(a.data, b.data[::-1, ::-1])
        c = ot.convolve(ot.RGBImage(img.data, [2 * img.s[0], 2 * img.s[1]]), psf, m=1)
        d = ot.convolve(img, psf, m=2)
        assert np.all(np.array(c.s) == np.array(d.s))
        assert np.allclose(c.data, d.data)


def test_no_bleeding_between_the_channels():
    """tests/test_convolve.py:515-541 (test_channel_orthogonality): a red picture, PSFs for green and blue that lack red, a
    zero PSF for red: nothing comes out, and normalize=False does not blow the rounding errors up."""
    rimg = ot.RenderImage([-1e-6, 1e-6, -1e-6, 1e-6])
    rimg.render()
    rimg._data[:, :] = 1
    rimg._data[:, :, 0] = 0
    rimg._data[:, :, :3] = rimg._data[:, :, :3] @ SRGB_TO_XYZ.T
    g = b = rimg.copy()
    rimg._data *= 0
    img0 = np.random.default_rng(3).random((1000, 1000, 3))
    img0[:, :, 1:] = 0
    with ot.global_options.no_warnings():
        res = ot.convolve(ot.RGBImage(img0, [1, 1]), [rimg, g, b], cargs=dict(normalize=False))
    assert abs(res.data.mean()) < 1e-6


def test_keep_size_returns_the_pictures_shape_and_place():
    """tests/test_convolve.py:543-572 (test_slicing)."""
    img0 = np.zeros((201, 201, 3))
    img0[100:150, 120:140] = 1
    with ot.global_options.no_warnings():
        for img in [ot.RGBImage(img0, [1, 1]), ot.GrayscaleImage(img0[:, :, 0], [1, 1])]:
            for padding in ["constant", "edge"]:
                for sp in [1.1e-2, 1e-2]:
                    for shape in [(100, 100), (100, 101), (101, 100), (101, 101)]:
                        res = ot.convolve(img, ot.GrayscaleImage(np.ones(shape), [sp, sp]), keep_size=True, padding_mode=padding)
                        assert np.all(np.array(res.s) == np.array(img.s)) and res.shape == img.shape
                        cm, cm2 = scipy.ndimage.center_of_mass(img.data), scipy.ndimage.center_of_mass(res.data)
                        assert abs(cm[0] - cm2[0]) < 0.7 and abs(cm[1] - cm2[1]) < 0.7


def test_padding_modes_and_values():
    """tests/test_convolve.py:574-615 (test_padding)."""
    psf = disc_psf(60)
    with ot.global_options.no_warnings():
        for img, black, white in [(ot.RGBImage(np.ones((100, 100, 3)), [1, 1]), [0, 0, 0], [1, 1, 1]),
                                  (ot.GrayscaleImage(np.ones((100, 100)), [1, 1]), 0, 1)]:
            for keep in [True, False]:
                dark = ot.convolve(img, psf, keep_size=keep, padding_mode="constant", padding_value=black)
                assert abs(edge_mean(dark.data) - 1) > 1e-5, "black padding darkens the edge"
                lit = ot.convolve(img, psf, keep_size=keep, padding_mode="constant", padding_value=white)
                assert abs(edge_mean(lit.data) - 1) < 1e-5, "white padding keeps it"
                if isinstance(img, ot.RGBImage):
                    flat = np.full((100, 100, 3), 0.3)
                    flat[:, :, 0] = 0
                    res = ot.convolve(ot.RGBImage(flat, [1, 1]), psf, keep_size=keep, padding_mode="edge", cargs=dict(normalize=False))
                    assert abs(edge_mean(res.data[:, :, 0])) < 1e-5
                    assert abs(edge_mean(res.data[:, :, 1:]) - 0.3) < 1e-5
                else:
                    res = ot.convolve(ot.GrayscaleImage(np.full((100, 100), 0.3), [1, 1]), psf, keep_size=keep,
                                      padding_mode="edge", cargs=dict(normalize=False))
                    assert abs(edge_mean(res.data) - 0.3) < 1e-5


def test_extents_of_shifted_pictures_and_psfs():
    """tests/test_convolve.py:617-644 (test_extent_shifting)."""
    img, psf = picture([2, 3]), halo_psf(side=0.1)
    with ot.global_options.no_warnings():
        res = ot.convolve(img, psf)
        assert res.extent[0] + res.extent[1] == 0 and res.extent[2] + res.extent[3] == 0
        img.extent = img.extent + np.array([-6, -6, 3, 3])
        psf.extent = psf.extent - np.array([-6, -6, 3, 3])
        res = ot.convolve(img, psf)
        assert res.extent[0] + res.extent[1] == 0 and res.extent[2] + res.extent[3] == 0
        psf.extent = psf.extent + np.array([-6, -6, 3, 3])
        res = ot.convolve(img, psf)
        assert (res.extent[0] + res.extent[1]) / 2 == -6 and (res.extent[2] + res.extent[3]) / 2 == 3


def test_flipped_pictures_behind_a_traced_psf():
    """tests/test_convolve.py:646-695 (test_image_flip): the PSF of a decentred ball lens (asymmetric), m = -1."""
    with ot.global_options.no_warnings():
        RT = ot.Raytracer([-10, 10, -10, 10, -100, 400])
        RT.add(ot.RaySource(ot.Point(), divergence="Lambertian", div_angle=3, pos=[0, 0, -20]))
        RT.add(ot.Lens(ot.SphericalSurface(r=4.99999999, R=5), ot.SphericalSurface(r=4.99999999, R=-5), d=10, pos=[0, -0.9, 0],
                       n=ot.RefractionIndex("Constant", n=1.3)))
        RT.add(ot.Detector(ot.RectangularSurface([10, 10]), pos=[0, 0, 18.987]))
        RT.trace(500_000)
        psf = RT.detector_image()
        data = np.zeros((1001, 1001))
        data[500, 500] = 1
        res = ot.convolve(ot.GrayscaleImage(data, [0.5, 0.5]), psf, m=-1, keep_size=True)
        cm = np.array(scipy.ndimage.center_of_mass(res.data[:, :, 0])) / res.shape[0]
        assert cm[0] > 0.53, "the blur lies above the centre"
        data[0, 0] = 1
        res = ot.convolve(ot.GrayscaleImage(data, [0.5, 0.5]), psf, m=-1)
        cm = np.array(scipy.ndimage.center_of_mass(res.data[:, :, 0])) / res.shape[0]
        assert cm[0] > 0.7 and cm[1] > 0.7, "the second point, lower left, appears upper right"


def test_means_survive_without_normalisation():
    """tests/test_convolve.py:697-724 (test_unnormalized_color_and_grayscale): the PSF is normalised whatever `normalize`."""
    psf = disc_psf(60)
    with ot.global_options.no_warnings():
        for rgb in [[0, 1, 0], [0.2, 0.3, 0.5], [0.1, 0.1, 0.1]]:
            img = ot.RGBImage(np.tile(rgb, (100, 100, 1)).astype(float), [1, 1])
            res = ot.convolve(img, psf, keep_size=True, padding_mode="edge", cargs=dict(normalize=False))
            for c in range(3):
                assert res.data[:, :, c].mean() == pytest.approx(rgb[c], abs=1e-5)
            assert to_linear(img.data).sum() == pytest.approx(to_linear(res.data).sum(), abs=0.01)
        for gv in [0.3, 0., 1.0, 0.297]:
            res = ot.convolve(ot.GrayscaleImage(np.full((100, 100), gv), [1, 1]), psf, keep_size=True, padding_mode="edge",
                              cargs=dict(normalize=False))
            assert res.data.mean() == pytest.approx(gv, abs=1e-5)
