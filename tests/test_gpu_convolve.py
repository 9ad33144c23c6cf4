"""convolve() (optrace_amd/convolve.py) against fixtures computed by the reference's own convolve (convolve.py:49-454)
on the same synthetic images and PSFs (tests/golden/convolve.npz, generator: tests/golden/generate_golden.py convolve).

The fixture cases use PSFs with the pixel pitch of the image, where the reference's cv2.resize(INTER_AREA) call is the
identity: everything else of the function -- gamma removal, PSF normalisation, colour PSFs through XYZ -> linear sRGB,
padding modes, flipping and scaling by m, full / keep_size slicing, the result extent, the final gamut mapping with its
rendering intents -- is pinned to the reference at 1e-9.  The area resize for other pitch ratios is checked by its
defining properties (sum preserved, block means for integer ratios) and by the reference's own behavioural tests for
it, restated below with the reference's tolerances (analytic Gaussian through a PSF of half the pitch; independence of the
PSF's resolution); it is not pinned by a reference VALUE (no OpenCV in the build container)."""
import numpy as np
import pytest
import torch

import optrace_amd as ot
from convolve_cases import convolve_cases, build_convolve_inputs
from helpers import load

pytestmark = pytest.mark.gpu
CASES = convolve_cases()


@pytest.mark.parametrize("name", sorted(CASES))
def test_convolve_matches_reference(name):
    g = load("convolve.npz")
    case = CASES[name]
    assert np.array_equal(g[f"{name}/img"], case["img"]) and np.array_equal(g[f"{name}/psf"], case["psf"]), \
        "fixture inputs = regenerated inputs"
    img, psf = build_convolve_inputs(ot, case)
    with ot.global_options.no_warnings():
        res = ot.convolve(img, psf, m=case["m"], **case["kwargs"])
    assert isinstance(res, ot.GrayscaleImage if case["psf_kind"] == "gray" and case["img"].ndim == 2 else ot.RGBImage)
    d = res.data
    assert tuple(d.shape) == tuple(g[f"{name}/shape"])
    np.testing.assert_allclose(res.extent, g[f"{name}/extent"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(d[1::4, 2::4], g[f"{name}/grid4"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(d.sum(axis=(0, 1)), g[f"{name}/sum"], rtol=1e-9)
    np.testing.assert_allclose(d.max(axis=(0, 1)), g[f"{name}/max"], rtol=0, atol=1e-9)


def test_area_resize_properties():
    """The PSF resize: row sums 1 (the average of a constant is the constant), total preserved after the `keep`
    factor, block means for integer ratios, identity for equal sizes."""
    from optrace_amd.convolve import _area_weights
    dev = torch.device("cuda")
    for n_in, n_out in [(61, 61), (120, 40), (61, 37), (50, 73), (200, 7)]:
        W = _area_weights(n_in, n_out, dev)
        assert torch.allclose(W.sum(dim=1), torch.ones(n_out, dtype=torch.float64, device=dev), atol=1e-13)
        assert torch.allclose(W.sum(dim=0) * (n_in / n_out), torch.ones(n_in, dtype=torch.float64, device=dev), atol=1e-12)
    assert torch.equal(_area_weights(61, 61, dev), torch.eye(61, dtype=torch.float64, device=dev))
    x = torch.rand(120, dtype=torch.float64, device=dev)
    assert torch.allclose(_area_weights(120, 40, dev) @ x, x.view(40, 3).mean(dim=1), atol=1e-14)


def test_convolve_finer_psf_keeps_power_and_position():
    """A PSF sampled three times finer than the image (the usual case, convolve.py:292-294 warns about the opposite):
    a single bright pixel becomes the PSF, with the PSF's sum, at the position image centre + PSF centre."""
    n = 101
    data = np.zeros((n, n))
    data[50, 50] = 1.0
    img = ot.GrayscaleImage(data, [1.0, 1.0])                     # pitch 0.01
    pn = 151                                                      # pitch 0.01 / 3
    yy, xx = np.mgrid[0:pn, 0:pn]
    blob = np.exp(-((xx - 75.0) ** 2 + (yy - 75.0) ** 2) / 300.0)
    psf = ot.GrayscaleImage(blob / blob.max(), extent=[0.2 - 0.25, 0.2 + 0.25, -0.25, 0.25])  # centred at x = +0.2
    with ot.global_options.no_warnings():
        res = ot.convolve(img, psf, cargs=dict(normalize=False))
    d = res.data
    lin = np.where(d <= 0.04045, d / 12.92, ((d + 0.055) / 1.055) ** 2.4)
    assert abs(lin.sum() - 1.0) < 1e-9, "normalised PSF: the pixel's linear value is spread, not changed"
    ys, xs = np.mgrid[0:d.shape[0], 0:d.shape[1]]
    cx = (lin * xs).sum() / lin.sum() / (d.shape[1] - 1) * (res.extent[1] - res.extent[0]) + res.extent[0]
    cy = (lin * ys).sum() / lin.sum() / (d.shape[0] - 1) * (res.extent[3] - res.extent[2]) + res.extent[2]
    assert abs(cx - 0.2) < 2e-3 and abs(cy) < 2e-3


@pytest.mark.parametrize("sz", [400, 401, 1999, 2000])
def test_coordinate_value_correctness(sz):
    """After the reference's test_coordinate_value_correctness (tests/test_convolve.py:178-224), same numbers and
    tolerances: a Gaussian image convolved with a Gaussian PSF of HALF its pixel pitch -- the PSF goes through the area
    resize with a ratio of two, the case no fixture covers -- must give the analytic Gaussian of the summed variances on the
    result's own coordinates (mean absolute deviation 5e-5 after normalisation), symmetric under transposition to 1e-12;
    odd and even, small and large pixel counts."""
    from optrace_amd.image import srgb_linear_to_srgb, srgb_to_srgb_linear

    def gaussian(d):
        sig = 0.175  # sigma approximating the zeroth order of an Airy disc
        ds = 5 * sig
        Y, X = np.mgrid[-ds:ds:sz * 1j, -ds:ds:sz * 1j]
        Z = np.exp(-(X ** 2 + Y ** 2) / 2 / sig ** 2)
        return np.repeat(Z[:, :, np.newaxis], 3, axis=2), [2 * ds * d / 1000, 2 * ds * d / 1000]

    psf, s_psf = gaussian(1)
    psf = srgb_linear_to_srgb(psf)[:, :, 0]
    img, s_img = gaussian(2)
    img = srgb_linear_to_srgb(img)
    with ot.global_options.no_warnings():
        res = ot.convolve(ot.RGBImage(img, s_img), ot.GrayscaleImage(psf, s_psf))
    img2, s2 = res.data, res.s
    img2 = srgb_to_srgb_linear(img2)[:, :, 0]
    Y, X = np.mgrid[-s2[0] / 2:s2[0] / 2:img2.shape[1] * 1j, -s2[1] / 2:s2[1] / 2:img2.shape[0] * 1j]
    Z = np.exp(-3.265306122449e6 * (X ** 2 + Y ** 2))
    Z /= np.max(Z)
    img2 = img2 / np.max(img2)
    diff = img2 - Z
    assert np.mean(np.abs(diff)) < 5e-5
    assert np.mean(np.abs(diff - diff.T)) < 1e-12


def _chart(n: int) -> np.ndarray:
    """Synthetic stand-in for the reference's ETDRS chart (an image file): white bars and blocks of several sizes on
    black, sampled at n x n from one continuous description so that every resolution shows the same object."""
    y, x = (np.mgrid[0:n, 0:n] + 0.5) / n
    img = np.zeros((n, n))
    for k, (x0, w) in enumerate([(0.08, 0.16), (0.30, 0.10), (0.46, 0.06), (0.58, 0.04), (0.68, 0.025), (0.75, 0.015)]):
        img[(x > x0) & (x < x0 + w) & (y > 0.1) & (y < 0.45)] = 1.0
        img[(y > x0) & (y < x0 + w) & (x > 0.55 + 0.0 * k) & (x < 0.92) & (y > 0.55)] = 1.0
    img[(np.hypot(x - 0.25, y - 0.72) < 0.13) & (np.hypot(x - 0.25, y - 0.72) > 0.07)] = 1.0
    return img


@pytest.mark.parametrize("s_psf", [0.01, 0.1, 0.99])
def test_size_consistency(s_psf):
    """After the reference's test_size_consistency (tests/test_convolve.py:391-416): the same object convolved with the same
    PSF must not depend on the resolution the PSF is sampled at -- PSFs of 2000, 1999, 400, 399 pixels a side against an
    image of fixed resolution, so the area resize runs with ratios from below one to forty, odd and even.  Tolerances of the
    reference: 1e-3 between the two finest, 5e-3 down to 400 pixels."""
    img = ot.GrayscaleImage(_chart(500), [1, 1])

    def psf_at(res):
        ds = s_psf / 2
        Y, X = np.mgrid[-ds:ds:res * 1j, -ds:ds:res * 1j]
        sig = s_psf / 8
        core = np.exp(-(X ** 2 + Y ** 2) / 2 / sig ** 2)
        halo = 0.1 * np.exp(-(np.hypot(X, Y) - 0.3 * s_psf) ** 2 / 2 / (sig / 3) ** 2)  # ring: structure finer than the core
        return ot.GrayscaleImage((core + halo) / (core + halo).max(), [s_psf, s_psf])

    ref = None
    with ot.global_options.no_warnings():
        for i, res in enumerate([2000, 1999, 400, 399]):
            d = ot.convolve(img, psf_at(res), keep_size=True).data  # (on the image's own grid, whatever the PSF's)
            if ref is None:
                ref = d
                continue
            assert d.shape == ref.shape
            dev = np.mean(np.abs(d - ref))
            assert dev < (1e-3 if i == 1 else 5e-3), (res, dev)


def test_convolve_argument_errors():
    img = ot.GrayscaleImage(np.full((60, 60), 0.5), [1, 1])
    psf = ot.GrayscaleImage(np.full((60, 60), 0.5), [0.5, 0.5])
    rgb = ot.RGBImage(np.full((60, 60, 3), 0.5), [1, 1])
    ri = ot.RenderImage([-0.25, 0.25, -0.25, 0.25])
    with pytest.raises(TypeError):
        ot.convolve(img, psf, m="1")
    with pytest.raises(ValueError):
        ot.convolve(img, psf, m=0)
    with pytest.raises(TypeError):
        ot.convolve(rgb, ri)                 # a colour image needs three colour PSFs
    with pytest.raises(TypeError):
        ot.convolve(img, [ri, ri, ri])       # a grey image needs one
    with pytest.raises(ValueError):
        ot.convolve(rgb, psf, padding_value=[0.1, 0.2])
    with pytest.raises(ValueError):
        ot.convolve(img, psf, padding_value=-1.0)
    with pytest.raises(ValueError):
        ot.convolve(img, ot.GrayscaleImage(np.full((60, 60), 0.5), [3, 3]))   # PSF more than twice the image
    with pytest.raises(ValueError):
        ot.convolve(img, ot.GrayscaleImage(np.full((40, 60), 0.5), [0.5, 0.5]))  # fewer than 50 PSF pixels
    with pytest.raises(ValueError):
        ot.convolve(ot.GrayscaleImage(np.full((40, 60), 0.5), [1, 1]), psf)
