"""convolve() (optrace_amd/convolve.py) against fixtures computed by the reference's own convolve (convolve.py:49-454)
on the same synthetic images and PSFs (tests/golden/convolve.npz, generator: tests/golden/generate_golden.py convolve).

The fixture cases use PSFs with the pixel pitch of the image, where the reference's cv2.resize(INTER_AREA) call is the
identity: everything else of the function -- gamma removal, PSF normalisation, colour PSFs through XYZ -> linear sRGB,
padding modes, flipping and scaling by m, full / keep_size slicing, the result extent, the final gamut mapping with its
rendering intents -- is pinned to the reference at 1e-9.  The area resize for other pitch ratios is checked by its
defining properties (sum preserved, block means for integer ratios); that part is not pinned by a reference value."""
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
