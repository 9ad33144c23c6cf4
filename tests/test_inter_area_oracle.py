"""The CPU restatement of OpenCV's INTER_AREA resize (oracle/inter_area.py; OpenCV 4.x modules/imgproc/src/resize.cpp) against
the properties its algorithm states -- there is no OpenCV binary in this image to pin it to (its header says so)."""
import sys
import pathlib

import numpy as np
import pytest

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from oracle import inter_area as ia


def test_dispatch_of_cv_resize():
    assert ia.mode((945, 945), (315, 315)) == "fast"      # RenderImage.get: always integer ratios (render_image.py:174)
    assert ia.mode((945, 2835), (189, 567)) == "fast"
    assert ia.mode((400, 300), (150, 100)) == "area"      # 2.67 x 3: one fractional ratio is enough
    assert ia.mode((401, 401), (200, 200)) == "area"
    assert ia.mode((100, 100), (100, 100)) == "fast"      # identity: ratio 1
    assert ia.mode((100, 100), (150, 50)) == "linear"     # one axis enlarged: the linear resampler's area mode, both axes
    assert ia.mode((100, 100), (250, 250)) == "linear"


@pytest.mark.parametrize("fact", [1, 3, 5, 7, 9, 15, 21, 27, 35, 45, 63, 105, 135, 189, 315])
def test_integer_ratios_are_block_means_times_a_float32_reciprocal(fact):
    rng = np.random.default_rng(fact)
    n = 945 // fact
    src = rng.uniform(0, 3, (fact * min(n, 12), fact * min(n, 9), 4))
    got = ia.resize_inter_area(src, (src.shape[1] // fact, src.shape[0] // fact))
    blocks = src.reshape(src.shape[0] // fact, fact, src.shape[1] // fact, fact, 4).mean(axis=(1, 3))
    if ia.mode((src.shape[1], src.shape[0]), (src.shape[1] // fact, src.shape[0] // fact)) == "fast":
        # resizeAreaFast_: `float scale = 1.f / area`, the block sum is multiplied by that float
        np.testing.assert_allclose(got, blocks * (fact * fact) * float(np.float32(1.0 / (fact * fact))), rtol=1e-12)
    # (1 / (n / 945) misses the integer by an ulp for some of the sizes -- 945 -> 9 is one: cv::resize's test
    # `abs(scale_x - iscale_x) < DBL_EPSILON` then fails and the general area path runs, float32 shares of 1 / fact each)
    np.testing.assert_allclose(got, blocks, rtol=2e-7)


@pytest.mark.parametrize("n_in,n_out", [(401, 200), (300, 113), (64, 63), (1000, 7), (37, 36)])
def test_fractional_reduction_is_the_overlap_average_with_float32_shares(n_in, n_out):
    tab = ia._area_tab(n_in, n_out, n_in / n_out)
    m = ia._matrix(tab, n_out, n_in)
    exact = ia.exact_area_matrix(n_in, n_out)
    assert np.abs(m - exact).max() < 2e-7 + 1e-3 / (n_in / n_out)  # shares below 1e-3 of a pixel are dropped (the 1e-3 tests)
    np.testing.assert_allclose(m.sum(axis=1), 1.0, atol=2e-3)      # ... which is all a row sum can miss
    assert (m >= 0).all()
    # every source pixel is used, no destination pixel reaches beyond its interval by more than one pixel
    assert (m.sum(axis=0) > 0).all()
    for d in range(n_out):
        nz = np.nonzero(m[d])[0]
        assert nz.min() >= np.floor(d * n_in / n_out) and nz.max() <= np.ceil((d + 1) * n_in / n_out)
    img = np.random.default_rng(1).uniform(0, 1, (n_in, n_in))
    out = ia.resize_inter_area(img, (n_out, n_out))
    assert abs(out.sum() * (n_in / n_out) ** 2 - img.sum()) < 5e-3 * img.sum()
    np.testing.assert_allclose(ia.resize_inter_area(np.full((n_in, n_in), 2.5), (n_out, n_out)), 2.5, rtol=5e-3)


def test_enlarging_takes_the_linear_resampler_in_area_mode():
    """Its taps are again the overlap shares: a destination pixel lies inside one source pixel (tap 1) or straddles one
    boundary (the shares on either side) -- `fx = (dx + 1) - (sx + 1) * inv_scale`, kept if positive.  An integer enlargement
    therefore replicates pixels."""
    m = ia._linear_area_taps(4, 8)
    np.testing.assert_allclose(m, np.kron(np.eye(4), np.ones((2, 1))))
    m = ia._linear_area_taps(2, 3)  # destination pixel 1 covers [2/3, 4/3): half of source pixel 0, half of pixel 1
    np.testing.assert_allclose(m, [[1, 0], [0.5, 0.5], [0, 1]], atol=1e-7)
    for n_in, n_out in ((10, 25), (7, 8), (100, 101), (33, 80)):
        m = ia._linear_area_taps(n_in, n_out)
        np.testing.assert_allclose(m.sum(axis=1), 1.0, atol=2e-7)
        assert (m >= 0).all() and (np.count_nonzero(m, axis=1) <= 2).all()
        np.testing.assert_allclose(m, ia.exact_area_matrix(n_in, n_out), atol=3e-7)  # float32 taps
        np.testing.assert_allclose(ia.resize_inter_area(np.full((n_in, n_in), 1.5), (n_out, n_out)), 1.5, rtol=3e-7)
    # mixed: one axis reduced, one enlarged -> BOTH through the two-tap resampler (cv::resize leaves the area branch as a
    # whole): the reduced axis then takes two source pixels per destination pixel, not the average over its interval
    mx, my, post = ia.axis_matrices((100, 150), (150, 50))  # x: 100 -> 150, y: 150 -> 50
    assert post == 1.0 and (np.count_nonzero(mx, axis=1) <= 2).all() and (np.count_nonzero(my, axis=1) <= 2).all()
    assert np.abs(my - ia.exact_area_matrix(150, 50)).max() > 0.2  # taps (1/3, 2/3) on two rows instead of three thirds
