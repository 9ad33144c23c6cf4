"""TEST INFRASTRUCTURE -- CPU restatement of `cv2.resize(src, dsize, interpolation=cv2.INTER_AREA)` for float64 images.

The reference calls it in two places of the path: `RenderImage.get` joins bins with it (image/render_image.py:174, always
integer ratios 945 / {1, 3, 5, ..., 315}) and `convolve()` brings the PSF to the image's pixel pitch with it
(convolve.py:372, any ratio).  OpenCV is a third-party dependency of the reference (`opencv-python-headless`, unpinned in
pyproject.toml:12) and is absent from this image, so this module restates its PUBLISHED algorithm -- OpenCV 4.x,
modules/imgproc/src/resize.cpp: `cv::resize` (dispatch), `resizeAreaFast_` (integer ratios), `computeResizeAreaTab` +
`resizeArea_` (fractional reduction), and the "area mode" of the linear resampler that INTER_AREA falls to as soon as one
axis is enlarged (`resizeGeneric_` with `HResizeLinear` / `VResizeLinear`) -- for the depth the reference uses (CV_64F:
working type double, coefficient type float).  PARITY UNPINNED: there is no OpenCV binary here to check these functions
against and the reference's own tests hold no vectors for this call; what is pinned are the properties the algorithm
states (tests/test_inter_area_oracle.py).  Only tests/ import this file.

Float32 enters where OpenCV stores coefficients as `float`: the reciprocal block area of the integer path, the overlap
shares `alpha` of the fractional path, the two taps of the linear path.  They are kept, so that the size of the product's
deviation from the real call (it uses exact f64 weights) can be stated.
"""
from __future__ import annotations

import math

import numpy as np

DBL_EPSILON = np.finfo(np.float64).eps


def _f32(x: float) -> float:
    return float(np.float32(x))


def mode(ssize: tuple, dsize: tuple) -> str:
    """Which branch of cv::resize serves (src (W, H) -> dst (W', H')) with INTER_AREA: "fast" | "area" | "linear"."""
    (sw, sh), (dw, dh) = ssize, dsize
    inv_scale_x, inv_scale_y = dw / sw, dh / sh
    scale_x, scale_y = 1.0 / inv_scale_x, 1.0 / inv_scale_y
    if scale_x >= 1 and scale_y >= 1:  # resize.cpp: `if( interpolation == INTER_AREA && scale_x >= 1 && scale_y >= 1 )`
        iscale_x, iscale_y = int(round(scale_x)), int(round(scale_y))  # saturate_cast<int>: round to nearest
        fast = abs(scale_x - iscale_x) < DBL_EPSILON and abs(scale_y - iscale_y) < DBL_EPSILON
        return "fast" if fast else "area"
    return "linear"


def _area_tab(ssize: int, dsize: int, scale: float) -> list:
    """computeResizeAreaTab: (destination index, source index, float32 share) triples along one axis."""
    tab = []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = math.ceil(fsx1), math.floor(fsx2)
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        if sx1 - fsx1 > 1e-3:
            tab.append((dx, sx1 - 1, _f32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            tab.append((dx, sx, _f32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            tab.append((dx, sx2, _f32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
    return tab


def _matrix(tab: list, n_out: int, n_in: int) -> np.ndarray:
    m = np.zeros((n_out, n_in))
    for d, s, a in tab:
        m[d, s] += a
    return m


def _linear_area_taps(ssize: int, dsize: int) -> np.ndarray:
    """The two-tap rows of the linear resampler in area mode (resize.cpp, `area_mode` branch of the coefficient loop,
    then HResizeLinear: destination pixels from xmax on copy their source pixel).  -> (dsize, ssize) matrix."""
    inv_scale = dsize / ssize
    scale = 1.0 / inv_scale
    m = np.zeros((dsize, ssize))
    xmax = dsize
    rows = []
    for dx in range(dsize):
        sx = math.floor(dx * scale)
        fx = _f32((dx + 1) - (sx + 1) * inv_scale)
        fx = 0.0 if fx <= 0 else fx - math.floor(fx)
        if sx < 0:
            fx, sx = 0.0, 0
        if sx + 1 >= ssize:  # sx + ksize2 >= ssize.width, ksize2 = 1
            xmax = min(xmax, dx)
            if sx >= ssize - 1:
                fx, sx = 0.0, ssize - 1
        rows.append((sx, _f32(1.0 - _f32(fx)), _f32(fx)))
    for dx, (sx, a0, a1) in enumerate(rows):
        if dx < xmax:
            m[dx, sx] += a0
            m[dx, sx + 1] += a1
        else:
            m[dx, sx] += 1.0
    return m


def axis_matrices(ssize: tuple, dsize: tuple) -> tuple:
    """(Mx (W', W), My (H', H), post factor): the resize is `My @ src @ Mx.T * post` per channel, whatever the branch."""
    (sw, sh), (dw, dh) = ssize, dsize
    which = mode(ssize, dsize)
    if which == "fast":
        ix, iy = sw // dw, sh // dh
        mx = np.kron(np.eye(dw), np.ones((1, ix)))  # plain block sums ...
        my = np.kron(np.eye(dh), np.ones((1, iy)))
        return mx, my, _f32(1.0 / (ix * iy))        # ... times `float scale = 1.f / area` (resizeAreaFast_)
    if which == "area":
        return _matrix(_area_tab(sw, dw, sw / dw), dw, sw), _matrix(_area_tab(sh, dh, sh / dh), dh, sh), 1.0
    return _linear_area_taps(sw, dw), _linear_area_taps(sh, dh), 1.0


def resize_inter_area(src: np.ndarray, dsize: tuple) -> np.ndarray:
    """cv2.resize(src, dsize, interpolation=cv2.INTER_AREA) for a float64 (H, W) or (H, W, C) array; dsize = (W', H')."""
    src = np.asarray(src, dtype=np.float64)
    sh, sw = src.shape[:2]
    mx, my, post = axis_matrices((sw, sh), (int(dsize[0]), int(dsize[1])))
    if src.ndim == 2:
        return my @ src @ mx.T * post
    return np.stack([my @ src[..., c] @ mx.T * post for c in range(src.shape[2])], axis=2)


def exact_area_matrix(n_in: int, n_out: int) -> np.ndarray:
    """The overlap shares in f64 (what the product uses for every ratio: optrace_amd/convolve.py::_area_weights)."""
    scale = n_in / n_out
    j = np.arange(n_out, dtype=np.float64)[:, None]
    i = np.arange(n_in, dtype=np.float64)[None, :]
    lo, hi = j * scale, (j + 1) * scale
    return np.clip(np.minimum(hi, i + 1) - np.maximum(lo, i), 0.0, None) / scale
