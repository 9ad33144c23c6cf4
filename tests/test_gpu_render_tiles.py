"""RenderImage.render on long hit lists: the tile path (ot_render_tiles.hpp) against the direct kernel and against
NumPy's add.at on the same hits (render_image.py:396-418, misc.py:59-91)."""
import os

import numpy as np
import pytest
import torch

import optrace_amd as ot
from optrace_amd import _capi
from optrace_amd._device import ptr, stream_ptr

pytestmark = pytest.mark.gpu


def render(x, y, w, wl, extent, Nx, Ny, path=None):
    lib = _capi.load_library()
    hist = torch.zeros(Ny * Nx * 4, dtype=torch.float64, device="cuda")
    import ctypes as C
    ext = (C.c_double * 4)(*extent)
    old = os.environ.pop("OT_RENDER_PATH", None)
    if path:
        os.environ["OT_RENDER_PATH"] = path
    try:
        _capi.check(lib.ot_render_accumulate(x.shape[0], ptr(x), ptr(y), ptr(w), ptr(wl), ext, Nx, Ny, ptr(hist),
                                             stream_ptr()))
        torch.cuda.synchronize()
    finally:
        os.environ.pop("OT_RENDER_PATH", None)
        if old is not None:
            os.environ["OT_RENDER_PATH"] = old
    return hist.view(Ny, Nx, 4)


def hits(n, seed, spread=True, Nx=945, Ny=945):
    g = torch.Generator(device="cuda").manual_seed(seed)
    if spread:
        x = torch.rand(n, generator=g, device="cuda", dtype=torch.float64) * 2.4 - 1.2   # some outside [-1, 1]
        y = torch.rand(n, generator=g, device="cuda", dtype=torch.float64) * 2.2 - 1.1
    else:
        x = torch.randn(n, generator=g, device="cuda", dtype=torch.float64) * 0.004
        y = torch.randn(n, generator=g, device="cuda", dtype=torch.float64) * 0.004 + 0.3
    w = torch.rand(n, generator=g, device="cuda", dtype=torch.float32)
    w[::7] = 0.                                                                            # absorbed rays
    wl = torch.rand(n, generator=g, device="cuda", dtype=torch.float32) * 520 + 340        # some outside 360..830
    # edge cases of binning_indices_2d: hits exactly on the upper edges and the corners
    x[:4] = torch.tensor([1., -1., 1., -1.], dtype=torch.float64)
    y[:4] = torch.tensor([1., 1., -1., -1.], dtype=torch.float64)
    w[:4] = 1.
    return x, y, w, wl


@pytest.mark.parametrize("shape", [(945, 945), (4725, 945), (945, 2835)])
def test_tiles_equal_direct_spread(shape):
    Nx, Ny = shape
    x, y, w, wl = hits(3_000_000, 1)
    ext = [-1., 1., -1., 1.]
    a = render(x, y, w, wl, ext, Nx, Ny, "direct")
    b = render(x, y, w, wl, ext, Nx, Ny, "tiles")
    assert float(a[..., 3].sum()) > 0
    scale = float(a.abs().max())
    assert float((a - b).abs().max()) <= 1e-12 * scale
    # pixel occupancy is identical, not only the sums
    assert torch.equal(a[..., 3] != 0, b[..., 3] != 0)


def test_tiles_against_numpy_small_list():
    Nx, Ny = 945, 945
    x, y, w, wl = hits(200_000, 2)
    ext = [-1., 1., -1., 1.]
    b = render(x, y, w, wl, ext, Nx, Ny, "tiles").cpu().numpy()
    xh, yh, wh = x.cpu().numpy(), y.cpu().numpy(), w.cpu().numpy().astype(np.float64)
    inside = (xh >= -1) & (xh <= 1) & (yh >= -1) & (yh <= 1)
    xi = np.floor(Nx / 2 * (xh + 1)).astype(np.int64)
    yi = np.floor(Ny / 2 * (yh + 1)).astype(np.int64)
    xi[xh == 1] = Nx - 1
    yi[yh == 1] = Ny - 1
    ref = np.zeros((Ny, Nx))
    np.add.at(ref, (yi[inside], xi[inside]), wh[inside])
    np.testing.assert_allclose(b[..., 3], ref, rtol=1e-12, atol=1e-300)


def test_default_choice_matches_both_paths():
    """Without a pinned path the probe decides: spread hits and concentrated hits both give the direct result."""
    ext = [-1., 1., -1., 1.]
    for spread in (True, False):
        x, y, w, wl = hits(5_000_000, 3, spread=spread)
        a = render(x, y, w, wl, ext, 945, 945, "direct")
        b = render(x, y, w, wl, ext, 945, 945)
        assert float((a - b).abs().max()) <= 1e-11 * float(a.abs().max())


def test_tiles_accumulate_into_existing_image():
    ext = [-1., 1., -1., 1.]
    x, y, w, wl = hits(2_500_000, 4)
    a = render(x, y, w, wl, ext, 945, 945, "tiles")
    lib = _capi.load_library()
    import ctypes as C
    os.environ["OT_RENDER_PATH"] = "tiles"
    try:
        hist = a.clone().reshape(-1)
        _capi.check(lib.ot_render_accumulate(x.shape[0], ptr(x), ptr(y), ptr(w), ptr(wl), (C.c_double * 4)(*ext), 945, 945,
                                             ptr(hist), stream_ptr()))
        torch.cuda.synchronize()
    finally:
        os.environ.pop("OT_RENDER_PATH", None)
    assert float((hist.view(945, 945, 4) - 2 * a).abs().max()) <= 1e-11 * float(a.abs().max())
