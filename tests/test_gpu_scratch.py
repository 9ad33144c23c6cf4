"""The library's scratch pool on the device (csrc/ot_scratch.hpp through ot_api.hip; its bookkeeping alone is tested on the CPU
by tests/test_scratch_pool.py): images do not depend on what the pool holds, `ot_scratch_trim` frees the idle blocks only -- an
automatic image that is open keeps its records --, two automatic images may be open side by side on one stream, the cap
evicts, and a torch allocation that runs out of memory gets the idle blocks back (`_device.alloc_retry`)."""
import ctypes as C

import numpy as np
import pytest
import torch

import optrace_amd as ot
from optrace_amd import _capi, detector as _detector, _device
from test_gpu_fused_detector import same_image, image_scene, pinned

pytestmark = pytest.mark.gpu


def stats(lib):
    kept, blocks, leased = C.c_int64(), C.c_int32(), C.c_int32()
    _capi.check(lib.ot_scratch_stats(C.byref(kept), C.byref(blocks), C.byref(leased)))
    return kept.value, blocks.value, leased.value


def auto_image(RT, N):
    """(AutoImage, RenderImage to finish into, histogram) for detector 0 of RT."""
    rq = RT._detector_requests([dict(detector_index=0, source_index=None, extent=None, projection_method="Equidistant")])[0]
    sd, proj = rq["surf_desc"], _capi.PROJECTIONS[rq["projection"]]
    e0 = _detector.detector_extent_sample(RT.rays, 0, N, sd, proj, 16)
    grid, tw, th = RT._auto_grid(e0, None, rq["projection"], RT.AUTO_MARGINS)
    auto = _detector.AutoImage(RT.rays, 0, N, sd, proj, grid)
    img = ot.RenderImage(extent=auto.extent.copy(), projection=None)
    img._limit = None
    img._fix_extent()
    Nx, Ny = img._pixel_counts()
    return auto, img, torch.zeros(Ny * Nx * 4, dtype=torch.float64, device="cuda"), (Nx, Ny)


def test_trim_leaves_an_open_automatic_image_alone_and_two_may_be_open():
    lib = _capi.load_library()
    with ot.global_options.no_warnings():
        N = 600_000
        RT = image_scene(N=N)
        ref = RT.detector_image(_unfused=True)  # hit-list chain
        _capi.check(lib.ot_scratch_trim())
        assert stats(lib)[2] == 0
        a, img_a, hist_a, (Nx, Ny) = auto_image(RT, N)
        b, img_b, hist_b, _ = auto_image(RT, N)  # a second one on the same stream: a block of its own
        kept, blocks, leased = stats(lib)
        assert leased == 2 and blocks >= 2
        _capi.check(lib.ot_scratch_trim())  # frees what is idle; the two open images keep their records
        kept2, blocks2, leased2 = stats(lib)
        assert leased2 == 2 and blocks2 == 2 and 0 < kept2 <= kept
        RT.detector_image(extent=[-8., 8., -8., 8.])  # other work on the stream in between
        b.finish(img_b.extent, Nx, Ny, hist_b)
        a.finish(img_a.extent, Nx, Ny, hist_a)
        torch.cuda.synchronize()
        assert stats(lib)[2] == 0
    for hist, img in ((hist_a, img_a), (hist_b, img_b)):
        got = hist.view(Ny, Nx, 4).cpu().numpy()
        np.testing.assert_array_equal(img.extent, ref.extent)
        assert np.array_equal(got[..., 3] != 0, ref._data[..., 3] != 0)
        assert np.abs(got - ref._data).max() <= 1e-11 * np.abs(ref._data).max()
    _capi.check(lib.ot_scratch_trim())
    assert stats(lib) == (0, 0, 0)


def test_images_do_not_depend_on_the_pool_trimmed_or_under_a_tiny_cap():
    lib = _capi.load_library()
    with pinned("tiles"), ot.global_options.no_warnings():
        RT = image_scene(N=1_500_000)
        one = RT.detector_image(extent=[-8., 8., -8., 8.])
        auto1 = RT.detector_image()
        assert stats(lib)[0] > 0
        _capi.check(lib.ot_scratch_trim())
        assert stats(lib) == (0, 0, 0)
        two = RT.detector_image(extent=[-8., 8., -8., 8.])
        _capi.check(lib.ot_scratch_set_cap(1))  # every block beyond the one on lease goes at the next allocation
        try:
            three = RT.detector_image(extent=[-8., 8., -8., 8.])
            auto2 = RT.detector_image()
            torch.cuda.synchronize()
            assert stats(lib)[1] <= 2
        finally:
            _capi.check(lib.ot_scratch_set_cap(64_000_000_000))
    # the same kernels on the same records: equal up to the order in which the LDS atomics of one pixel arrive, which no two
    # runs share (bit-equality is not on offer with floating-point atomics, trim or no trim)
    same_image(one, two, tol=1e-13)
    same_image(one, three, tol=1e-13)
    same_image(auto1, auto2, tol=1e-13)


def test_alloc_retry_trims_the_pool_when_torch_runs_out_of_memory():
    lib = _capi.load_library()
    with ot.global_options.no_warnings():
        RT = image_scene(N=1_500_000)
        with pinned("tiles"):
            RT.detector_image(extent=[-8., 8., -8., 8.])
    assert stats(lib)[0] > 0
    calls = []

    def make():
        calls.append(stats(lib)[0])
        if len(calls) == 1:
            raise torch.OutOfMemoryError("simulated")
        return torch.empty(16, device="cuda")

    t = _device.alloc_retry(make)
    assert t.numel() == 16 and len(calls) == 2
    assert calls[0] > 0 and calls[1] == 0, "the idle scratch went back to the driver before the second try"
