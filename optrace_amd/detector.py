"""Detector stage on the GPU: section search, detector intersection, sphere projection.

Device implementation of Raytracer._hit_detector (raytracer.py:881-1051) through `ot_detector_hits`.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _capi
from ._device import require_device, stream_ptr, ptr, to_dev, f_order_flat, from_f_order, mailbox, sync_stream, alloc_retry


def detector_hits_multi(rays, first: int, count: int, requests: list):
    """Hit search for several detectors in one pass over the ray sections (`ot_detector_hits_multi`).

    requests: dicts with surf_desc (_capi.Surface), projection (int), want_extent (bool), crop ([x0, x1, y0, y1] or
    None: hits outside come back with weight 0, raytracer.py:1036-1040); compact (bool): only the valid hits are written,
    gathered at the front of the list's 1024 pieces (`ot_detector_req.fill`) -- the tuple then carries (wl, fill) in
    fifth place, the hit wavelengths and the fill counts as device tensors.
    -> list of (ph flat f64 device tensor: x plane, y plane and, with want_z, z plane of count entries each, hw (count)
    f32 device tensor, extent4 or None, ill_count).  Binning and spectra use x and y only."""
    lib = _capi.load_library()
    dev = require_device()
    n = len(requests)
    reqs = (_capi.DetectorReq * n)()
    keep, outs = [], []
    # extents go straight to a pinned host buffer (plain stores of one small kernel), 4 doubles per request behind 2n unused
    # words; the ill-conditioned counts are device atomics and stay in device memory (read back for numeric detectors only)
    mb_t, mb = mailbox()
    mbf = mb.view(np.float64)
    ill = torch.zeros(2 * n, dtype=torch.int64, device=dev)
    any_numeric = False
    for k, rq in enumerate(requests):
        sd = rq["surf_desc"]
        want_z = bool(rq.get("want_z", False))
        compact = bool(rq.get("compact", False)) and not want_z
        cap = _capi.HIT_PIECES * int(lib.ot_hit_piece_len(int(count))) if compact else count  # entries per plane
        no_pos = compact and bool(rq.get("weights_only", False))  # (detector spectrum: weights and wavelengths alone)
        # (hit lists are the large allocations of this stage: out of memory -> the library's kept scratch goes back first)
        ph, hw, wl_c, fill = alloc_retry(lambda: (
            None if no_pos else torch.empty((3 if want_z else 2) * cap, dtype=torch.float64, device=dev),
            torch.empty(cap, dtype=torch.float32, device=dev),
            torch.empty(cap, dtype=torch.float32, device=dev) if compact else None,
            torch.zeros(_capi.HIT_PIECES, dtype=torch.int32, device=dev) if compact else None))
        ext = None
        if rq["want_extent"]:
            ext = 2 * n + 4 * k  # word offset in the mailbox
            mbf[ext:ext + 4] = [np.inf, -np.inf, np.inf, -np.inf]
        crop4 = None if rq.get("crop") is None else (C.c_double * 4)(*(float(v) for v in rq["crop"]))
        keep.append((sd, crop4))
        r = reqs[k]
        r.detector = C.addressof(sd)
        r.projection = int(rq["projection"])
        r.xy_only = 0 if want_z else 1
        r.crop4 = None if crop4 is None else C.addressof(crop4)
        r.ph, r.hw = (None if ph is None else ph.data_ptr()), hw.data_ptr()
        r.extent4 = mb_t.data_ptr() + 8 * ext if ext is not None else None
        r.wl_out, r.fill = (wl_c.data_ptr(), fill.data_ptr()) if compact else (None, None)
        r.ill_count = ill.data_ptr() + 16 * k
        # closed-form hits (flat / conic detectors) can neither be ill-conditioned nor time out
        numeric = sd.kind >= _capi.SURF_ASPHERE and sd.z_min != sd.z_max
        any_numeric = any_numeric or numeric
        outs.append([ph, hw, ext, numeric, (wl_c, fill) if compact else None])
    rs = rays._rays_struct()
    _capi.check(lib.ot_detector_hits_multi(C.byref(rs), int(first), int(count), reqs, n, stream_ptr()))
    ill_h = ill.cpu().numpy() if any_numeric else np.zeros(2 * n, dtype=np.int64)  # no read-back otherwise
    if any(o[2] is not None for o in outs):
        sync_stream()  # the mailbox is complete; no wait at all for closed-form detectors with user extents
    res = []
    for k, (ph, hw, ext, numeric, comp) in enumerate(outs):
        if ill_h[2 * k + 1]:
            raise TimeoutError("Timeout after 200 iterations in hit finding. Try retracing.")
        one = (ph, hw, (mbf[ext:ext + 4].copy() if ext is not None else None), int(ill_h[2 * k]))
        res.append(one + (comp,) if comp is not None else one)
    return res


def detector_extents(rays, first: int, count: int, requests: list) -> list:
    """Extent of the valid hits of up to 8 detectors in one pass over the ray sections, without hit lists
    (`ot_detector_hits_multi` with ph = hw = NULL): 52 B read per ray, nothing written.  The first half of an image with
    an automatic extent (raytracer.py:1042-1046); the second is `detector_images` with that extent.
    requests: dicts with surf_desc, projection.  -> list of (extent4 numpy [x0, x1, y0, y1], +-inf without a hit;
    ill_count)."""
    lib = _capi.load_library()
    dev = require_device()
    n = len(requests)
    reqs = (_capi.DetectorReq * n)()
    keep = []
    mb_t, mb = mailbox()
    mbf = mb.view(np.float64)
    mbf[2 * n:6 * n] = [np.inf, -np.inf, np.inf, -np.inf] * n
    ill = torch.zeros(2 * n, dtype=torch.int64, device=dev)
    any_numeric = False
    for k, rq in enumerate(requests):
        sd = rq["surf_desc"]
        keep.append(sd)
        r = reqs[k]
        r.detector = C.addressof(sd)
        r.projection = int(rq["projection"])
        r.xy_only = 1
        r.crop4 = None
        r.ph, r.hw = None, None
        r.wl_out, r.fill = None, None
        r.extent4 = mb_t.data_ptr() + 8 * (2 * n + 4 * k)
        r.ill_count = ill.data_ptr() + 16 * k
        any_numeric = any_numeric or (sd.kind >= _capi.SURF_ASPHERE and sd.z_min != sd.z_max)
    rs = rays._rays_struct()
    _capi.check(lib.ot_detector_hits_multi(C.byref(rs), int(first), int(count), reqs, n, stream_ptr()))
    sync_stream()
    ext_h = mbf[2 * n:6 * n].copy().reshape(n, 4)
    ill_h = ill.cpu().numpy() if any_numeric else np.zeros(2 * n, dtype=np.int64)
    if ill_h[1::2].any():
        raise TimeoutError("Timeout after 200 iterations in hit finding. Try retracing.")
    return [(ext_h[k].copy(), int(ill_h[2 * k])) for k in range(n)]


def auto_image_supported(surf_desc: _capi.Surface, projection: int) -> bool:
    """Detectors `AutoImage` serves: closed-form hit (flat, conic / spherical), no sphere projection with transcendentals."""
    closed = surf_desc.kind <= _capi.SURF_CONIC or surf_desc.z_min == surf_desc.z_max
    return closed and projection in (_capi.PROJECTIONS[None], _capi.PROJECTIONS["Orthographic"])


def detector_extent_sample(rays, first: int, count: int, surf_desc: _capi.Surface, projection: int,
                           stride: int) -> np.ndarray:
    """Extent [x0, x1, y0, y1] (+-inf without a hit) of the hits of every `stride`-th wave of 64 rays
    (`ot_detector_extent_sample`): a box inside the automatic extent of raytracer.py:1042-1046, for ~1 / stride of the
    bytes."""
    lib = _capi.load_library()
    require_device()
    mb_t, mb = mailbox()
    rs = rays._rays_struct()
    _capi.check(lib.ot_detector_extent_sample(C.byref(rs), int(first), int(count), C.byref(surf_desc), int(projection),
                                              int(stride), C.c_void_p(mb_t.data_ptr()), stream_ptr()))
    return mb.view(np.float64)[:4].copy()  # (the call waits for the stream)


class AutoImage:
    """Detector image with an automatic extent in one pass over the ray sections (`ot_detector_image_auto_*`): the hits
    are sorted into the tiles of a provisional grid, the exact extent comes back, `finish` bins into the final grid.

    grid: (X0, Y0, tile_w, tile_h, tiles_x, tiles_y).  After construction: extent (numpy, +-inf without a hit),
    escaped (hits outside the grid) and escape_capacity; `finish` or `cancel` must follow."""

    def __init__(self, rays, first: int, count: int, surf_desc: _capi.Surface, projection: int, grid: tuple) -> None:
        self._lib = _capi.load_library()
        require_device()
        mb_t, mb = mailbox()
        rs = rays._rays_struct()
        origin = (C.c_double * 2)(float(grid[0]), float(grid[1]))
        tile = (C.c_double * 2)(float(grid[2]), float(grid[3]))
        tiles = (C.c_int32 * 2)(int(grid[4]), int(grid[5]))
        self._handle = C.c_void_p()
        _capi.check(self._lib.ot_detector_image_auto_begin(C.byref(rs), int(first), int(count), C.byref(surf_desc),
                                                           int(projection), origin, tile, tiles,
                                                           C.c_void_p(mb_t.data_ptr()), C.byref(self._handle),
                                                           stream_ptr()))
        res = mb.view(np.float64)[:6].copy()  # (the call waits for the stream)
        self.extent = res[:4]
        self.escaped = int(res[4])
        self.escape_capacity = int(res[5])

    def finish(self, extent, Nx: int, Ny: int, hist: torch.Tensor) -> None:
        """Add the image to hist (flat f64 device tensor of Ny * Nx * 4 entries); extent = the fixed image extent."""
        h, self._handle = self._handle, C.c_void_p()
        ext = (C.c_double * 4)(*[float(v) for v in extent])
        _capi.check(self._lib.ot_detector_image_auto_finish(h, ext, int(Nx), int(Ny), ptr(hist), stream_ptr()))

    def cancel(self) -> None:
        if self._handle:
            self._lib.ot_detector_image_auto_cancel(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.cancel()
        except Exception:
            pass


def detector_images(rays, first: int, count: int, requests: list) -> list:
    """Hit search and binning fused (`ot_detector_images`) for detector images whose extent is known beforehand.

    requests: dicts with surf_desc (_capi.Surface), projection (int), crop ([x0, x1, y0, y1]: the user extent hits are
    restricted to, or None), extent (image extent after RenderImage._fix_extent), Nx, Ny, hist (flat f64 device tensor
    of Ny * Nx * 4 entries that the hits are ADDED to), weight_scale (optional: every weight times this before it is added).
    At most 8 per call.  -> ill-conditioned count per request."""
    lib = _capi.load_library()
    dev = require_device()
    n = len(requests)
    reqs = (_capi.DetectorImageReq * n)()
    keep = []
    ill = torch.zeros(2 * n, dtype=torch.int64, device=dev)
    any_numeric = False
    for k, rq in enumerate(requests):
        sd = rq["surf_desc"]
        crop4 = None if rq.get("crop") is None else (C.c_double * 4)(*(float(v) for v in rq["crop"]))
        keep.append((sd, crop4))
        r = reqs[k]
        r.detector = C.addressof(sd)
        r.projection = int(rq["projection"])
        r.Nx, r.Ny = int(rq["Nx"]), int(rq["Ny"])
        r.crop4 = None if crop4 is None else C.addressof(crop4)
        r.extent[:] = [float(v) for v in rq["extent"]]
        r.hist = rq["hist"].data_ptr()
        r.weight_scale = float(rq.get("weight_scale", 1.0))
        r.ill_count = ill.data_ptr() + 16 * k
        any_numeric = any_numeric or (sd.kind >= _capi.SURF_ASPHERE and sd.z_min != sd.z_max)
    rs = rays._rays_struct()
    _capi.check(lib.ot_detector_images(C.byref(rs), int(first), int(count), reqs, n, stream_ptr()))
    ill_h = ill.cpu().numpy() if any_numeric else np.zeros(2 * n, dtype=np.int64)  # no read-back, no sync otherwise
    if ill_h[1::2].any():
        raise TimeoutError("Timeout after 200 iterations in hit finding. Try retracing.")
    return [int(v) for v in ill_h[0::2]]


def detector_hits(rays, first: int, count: int, surf_desc: _capi.Surface, projection: int, want_extent: bool,
                  crop=None):
    """-> (ph flat (3*count) f64 device tensor, hw (count) f32 device tensor, extent4 or None, ill_count).
    `crop` = user extent [x0, x1, y0, y1]: hits outside it come back with weight 0 (raytracer.py:1036-1040)."""
    return detector_hits_multi(rays, first, count, [dict(surf_desc=surf_desc, projection=projection, want_z=True,
                                                         want_extent=want_extent, crop=crop)])[0]


def project_points(surf_desc: _capi.Surface, p: np.ndarray, projection: int) -> np.ndarray:
    """SphericalSurface.sphere_projection (spherical_surface.py:36-97) via `ot_sphere_projection`."""
    lib = _capi.load_library()
    dev = require_device()
    n = int(np.shape(p)[0])
    dp = to_dev(f_order_flat(p), np.float64)
    out = torch.empty(3 * n, dtype=torch.float64, device=dev)
    _capi.check(lib.ot_sphere_projection(C.byref(surf_desc), int(projection), n, ptr(dp), ptr(out), stream_ptr()))
    return from_f_order(out, n, 3).copy()
