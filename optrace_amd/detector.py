"""Detector stage on the GPU: section search, detector intersection, sphere projection.

Device implementation of Raytracer._hit_detector (raytracer.py:881-1051) through `ot_detector_hits`.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _capi
from ._device import require_device, stream_ptr, ptr, to_dev, f_order_flat, from_f_order


def detector_hits(rays, first: int, count: int, surf_desc: _capi.Surface, projection: int, want_extent: bool,
                  crop=None):
    """-> (ph flat (3*count) f64 device tensor, hw (count) f32 device tensor, extent4 or None, ill_count).
    `crop` = user extent [x0, x1, y0, y1]: hits outside it come back with weight 0 (raytracer.py:1036-1040)."""
    lib = _capi.load_library()
    dev = require_device()
    ph = torch.empty(3 * count, dtype=torch.float64, device=dev)
    hw = torch.empty(count, dtype=torch.float32, device=dev)
    ill = torch.zeros(2, dtype=torch.int64, device=dev)
    ext = None
    if want_extent:
        ext = torch.tensor([np.inf, -np.inf, np.inf, -np.inf], dtype=torch.float64, device=dev)
    rs = rays._rays_struct()
    crop4 = None if crop is None else (C.c_double * 4)(*(float(v) for v in crop))
    _capi.check(lib.ot_detector_hits(C.byref(rs), int(first), int(count), C.byref(surf_desc), int(projection), crop4,
                                     ptr(ph), ptr(hw), ptr(ext), ptr(ill), stream_ptr()))
    # closed-form hits (flat / conic detectors) can neither be ill-conditioned nor time out: no read-back, no sync
    numeric = surf_desc.kind >= _capi.SURF_ASPHERE and surf_desc.z_min != surf_desc.z_max
    ill_h = ill.cpu().numpy() if numeric else (0, 0)
    if ill_h[1]:
        raise TimeoutError("Timeout after 200 iterations in hit finding. Try retracing.")
    return ph, hw, (ext.cpu().numpy() if ext is not None else None), int(ill_h[0])


def project_points(surf_desc: _capi.Surface, p: np.ndarray, projection: int) -> np.ndarray:
    """SphericalSurface.sphere_projection (spherical_surface.py:36-97) via `ot_sphere_projection`."""
    lib = _capi.load_library()
    dev = require_device()
    n = int(np.shape(p)[0])
    dp = to_dev(f_order_flat(p), np.float64)
    out = torch.empty(3 * n, dtype=torch.float64, device=dev)
    _capi.check(lib.ot_sphere_projection(C.byref(surf_desc), int(projection), n, ptr(dp), ptr(out), stream_ptr()))
    return from_f_order(out, n, 3).copy()
