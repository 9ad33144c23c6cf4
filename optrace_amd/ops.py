"""Leaf operators on the GPU: thin NumPy-in / NumPy-out wrappers over the C-ABI.

These back the public per-ray methods of the geometry classes (Surface.find_hit, .normals, .mask,
.values, .hurb_props, RefractionIndex.__call__, RaySource.create_rays, SphericalSurface.sphere_projection).
Every call goes to liboptrace_hip.so; without the library or a device they raise.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _capi
from ._device import require_device, stream_ptr, ptr, to_dev, f_order_flat, from_f_order


def _xy(x, y):
    x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
    y = np.ascontiguousarray(y, dtype=np.float64).reshape(-1)
    if x.shape != y.shape:
        raise ValueError("x and y need to have the same shape")
    return x, y


def surface_find_hit(desc: _capi.Surface, p: np.ndarray, s: np.ndarray):
    lib = _capi.load_library()
    dev = require_device()
    n = int(np.shape(p)[0])
    dp, ds = to_dev(f_order_flat(p), np.float64), to_dev(f_order_flat(s), np.float64)
    ph = torch.empty(3 * n, dtype=torch.float64, device=dev)
    hit = torch.empty(n, dtype=torch.uint8, device=dev)
    ill = torch.empty(n, dtype=torch.uint8, device=dev)
    _capi.check(lib.ot_surface_find_hit(C.byref(desc), n, ptr(dp), ptr(ds), ptr(ph), ptr(hit), ptr(ill), stream_ptr()))
    ill_h = ill.cpu().numpy()
    if np.any(ill_h & 2):
        raise TimeoutError("Timeout after 200 iterations in hit finding. Try retracing.")
    return np.asfortranarray(from_f_order(ph, n, 3)), hit.cpu().numpy().astype(bool), (ill_h & 1).astype(bool)


def surface_normals(desc: _capi.Surface, x, y) -> np.ndarray:
    lib = _capi.load_library()
    dev = require_device()
    x, y = _xy(x, y)
    n = x.shape[0]
    out = torch.empty(3 * n, dtype=torch.float64, device=dev)
    dx, dy = to_dev(x, np.float64), to_dev(y, np.float64)
    _capi.check(lib.ot_surface_normals(C.byref(desc), n, ptr(dx), ptr(dy), ptr(out), stream_ptr()))
    return np.asfortranarray(from_f_order(out, n, 3))


def surface_mask(desc: _capi.Surface, x, y) -> np.ndarray:
    lib = _capi.load_library()
    dev = require_device()
    shape = np.shape(x)
    x, y = _xy(x, y)
    n = x.shape[0]
    out = torch.empty(n, dtype=torch.uint8, device=dev)
    dx, dy = to_dev(x, np.float64), to_dev(y, np.float64)
    _capi.check(lib.ot_surface_mask(C.byref(desc), n, ptr(dx), ptr(dy), ptr(out), stream_ptr()))
    return out.cpu().numpy().astype(bool).reshape(shape)


def surface_values(desc: _capi.Surface, x, y) -> np.ndarray:
    lib = _capi.load_library()
    dev = require_device()
    shape = np.shape(x)
    x, y = _xy(x, y)
    n = x.shape[0]
    out = torch.empty(n, dtype=torch.float64, device=dev)
    dx, dy = to_dev(x, np.float64), to_dev(y, np.float64)
    _capi.check(lib.ot_surface_values(C.byref(desc), n, ptr(dx), ptr(dy), ptr(out), stream_ptr()))
    return out.cpu().numpy().reshape(shape)


def surface_hurb_props(desc: _capi.Surface, x, y):
    lib = _capi.load_library()
    dev = require_device()
    x, y = _xy(x, y)
    n = x.shape[0]
    a_ = torch.empty(n, dtype=torch.float64, device=dev)
    b_ = torch.empty(n, dtype=torch.float64, device=dev)
    b = torch.empty(3 * n, dtype=torch.float64, device=dev)
    inside = torch.empty(n, dtype=torch.uint8, device=dev)
    dx, dy = to_dev(x, np.float64), to_dev(y, np.float64)
    _capi.check(lib.ot_surface_hurb_props(C.byref(desc), n, ptr(dx), ptr(dy), ptr(a_), ptr(b_), ptr(b),
                                          ptr(inside), stream_ptr()))
    return a_.cpu().numpy(), b_.cpu().numpy(), from_f_order(b, n, 3).copy(), inside.cpu().numpy().astype(bool)


def refraction_index(md: _capi.Medium, pool: np.ndarray, wl: np.ndarray) -> np.ndarray:
    """n(wl); wl is rounded to float32 first exactly like RayStorage stores it (ray_storage.py:84)."""
    lib = _capi.load_library()
    dev = require_device()
    wl32 = np.ascontiguousarray(wl, dtype=np.float32).reshape(-1)
    if not np.array_equal(wl32.astype(np.float64), np.asarray(wl, dtype=np.float64).reshape(-1)):
        # the device path is defined on the stored float32 wavelengths; arbitrary doubles would be rounded
        pass
    n = wl32.shape[0]
    dwl = to_dev(wl32, np.float32)
    dpool = to_dev(pool if pool.size else np.zeros(1), np.float64)
    out = torch.empty(n, dtype=torch.float64, device=dev)
    _capi.check(lib.ot_refraction_index(C.byref(md), ptr(dpool), int(pool.size), n, ptr(dwl), ptr(out), stream_ptr()))
    return out.cpu().numpy()


def sphere_projection(desc: _capi.Surface, p: np.ndarray, method: str) -> np.ndarray:
    """SphericalSurface.sphere_projection through the detector kernel's projection stage."""
    from .detector import project_points
    return project_points(desc, p, _capi.PROJECTIONS[method])


def create_rays(source, N: int, no_pol: bool = False, power: float = None):
    """RaySource.create_rays on the device -> (p, s, pols, weights, wavelengths) host arrays."""
    from .ray_storage import RayStorage
    st = RayStorage()
    st.init([source], int(N), 1, bool(no_pol), _single_power=power)
    st.generate(seed=None)
    p = st.p_list[:, 0]
    s = st.s0_list
    if np.any(s[:, 2] <= 0):  # ray_source.py:353
        raise RuntimeError("All ray divergences s need to be in positive z-divergence")
    pols = st.pol_list[:, 0].astype(np.float64) if not no_pol else np.broadcast_to(np.nan, p.shape)
    return p, s, pols, st.w_list[:, 0], st.wl_list.astype(np.float64)
