"""ctypes front end of the CPU oracle (oracle/oracle.c) for the tests.

Test infrastructure: builds oracle/_build/liboracle.so with gcc on first use and wraps its entry points
in NumPy-in / NumPy-out functions.  The descriptors are the ctypes structures of optrace_amd._capi
(plain data shared with the C-ABI header); no optrace_amd code runs inside the oracle.
"""
from __future__ import annotations

import ctypes as C
import pathlib
import subprocess

import numpy as np

from optrace_amd import _capi

ROOT = pathlib.Path(__file__).resolve().parent.parent
LIB = ROOT / "oracle" / "_build" / "liboracle.so"

_lib = None
dp = C.POINTER(C.c_double)
fp = C.POINTER(C.c_float)
u8p = C.POINTER(C.c_uint8)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        src = ROOT / "oracle" / "oracle.c"
        if not LIB.exists() or LIB.stat().st_mtime < src.stat().st_mtime:
            subprocess.run(["make", "-C", str(ROOT / "oracle")], check=True, capture_output=True)
        _lib = C.CDLL(str(LIB))
        _lib.orc_trace.restype = C.c_int
        _lib.orc_surface_find_hit.restype = C.c_int
        _lib.orc_detector_hits.restype = C.c_int
    return _lib


def _d(a):
    return a.ctypes.data_as(dp)


def _f(a):
    return a.ctypes.data_as(fp)


def fflat(a, dtype=np.float64):
    """(n, k) -> flat component-major copy."""
    return np.ascontiguousarray(np.asarray(a, dtype=dtype).T).reshape(-1)


def unflat(a, n, k):
    return a.reshape(k, n).T


class HostRays:
    """Host buffers in the RayStorage layout + the ot_rays struct pointing at them."""

    def __init__(self, N: int, nt: int, no_pol: bool):
        self.N, self.nt = N, nt
        self.p = np.zeros(3 * nt * N)
        self.s = np.zeros(3 * N)
        self.w = np.zeros(nt * N, dtype=np.float32)
        self.n = np.zeros(nt * N)
        self.wl = np.zeros(N, dtype=np.float32)
        self.pol = None if no_pol else np.zeros(3 * nt * N, dtype=np.float32)
        r = _capi.Rays()
        r.N, r.nt = N, nt
        r.p, r.s, r.w, r.n, r.wl = (self.p.ctypes.data, self.s.ctypes.data, self.w.ctypes.data,
                                    self.n.ctypes.data, self.wl.ctypes.data)
        r.pol = self.pol.ctypes.data if self.pol is not None else None
        self.struct = r

    def set_initial(self, p0, s0, pol0, w0, wl):
        N, nt = self.N, self.nt
        for c in range(3):
            self.p[c * nt * N: c * nt * N + N] = p0[:, c]
            self.s[c * N:(c + 1) * N] = s0[:, c]
            if self.pol is not None:
                self.pol[c * nt * N: c * nt * N + N] = pol0[:, c].astype(np.float32)
        self.w[:N] = w0
        self.wl[:] = wl

    @classmethod
    def from_lists(cls, p_list, w_list, wl_list, no_pol=True):
        """Build storage from full (N, nt, 3) arrays (for the detector stage)."""
        N, nt = p_list.shape[:2]
        h = cls(N, nt, no_pol)
        h.p[:] = np.ascontiguousarray(p_list.transpose(2, 1, 0)).reshape(-1)
        h.w[:] = np.ascontiguousarray(w_list.T).reshape(-1)
        h.wl[:] = wl_list
        return h

    @property
    def p_list(self):
        return self.p.reshape(3, self.nt, self.N).transpose(2, 1, 0)

    @property
    def pol_list(self):
        return self.pol.reshape(3, self.nt, self.N).transpose(2, 1, 0)

    @property
    def w_list(self):
        return self.w.reshape(self.nt, self.N).T

    @property
    def n_list(self):
        return self.n.reshape(self.nt, self.N).T

    @property
    def s_final(self):
        return self.s.reshape(3, self.N).T


def trace(desc: _capi.SceneDesc, rays: HostRays, hurb_normals=None, threads: int = 1) -> tuple[np.ndarray, int]:
    """orc_trace on one thread, or orc_trace_mt on `threads` host threads (same result: rays are independent)."""
    msgs = np.zeros(5 * rays.nt, dtype=np.int64)
    hn = None
    if hurb_normals is not None:
        hn = np.ascontiguousarray(hurb_normals, dtype=np.float64).reshape(-1)
    args = (C.byref(desc), C.byref(rays.struct), _d(hn) if hn is not None else None, msgs.ctypes.data_as(C.POINTER(C.c_int64)))
    st = lib().orc_trace(*args) if threads <= 1 else lib().orc_trace_mt(*args, C.c_int(int(threads)))
    return msgs.reshape(5, rays.nt), st


def find_hit(sd: _capi.Surface, p, s):
    n = p.shape[0]
    pf, sf = fflat(p), fflat(s)
    ph = np.zeros(3 * n)
    hit = np.zeros(n, dtype=np.uint8)
    ill = np.zeros(n, dtype=np.uint8)
    st = lib().orc_surface_find_hit(C.byref(sd), C.c_int64(n), _d(pf), _d(sf), _d(ph), hit.ctypes.data_as(u8p),
                                    ill.ctypes.data_as(u8p))
    return unflat(ph, n, 3), hit.astype(bool), ill.astype(bool), st


def normals(sd, x, y):
    n = x.shape[0]
    x, y = np.ascontiguousarray(x, dtype=np.float64), np.ascontiguousarray(y, dtype=np.float64)
    out = np.zeros(3 * n)
    lib().orc_surface_normals(C.byref(sd), C.c_int64(n), _d(x), _d(y), _d(out))
    return unflat(out, n, 3)


def mask(sd, x, y):
    n = x.shape[0]
    x, y = np.ascontiguousarray(x, dtype=np.float64), np.ascontiguousarray(y, dtype=np.float64)
    out = np.zeros(n, dtype=np.uint8)
    lib().orc_surface_mask(C.byref(sd), C.c_int64(n), _d(x), _d(y), out.ctypes.data_as(u8p))
    return out.astype(bool)


def values(sd, x, y):
    n = x.shape[0]
    x, y = np.ascontiguousarray(x, dtype=np.float64), np.ascontiguousarray(y, dtype=np.float64)
    out = np.zeros(n)
    lib().orc_surface_values(C.byref(sd), C.c_int64(n), _d(x), _d(y), _d(out))
    return out


def hurb_props(sd, x, y):
    n = x.shape[0]
    x, y = np.ascontiguousarray(x, dtype=np.float64), np.ascontiguousarray(y, dtype=np.float64)
    a_, b_, b = np.zeros(n), np.zeros(n), np.zeros(3 * n)
    inside = np.zeros(n, dtype=np.uint8)
    lib().orc_surface_hurb_props(C.byref(sd), C.c_int64(n), _d(x), _d(y), _d(a_), _d(b_), _d(b),
                                 inside.ctypes.data_as(u8p))
    return a_, b_, unflat(b, n, 3), inside.astype(bool)


def refraction_index(md: _capi.Medium, pool: np.ndarray, wl: np.ndarray):
    wl = np.ascontiguousarray(wl, dtype=np.float32)
    pool = np.ascontiguousarray(pool if len(pool) else np.zeros(1), dtype=np.float64)
    out = np.zeros(wl.shape[0])
    lib().orc_refraction_index(C.byref(md), _d(pool), C.c_int64(wl.shape[0]), _f(wl), _d(out))
    return out


def filter_T(fd: _capi.Filter, pool: np.ndarray, wl: np.ndarray):
    wl = np.ascontiguousarray(wl, dtype=np.float32)
    pool = np.ascontiguousarray(pool if len(pool) else np.zeros(1), dtype=np.float64)
    out = np.zeros(wl.shape[0])
    lib().orc_filter(C.byref(fd), _d(pool), C.c_int64(wl.shape[0]), _f(wl), _d(out))
    return out


def observer_table() -> np.ndarray:
    return np.ascontiguousarray(np.load(ROOT / "optrace_amd" / "data" / "cie_tables.npz")["observers"])


def observers(wl):
    tab = observer_table()
    wl = np.ascontiguousarray(wl, dtype=np.float32)
    out = np.zeros(3 * wl.shape[0])
    lib().orc_observers(_d(tab), C.c_int64(tab.shape[0]), C.c_int64(wl.shape[0]), _f(wl), _d(out))
    return unflat(out, wl.shape[0], 3)


def binning(x, y, w, Nx, Ny, extent):
    n = x.shape[0]
    x, y = np.ascontiguousarray(x, dtype=np.float64), np.ascontiguousarray(y, dtype=np.float64)
    w = np.ascontiguousarray(w, dtype=np.float32)
    ext = np.ascontiguousarray(extent, dtype=np.float64)
    xi, yi = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
    wm = np.zeros(n, dtype=np.float32)
    lib().orc_binning_indices_2d(C.c_int64(n), _d(x), _d(y), _f(w), C.c_int(Nx), C.c_int(Ny), _d(ext),
                                 xi.ctypes.data_as(C.POINTER(C.c_int32)), yi.ctypes.data_as(C.POINTER(C.c_int32)),
                                 _f(wm))
    return xi, yi, wm


def detector_hits(rays: HostRays, first, count, sd: _capi.Surface, projection: int):
    ph = np.zeros(3 * count)
    hw = np.zeros(count, dtype=np.float32)
    ext = np.array([np.inf, -np.inf, np.inf, -np.inf])
    ill = np.zeros(1, dtype=np.int64)
    st = lib().orc_detector_hits(C.byref(rays.struct), C.c_int64(first), C.c_int64(count), C.byref(sd),
                                 C.c_int32(projection), _d(ph), _f(hw), _d(ext),
                                 ill.ctypes.data_as(C.POINTER(C.c_int64)))
    return unflat(ph, count, 3), hw, ext, int(ill[0]), st


def render(px, py, w, wl, extent, Nx, Ny):
    tab = observer_table()
    n = px.shape[0]
    px, py = np.ascontiguousarray(px, dtype=np.float64), np.ascontiguousarray(py, dtype=np.float64)
    w, wl = np.ascontiguousarray(w, dtype=np.float32), np.ascontiguousarray(wl, dtype=np.float32)
    ext = np.ascontiguousarray(extent, dtype=np.float64)
    hist = np.zeros((Ny, Nx, 4))
    lib().orc_render_accumulate(_d(tab), C.c_int64(tab.shape[0]), C.c_int64(n), _d(px), _d(py), _f(w), _f(wl),
                                _d(ext), C.c_int(Nx), C.c_int(Ny), _d(hist))
    return hist
