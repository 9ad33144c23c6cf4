"""Tilted, data-defined and function-defined surfaces.

Mirror of optrace/tracer/geometry/surface/{tilted_surface,data_surface_1d,data_surface_2d,function_surface_1d,
function_surface_2d}.py.  Per-ray work (hit search, values, normals, masks) runs in the HIP kernels; this
module holds the property handling and the one-off set-up: for data surfaces the same SciPy spline fit the
reference performs in its constructor, handed to the device as knots + B-spline coefficients (value and first
derivatives) which the kernels evaluate in FITPACK's operation order (csrc/ot_spline.hpp).

Function surfaces are defined by Python callables, which cannot run inside a kernel: they are sampled once at
construction on a dense grid and carried as the same quartic splines ("tabulated FunctionSurface").
The sampling density is a class attribute; the measured residual against the callable is checked at
construction and reported if it exceeds `TAB_TOL`.
"""
from __future__ import annotations

import copy
from typing import Any, Callable

import numpy as np
import scipy.interpolate

from .. import _capi
from ..base import check_type, check_above
from .._warn import warning
from .surfaces import Surface

_K = _capi.SPL_K


# ---- spline tables (layout: include/optrace_amd.h, ot_surface.tab) -------------------------------------------
def _deriv_coeffs_1d(t: np.ndarray, c: np.ndarray, k: int) -> np.ndarray:
    """First-derivative coefficients as FITPACK's splder.f builds them: wrk(i) = k (c(i+1) - c(i)) / (t(i+k+1) - t(i+1))."""
    n = t.shape[0]
    nk1 = n - k - 1
    wrk = c[:nk1].copy()
    fac = t[1 + k:k + nk1] - t[1:nk1]
    ok = fac > 0
    wrk[:nk1 - 1][ok] = (k * (c[1:nk1] - c[:nk1 - 1]) / fac)[ok]
    out = np.zeros(n)
    out[:nk1] = wrk
    return out


def _table_1d(spl) -> tuple[np.ndarray, int]:
    t, c, k = spl._eval_args
    assert k == _K
    t, c = np.asarray(t, dtype=np.float64), np.asarray(c, dtype=np.float64)
    n = t.shape[0]
    cc = np.zeros(n)
    cc[:c.shape[0]] = c[:n]
    return np.ascontiguousarray(np.concatenate((t, cc, _deriv_coeffs_1d(t, cc, k)))), n


def _table_2d(spl) -> tuple[np.ndarray, int]:
    tx, ty, c = spl.tck
    kx, ky = spl.degrees
    assert kx == _K and ky == _K and np.array_equal(tx, ty)
    t = np.asarray(tx, dtype=np.float64)
    n = t.shape[0]
    nc = n - _K - 1
    C = np.asarray(c, dtype=np.float64).reshape(nc, nc)
    # parder.f / pardeu.f: (c(i+1, :) - c(i, :)) * k / (t(i+k+1) - t(i+1)), same along y
    fac = t[1 + _K:_K + nc] - t[1:nc]
    ok = fac > 0
    cx = C[:-1, :].copy()
    cx[ok, :] = ((C[1:, :] - C[:-1, :]) * _K / fac[:, None])[ok, :]
    cy = C[:, :-1].copy()
    cy[:, ok] = ((C[:, 1:] - C[:, :-1]) * _K / fac[None, :])[:, ok]
    return np.ascontiguousarray(np.concatenate((t, C.ravel(), cx.ravel(), cy.ravel()))), n


# ---- TiltedSurface ---------------------------------------------------------------------------------------------
class TiltedSurface(Surface):
    """Circular plane section with an arbitrary normal (tilted_surface.py:10-161)."""

    rotational_symmetry = False
    _kind = _capi.SURF_TILTED

    def __init__(self, r: float, normal=None, normal_sph=None, **kwargs) -> None:
        Surface.__init__(self, r, **kwargs)
        if normal is None:
            if normal_sph is None:
                raise RuntimeError("normal or normal_sph parameter needs to be specified.")
            check_type("normal_sph", normal_sph, (list, np.ndarray))
            polar, azimuth = (np.radians(angle) for angle in normal_sph[:2])
            normal = [np.sin(polar) * np.cos(azimuth), np.sin(polar) * np.sin(azimuth), np.cos(polar)]
        self.normal = normal
        self.parax_roc = None
        self._measure_z_range()
        self.lock()

    def _measure_z_range(self) -> None:
        """The extreme heights lie on the edge, in the direction of the projected normal and opposite to it
        (tilted_surface.py:46-50)."""
        azimuth = np.arctan2(self.normal[1], self.normal[0])
        ex, ey = self.r * np.cos(azimuth), self.r * np.sin(azimuth)
        ends = [self.pos[2] + self._values_rel_host(np.array([sx]), np.array([sy]))[0] for sx, sy in ((ex, ey), (-ex, -ey))]
        self.z_min, self.z_max = min(ends), max(ends)

    @property
    def info(self) -> str:
        nx, ny, nz = self.normal
        return f"{super().info}, normal = [{nx:.4f}, {ny:.4f}, {nz:.4f}]"

    def _values_rel_host(self, x, y):
        nx, ny, nz = self.normal
        return x * (-nx / nz) + y * (-ny / nz)

    def _turn_normal(self, nx: float, ny: float) -> None:
        with self._edit():   # unit length is kept by both operations: stored as it is, not normalised again
            Surface.__setattr__(self, "normal", np.array([nx, ny, self.normal[2]], dtype=np.float64))

    def flip(self) -> None:
        """Flip around the x-axis: [x, y, z] -> [x, -y, -z], negated to point towards +z: [-x, y, z]."""
        self._turn_normal(-self.normal[0], self.normal[1])

    def rotate(self, angle: float) -> None:
        self._turn_normal(*self._rotate_rc(self.normal[0], self.normal[1], np.deg2rad(angle)))

    def _desc(self):
        d = super()._desc()
        d.normal[:] = [float(v) for v in self.normal]
        return d

    def __setattr__(self, key: str, val: Any) -> None:
        if key == "normal" and val is not None:
            check_type(key, val, (list, np.ndarray))
            val = np.asarray_chkfinite(val, dtype=np.float64) / np.linalg.norm(val)
            check_above("normal[2]", val[2], 0)
        super().__setattr__(key, val)


# ---- data surfaces ---------------------------------------------------------------------------------------------
def _mirrored(r0: np.ndarray, z: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """Radial profile 0..r continued to -r..r: symmetric, the centre once (data_surface_2d.py:67-70)."""
    return np.concatenate((-r0[:0:-1], r0)), np.concatenate((z[:0:-1], z))


class DataSurface2D(Surface):
    """Surface given by a square grid of heights (data_surface_2d.py:10-227), interpolated by a quartic
    tensor-product B-spline (RectBivariateSpline(kx=4, ky=4)) that the device evaluates."""

    rotational_symmetry = False
    _1D = False
    _kind = _capi.SURF_DATA2D

    def _reset_frame(self) -> None:
        self._sign, self._angle = 1, 0
        self._interp, self._offset, self._tab = None, 0., None

    def _label(self) -> str:
        return f"{type(self).__name__} {self.get_desc(hex(id(self)))}"

    def __init__(self, r: float, data: np.ndarray, parax_roc: float = None, **kwargs) -> None:
        Surface.__init__(self, r, **kwargs)
        self._reset_frame()
        self.parax_roc = parax_roc

        check_type("data", data, (np.ndarray, list))
        heights = np.array(np.asarray_chkfinite(data, dtype=np.float64))
        samples = heights.shape[0]
        if samples < 50:
            raise ValueError("For a good surface representation 'data' should have at least 50 values per dimension")
        if samples < 200:
            warning(f"{self._label()}: At least 200 values per dimension are advised for a 'data' matrix, but got "
                    f"{samples} values for surface {self.get_desc(hex(id(self)))}.")
        given = self._fit_profile(heights) if self._1D else self._fit_grid(heights)

        # the spline can overshoot the samples (data_surface_2d.py:108-119)
        fitted = self.z_max - self.z_min
        if abs(given - fitted) > self.N_EPS:
            growth = (fitted - given) / given
            alarm = ("WARNING: Deviations this high can be due to noise or abrupt changes in the data. "
                     "DO NOT USE SUCH SURFACES HERE.") if growth > 0.05 else ""
            warning(f"{self._label()}: Due to biquadratic interpolation the z_range of the surface has increased "
                    f"from {given:.9g} to {fitted:.9g}, a change of {growth*100:.5g}%. {alarm}")
        self.lock()

    def _fit_profile(self, z: np.ndarray) -> float:
        """Spline through the mirrored radial profile (data_surface_2d.py:60-86); returns the z range of the data."""
        if z.ndim != 1:
            raise ValueError("data array needs to have exactly one dimension.")
        z -= z[0]
        radii = np.linspace(0., self.r, z.shape[0])
        self._interp = scipy.interpolate.InterpolatedUnivariateSpline(*_mirrored(radii, z), k=_K)
        self._tab, self._nknots = _table_1d(self._interp)
        self._offset = float(self._call(0, 0))
        self._measure_profile_range()
        return float(z.max() - z.min())

    def _measure_profile_range(self) -> None:
        radii = np.linspace(0., self.r, 10000)
        sag = self._values_rel_host(radii, np.zeros(radii.shape))
        if getattr(self, "mask_func", None) is not None:   # function_surface_2d.py:86-90: where the mask holds
            sag = sag[self._mask_host(radii + self.pos[0], self.pos[1] + np.zeros(radii.shape))]
        self.z_min, self.z_max = float(sag.min()), float(sag.max())

    def _fit_grid(self, Z: np.ndarray) -> float:
        """Tensor-product spline through the square grid (data_surface_2d.py:88-104); returns the z range of the
        samples inside the disc."""
        if Z.ndim != 2:
            raise ValueError("data array needs to have exactly two dimensions.")
        n = Z.shape[1]
        if Z.shape[0] != n:
            raise ValueError("Array 'data' needs to be of square shape.")
        c = n // 2
        # height at the centre goes; for odd n the reference averages these four samples, in this order
        Z -= Z[[c, c + 1, c, c + 1], [c, c, c + 1, c + 1]].mean() if n % 2 else Z[c, c]
        axis = np.linspace(-self.r, self.r, n)
        self._interp = scipy.interpolate.RectBivariateSpline(axis, axis, Z, kx=_K, ky=_K)
        self._tab, self._nknots = _table_2d(self._interp)
        self._offset = float(self._call(0, 0))
        self.z_min, self.z_max = self._find_bounds()
        inside = self._mask_host(*(g.ravel() for g in np.meshgrid(axis, axis))).reshape(Z.shape)
        return float(Z[inside].max() - Z[inside].min())

    def _find_bounds(self) -> tuple[float, float]:
        """z range from sunflower sampling of the disc plus its edge (surface.py:57-93); set-up only."""
        count = 50000
        i = np.arange(count, dtype=np.float64)
        rho, turn = np.sqrt(i / count) * self.r, 2 * np.pi * (1 + 5 ** 0.5) / 2 * i
        sx, sy = rho * np.cos(turn), rho * np.sin(turn)
        area = np.array(self._values_rel_host(sx, sy), dtype=np.float64)
        area = area[self._mask_host(sx - self.pos[0], sy - self.pos[1])]
        ex, ey, ez = self.edge(3001)
        rim = (ez - self.pos[2])[self._mask_host(ex, ey)]
        both = np.concatenate((area, rim))
        return float(np.nanmin(both)), float(np.nanmax(both))

    def _call(self, x, y, **kwargs):
        if self._1D:
            return self._interp(np.hypot(x, y), **kwargs)
        return self._interp(x, y, grid=False, **kwargs)

    def _values_rel_host(self, x, y):
        if not self.rotational_symmetry:
            x, y = self._rotate_rc(x, y, -self._angle)
        return self._sign * (self._call(x, self._sign * y) - self._offset)

    def flip(self) -> None:
        with self._edit():
            self._sign = -self._sign
            if self.parax_roc is not None:
                self.parax_roc = -self.parax_roc
            self._mirror_z_range()

    def rotate(self, angle: float) -> None:
        if self.rotational_symmetry:
            return
        with self._edit():
            self._angle = self._angle + np.deg2rad(angle)

    def _desc(self):
        d = super()._desc()
        d.sign, d.offset, d.angle = float(self._sign), float(self._offset), float(self._angle)
        d.tab = self._tab.ctypes.data_as(_capi.C.POINTER(_capi.C.c_double))  # kept alive by this surface
        d.tab_len, d.nknots = int(self._tab.shape[0]), int(self._nknots)
        return d


class DataSurface1D(DataSurface2D):
    """Rotationally symmetric surface from an equi-spaced radial profile 0..r (data_surface_1d.py:6-30)."""

    rotational_symmetry = True
    _1D = True
    _kind = _capi.SURF_DATA1D


def _nearest_finite(Z: np.ndarray) -> np.ndarray:
    """Samples that are not finite (func undefined outside its mask) replaced by the nearest finite one."""
    bad = ~np.isfinite(Z)
    if not bad.any():
        return Z
    import scipy.ndimage
    nearest = scipy.ndimage.distance_transform_edt(bad, return_distances=False, return_indices=True)
    return Z[tuple(nearest)]


# ---- function surfaces -------------------------------------------------------------------------------------------
class FunctionSurface2D(DataSurface2D):
    """Surface defined by a Python callable z = func(x, y) (function_surface_2d.py:12-309), carried on the
    device as a quartic spline of `N_SAMPLES` x `N_SAMPLES` samples taken at construction.

    Differences to the reference, which calls the Python function for every ray:
    * values and normals come from the spline; the residual against `func` on a staggered grid inside the disc
      is measured at construction and a warning is raised above `TAB_TOL` (relative to r);
    * `deriv_func` is used only for that check -- normals are the spline's analytic derivative;
    * samples outside the disc (the square's corners) are filled by a quadratic radial continuation from the
      edge, so `func` is never evaluated outside r;
    * `mask_func` travels as a bitmap of `N_MASK` x `N_MASK` cells over the square around the disc (`N_MASK_1D`
      cells along the radius for the 1-D class), sampled at the cell centres at construction: a position closer to the
      mask's edge than one cell (2 r / N_MASK) can be classified differently from the callable.  Mask edges that fall on
      cell borders are exact.  `func` is still sampled on the whole disc; where it is not finite outside the mask the
      nearest finite sample stands in.
    """

    N_SAMPLES: tuple = (17, 33, 65, 129, 257, 401, 801, 1601)
    """samples per dimension of the tabulation grid (2D): densities are tried in turn until the residual
    meets TAB_TOL (smooth surfaces keep small tables that stay in L2)"""
    N_SAMPLES_1D: tuple = (17, 33, 65, 129, 257, 513, 1025, 2049, 4001)
    """samples of the radial profile (1D), tried in turn like N_SAMPLES: a finer grid than the function needs only
    amplifies the rounding noise of func in the slopes (noise / spacing)"""
    N_MASK: int = 4096       #: cells per dimension of the mask bitmap (2D): 2 MiB, resolution 2 r / 4096
    N_MASK_1D: int = 65536   #: cells of the radial mask bitmap (1D)
    TAB_TOL: float = 1e-9  #: accepted spline residual relative to r
    GRAD_TOL: float = 2e-9  #: accepted residual of the spline gradient (2D; against central differences of func)

    rotational_symmetry = False
    _1D = False
    _kind = _capi.SURF_DATA2D

    def __init__(self, r: float, func: Callable, mask_func: Callable = None, deriv_func: Callable = None,
                 func_args: dict = {}, mask_args: dict = {}, deriv_args: dict = {}, z_min: float = None,
                 z_max: float = None, parax_roc: float = None, **kwargs) -> None:
        Surface.__init__(self, r, **kwargs)
        self._reset_frame()
        self.func, self._func_args = func, func_args
        self.mask_func, self._mask_args = mask_func, mask_args
        self.deriv_func, self._deriv_args = deriv_func, deriv_args
        self.parax_roc = parax_roc
        self._f0 = self._eval_func(np.array([0.]), np.array([0.]))[0]  # the reference's _offset (centre value)
        self._tabulate()
        if mask_func is not None:
            self._tab = np.concatenate((self._tab, self._tabulate_mask()))   # behind the spline tables
        self._set_zmin_zmax(z_min, z_max)
        self.lock()

    # the user function in the surface's own frame, with the reference's return type checks
    def _eval_func(self, x: np.ndarray, y: np.ndarray) -> np.ndarray:
        where = (np.sqrt(x ** 2 + y ** 2),) if self._1D else (x, y)   # sqrt form: function_surface_2d.py:143
        out = self.func(*where, **self._func_args)
        if not isinstance(out, np.ndarray):
            raise RuntimeError(f"func must return a np.ndarray, but returns type {type(out)}.")
        if out.shape[0] and not isinstance(out[0], np.float64):
            raise RuntimeError("Elements of return value of func must be of type np.float64")
        return out

    def _eval_mask(self, x: np.ndarray, y: np.ndarray) -> np.ndarray:
        """mask_func in the surface's own frame, with the reference's return type checks
        (function_surface_2d.py:175-187)."""
        where = (np.sqrt(x ** 2 + y ** 2),) if self._1D else (x, y)
        out = self.mask_func(*where, **self._mask_args)
        if not isinstance(out, np.ndarray):
            raise RuntimeError(f"mask_func must return a np.ndarray, but returns type {type(out)}.")
        if out.shape[0] and not isinstance(out[0], (bool, np.bool_)):
            raise RuntimeError("Elements of return value of mask_func must be of type bool")
        return out

    def _mask_host(self, x, y):
        """Set-up helper: inside r and, with a mask_func, inside that too (function_surface_2d.py:158-191)."""
        inside = Surface._mask_host(self, x, y)
        if self.mask_func is None:
            return inside
        dx, dy = np.asarray(x, dtype=np.float64) - self.pos[0], np.asarray(y, dtype=np.float64) - self.pos[1]
        if not self._1D:
            dx, dy = self._rotate_rc(dx, dy, -self._angle)
            dy = self._sign * dy
        return inside & self._eval_mask(dx, dy)

    def _tabulate_mask(self) -> np.ndarray:
        """The mask bitmap for the device (layout: include/optrace_amd.h, OT_SURF_FLAG_MASK_TABLE): mask_func at the
        cell centres, in the function's own frame; -> cell count followed by the packed bits, as float64 words."""
        R = self.r
        if self._1D:
            n = int(self.N_MASK_1D)
            centres = (np.arange(n) + 0.5) * (R / n)
            bits = np.asarray(self._eval_mask(centres, np.zeros(n)), dtype=bool)
        else:
            n = int(self.N_MASK)
            centres = -R + (np.arange(n) + 0.5) * (2 * R / n)
            bits = np.empty((n, n), dtype=bool)   # [iy, ix]
            rows = max(1, (1 << 20) // n)         # about a million points per call of mask_func
            for y0 in range(0, n, rows):
                X, Y = np.meshgrid(centres, centres[y0:y0 + rows])
                bits[y0:y0 + rows] = np.asarray(self._eval_mask(X.ravel(), Y.ravel()), dtype=bool).reshape(X.shape)
        packed = np.packbits(bits.ravel(), bitorder="little")
        packed = np.concatenate((packed, np.zeros(-packed.size % 8, dtype=np.uint8)))
        return np.concatenate(([float(n)], packed.view(np.float64)))

    def _values_rel_host(self, x, y):
        """Set-up helper (z range, geometry checks): the function itself, as the reference evaluates it
        (function_surface_2d.py:133-156)."""
        x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
        if self._1D:
            vals = self._eval_func(x, y)
        else:
            x_, y_ = self._rotate_rc(x, y, -self._angle)
            vals = self._eval_func(x_, self._sign * y_)
        return self._sign * (vals - self._f0)

    def _tabulate(self) -> None:
        R = self.r
        if self._1D:
            # step of the reference's central differences (surface.py:262-266): what its normals are made of
            eps = max((3 * np.finfo(np.float64).eps * 50) ** (1 / 3), float(np.spacing(2 * R)))
            best = None
            for n in self.N_SAMPLES_1D:
                r0 = np.linspace(0, R, n)
                Z = self._sample(r0, np.zeros_like(r0))
                self._interp = scipy.interpolate.InterpolatedUnivariateSpline(*_mirrored(r0, _nearest_finite(Z)), k=_K)
                # residuals at the interval midpoints and at points anywhere on the profile: values, and slopes against
                # central differences of func
                rm = np.concatenate(((r0[1:] + r0[:-1]) / 2, R * np.random.default_rng(n).random(512)))
                rm = rm[self._checked(rm, np.zeros_like(rm))]
                res = np.abs(self._interp(rm) - (self._eval_func(rm, np.zeros_like(rm)) - self._f0))
                gres = 0.
                rg = rm[(rm > eps) & (rm < R - eps)]
                if rg.size:
                    zero = np.zeros_like(rg)
                    fd = (self._eval_func(rg + eps, zero) - self._eval_func(rg - eps, zero)) / (2 * eps)
                    gres = float(np.max(np.abs(self._interp.derivative()(rg) - fd)))
                ok = res.max(initial=0.) <= self.TAB_TOL * R
                if best is None or (ok, -gres) > (best[0], -best[1]):
                    best = (ok, gres, self._interp, res)
                if ok and gres <= self.GRAD_TOL:
                    break
            # no density met both tolerances (func itself is noisy at this scale, e.g. a sag of 1e-7 on an offset of
            # 80): the one with the best slopes among those whose values fit
            _, gres, self._interp, res = best
            self._tab, self._nknots = _table_1d(self._interp)
            self._grad_residual = gres
            self._finish_tabulation(res)
            return
        best = None
        for n in self.N_SAMPLES:
            res, gres = self._tabulate_2d(n)
            ok = res.size == 0 or res.max() <= self.TAB_TOL * R
            if best is None or (ok, -gres) > (best[0], -best[1]):
                best = (ok, gres, n)
            if res.size == 0 or (ok and gres <= self.GRAD_TOL):
                break
        if best[2] != n:  # no density met both tolerances: back to the one with the best slopes (see the 1-D case)
            res, gres = self._tabulate_2d(best[2])
        self._grad_residual = gres
        self._finish_tabulation(res)

    def _tabulate_2d(self, n: int) -> np.ndarray:
        """Fit the quartic spline to n x n samples; returns the residuals at the cell centres inside the disc
        and the largest gradient residual there (central differences of func, step as in surface.py:262-266)."""
        R = self.r
        xy = np.linspace(-R, R, n)
        X, Y = np.meshgrid(xy, xy, indexing="ij")
        rr = np.hypot(X, Y)
        inside = rr <= R
        Z = np.zeros_like(X)
        Z[inside] = self._sample(X[inside], Y[inside])
        # corners of the square (never touched by the reference): the function itself where it is defined there,
        # otherwise a quadratic radial continuation from three points at the edge
        out = ~inside
        if np.any(out):
            try:
                with np.errstate(all="ignore"):
                    Z[out] = self._eval_func(X[out], Y[out]) - self._f0
            except Exception:
                Z[out] = np.nan
            bad = out & ~np.isfinite(Z)
            if np.any(bad):
                ux, uy, d = X[bad] / rr[bad], Y[bad] / rr[bad], rr[bad] - R
                h = R / (n - 1)
                f0 = self._eval_func(ux * R, uy * R) - self._f0
                f1 = self._eval_func(ux * (R - h), uy * (R - h)) - self._f0
                f2 = self._eval_func(ux * (R - 2 * h), uy * (R - 2 * h)) - self._f0
                d1 = (3 * f0 - 4 * f1 + f2) / (2 * h)
                d2 = (f0 - 2 * f1 + f2) / h ** 2
                Z[bad] = f0 + d1 * d + d2 / 2 * d ** 2
        self._interp = scipy.interpolate.RectBivariateSpline(xy, xy, _nearest_finite(Z), kx=_K, ky=_K)
        self._tab, self._nknots = _table_2d(self._interp)
        xm = (xy[1:] + xy[:-1]) / 2
        Xm, Ym = np.meshgrid(xm, xm, indexing="ij")
        sel = np.hypot(Xm, Ym) <= R
        xs, ys = Xm[sel], Ym[sel]
        # cell centres plus points anywhere in the disc (a coarse grid must not pass because an oscillation happens
        # to vanish at its nodes and centres)
        rng = np.random.default_rng(n)
        pr, pa = R * np.sqrt(rng.random(2048)), 2 * np.pi * rng.random(2048)
        xs, ys = np.concatenate((xs, pr * np.cos(pa))), np.concatenate((ys, pr * np.sin(pa)))
        keep = self._checked(xs, ys)
        xs, ys = xs[keep], ys[keep]
        res = np.abs(self._interp(xs, ys, grid=False) - (self._eval_func(xs, ys) - self._f0))
        eps = (3 * np.finfo(np.float64).eps * 50) ** (1 / 3)
        keep = np.hypot(xs, ys) <= R - max(2 * R / (n - 1), 2 * eps)
        xs, ys = xs[keep], ys[keep]
        if xs.shape[0] > 6000:
            xs, ys = xs[::xs.shape[0] // 6000 + 1], ys[::ys.shape[0] // 6000 + 1]  # a subset is enough for the gradient
        gx = (self._eval_func(xs + eps, ys) - self._eval_func(xs - eps, ys)) / (2 * eps)
        gy = (self._eval_func(xs, ys + eps) - self._eval_func(xs, ys - eps)) / (2 * eps)
        gres = max(np.abs(self._interp(xs, ys, dx=1, grid=False) - gx).max(initial=0.),
                   np.abs(self._interp(xs, ys, dy=1, grid=False) - gy).max(initial=0.))
        return res, float(gres)

    def _sample(self, x: np.ndarray, y: np.ndarray) -> np.ndarray:
        """func - func(0, 0) for the tabulation; outside a mask_func the function may be undefined (not finite)."""
        with np.errstate(all="ignore"):
            return self._eval_func(x, y) - self._f0

    def _checked(self, x: np.ndarray, y: np.ndarray) -> np.ndarray:
        """Which of the points (own frame) count for the residuals of the tabulation: those the mask_func keeps."""
        if self.mask_func is None:
            return np.ones(x.shape, dtype=bool)
        return np.asarray(self._eval_mask(x, y), dtype=bool)

    def _finish_tabulation(self, res: np.ndarray) -> None:
        R = self.r
        self._offset = 0.  # the samples already have the centre value removed
        self._tab_residual = float(res.max()) if res.size else 0.
        if self._tab_residual > self.TAB_TOL * R:
            warning(f"{type(self).__name__}: the spline tabulation of func deviates by up to "
                    f"{self._tab_residual:.3g} mm from the function; raise N_SAMPLES or smooth the function.")

    def _set_zmin_zmax(self, z_min: float, z_max: float) -> None:
        """z range: measured from the surface, or the user's values if they are plausible against the measured
        ones (function_surface_2d.py:76-128).  User values refer to func itself, the stored ones to func - func(0)."""
        if self._1D:
            self._measure_profile_range()
        else:
            self.z_min, self.z_max = self._find_bounds()
        name, centre, tol = self._label(), self._f0, self.N_EPS
        if (z_min is None) != (z_max is None):
            raise ValueError("z_max and z_min need to be both None or both need a value")
        if z_min is None:
            warning(f"Estimated z-bounds of {name}: [{centre+self.z_min:.9g}, "
                    f"{centre+self.z_max:.9g}], provide actual values for higher precision.")
            return
        measured, stated = self.z_max - self.z_min, z_max - z_min
        if measured and stated + tol < measured:
            warning(f"{name}: Provided a z-extent of {stated},but measured range is at least {measured}, an increase "
                    f"of at least {100*(measured - stated)/measured:.5g}. I will use the measured values for now.")
            return
        slack = 1.2
        if stated > slack * measured:
            warning(f"{name}: Provided z-range is more than {(slack-1)*100:.5g}% larger than measured z-range")
        top, bottom = self.z_max + centre, self.z_min + centre
        if z_max + tol < top:
            warning(f"{name}: Provided z_max={z_max} lower than measured value of {top}. "
                    f"Using the measured values for now")
        elif z_min - tol > bottom:
            warning(f"{name}: Provided z_min={z_min} higher than measured value of {bottom}. "
                    f"Using the measured values for now")
        else:
            self.z_min, self.z_max = z_min - centre, z_max - centre

    def _desc(self):
        d = super()._desc()
        if self.deriv_func is not None and not self._1D:
            # the reference calls deriv_func at the unrotated coordinates (function_surface_2d.py:235)
            d.flags |= _capi.SURF_FLAG_DERIV_UNROTATED
        if self.mask_func is not None:
            d.flags |= _capi.SURF_FLAG_MASK_TABLE
        return d

    def __setattr__(self, key: str, val: Any) -> None:
        if key in ("z_max", "z_min"):
            check_type(key, val, (float, int))
            val = float(val)
        elif key == "func" and not callable(val):
            raise TypeError("func needs to be callable.")
        elif key in ("deriv_func", "mask_func") and not (val is None or callable(val)):
            raise TypeError(f"{key} needs to be callable or None.")
        elif key in ("_deriv_args", "_func_args", "_mask_args"):
            check_type(key, val, dict)
            val = copy.deepcopy(val)
        super().__setattr__(key, val)


class FunctionSurface1D(FunctionSurface2D):
    """Rotationally symmetric surface z = func(r) (function_surface_1d.py:8-50), tabulated like FunctionSurface2D."""

    rotational_symmetry = True
    _1D = True
    _kind = _capi.SURF_DATA1D
