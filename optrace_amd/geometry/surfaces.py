"""Surface shapes of the sequential tracer.

Host-side mirror of optrace/tracer/geometry/surface/*.py: the classes keep the reference's
constructor arguments, attributes and error behaviour and know how to describe themselves to the
device (`_desc()` -> `ot_surface`).  The per-ray operators (`find_hit`, `normals`, `mask`, `values`,
`hurb_props`) run on the GPU through the C-ABI (include/optrace_amd.h); only scene set-up code
(z-range of a surface, collision checks) evaluates `_mask_host` / `_values_host` in NumPy.
"""
from __future__ import annotations

from typing import Any

import numpy as np

from .. import _capi
from ..base import BaseClass, check_above, check_type
from .._warn import warning


class Surface(BaseClass):
    """Common behaviour of all surfaces (reference: surface.py:15-496)."""

    C_EPS: float = 1e-6   #: solution precision of numerical hit finding (surface.py:17)
    N_EPS: float = 1e-10  #: comparison epsilon (surface.py:20)
    rotational_symmetry: bool = False
    _kind: int = _capi.SURF_CIRCLE

    def __init__(self, r: float, **kwargs) -> None:
        self._lock = False
        self.pos = np.asarray_chkfinite([0., 0., 0.], dtype=np.float64)
        self.r = r
        self.parax_roc = None
        self.z_min, self.z_max = np.nan, np.nan
        super().__init__(**kwargs)

    # ---- geometry bookkeeping (host) ------------------------------------------------------------
    def is_flat(self) -> bool:
        return self.z_max == self.z_min

    @property
    def info(self) -> str:
        return (f"{type(self).__name__}, pos = [{self.pos[0]:.5g} mm, {self.pos[1]:.5g} mm, "
                f"{self.pos[2]:.5g} mm], r = {self.r:.5g} mm")

    def move_to(self, pos) -> None:
        """Move the surface centre; z_min/z_max shift by the z difference (surface.py:95-110)."""
        self._lock = False
        self.z_min += pos[2] - self.pos[2]
        self.z_max += pos[2] - self.pos[2]
        self.pos = np.asarray_chkfinite(pos, dtype=np.float64)
        self.lock()

    @property
    def extent(self) -> tuple:
        """Smallest box around the surface: (x0, x1, y0, y1, z0, z1) (surface.py:113-120)."""
        return (*(self.r * np.array([-1, 1, -1, 1]) + self.pos[:2].repeat(2)), self.z_min, self.z_max)

    @property
    def ds(self) -> float:
        return float(self.z_max - self.z_min)

    @property
    def dn(self) -> float:
        return float(self.pos[2] - self.z_min)

    @property
    def dp(self) -> float:
        return float(self.z_max - self.pos[2])

    def flip(self) -> None:
        assert self.is_flat()

    def rotate(self, angle: float) -> None:
        assert self.rotational_symmetry

    @staticmethod
    def _rotate_rc(x, y, alpha):
        if alpha:
            return x * np.cos(alpha) - y * np.sin(alpha), x * np.sin(alpha) + y * np.cos(alpha)
        return x, y

    # ---- host evaluation used by scene set-up only -----------------------------------------------
    def _values_rel_host(self, x: np.ndarray, y: np.ndarray) -> np.ndarray:
        return np.broadcast_to(0., x.shape)

    def _mask_host(self, x: np.ndarray, y: np.ndarray) -> np.ndarray:
        x0, y0, _ = self.pos
        return (x - x0) ** 2 + (y - y0) ** 2 <= (self.r + self.N_EPS) ** 2

    def _values_host(self, x: np.ndarray, y: np.ndarray) -> np.ndarray:
        """Surface height in absolute coordinates with radial edge continuation (surface.py:137-164)."""
        if self.is_flat():
            return np.broadcast_to(self.z_max, x.shape)
        z = np.full_like(x, self.z_max, dtype=np.float64)
        inside = self._mask_host(x, y)
        z[inside] = self.pos[2] + self._values_rel_host(x[inside] - self.pos[0], y[inside] - self.pos[1])
        if np.any(~inside):
            r = self.r - self.N_EPS
            z[~inside] = self.pos[2] + self._values_rel_host(np.array([r]), np.array([0.]))[0]
        return z

    def edge(self, nc: int):
        """nc points on the surface edge (surface.py:287-304)."""
        if nc < 20:
            raise ValueError("Expected at least nc=20")
        theta = np.linspace(-3 / 4 * np.pi, 5 / 4 * np.pi, nc)
        xd, yd = self.r * np.cos(theta), self.r * np.sin(theta)
        zd = self._values_rel_host(xd, yd)
        return xd + self.pos[0], yd + self.pos[1], zd + self.pos[2]

    # ---- device descriptor ----------------------------------------------------------------------
    def _desc(self) -> _capi.Surface:
        d = _capi.Surface()
        d.kind = self._kind
        d.pos[:] = [float(v) for v in self.pos]
        d.r = float(self.r)
        d.z_min, d.z_max = float(self.z_min), float(self.z_max)
        return d

    # ---- per-ray operators: GPU --------------------------------------------------------------------
    def find_hit(self, p: np.ndarray, s: np.ndarray, where=None):
        """Intersections of rays (p, s) with the surface -> (p_hit, is_hit, ill).

        Same contract as Surface.find_hit (surface.py:307) / ConicSurface.find_hit
        (conic_surface.py:126); computed by `ot_surface_find_hit` on the GPU.
        """
        from .. import ops
        p_hit, is_hit, ill = ops.surface_find_hit(self._desc(), p, s)
        w = where if where is not None else slice(None)
        numeric = not self.is_flat() and self._kind >= _capi.SURF_ASPHERE
        return p_hit[w], is_hit[w], (ill[w] if numeric else np.array([]))

    def normals(self, x: np.ndarray, y: np.ndarray) -> np.ndarray:
        """Unit normals at (x, y) (surface.py:247, conic_surface.py:70, function_surface_2d.py:202)."""
        from .. import ops
        return ops.surface_normals(self._desc(), x, y)

    def mask(self, x: np.ndarray, y: np.ndarray) -> np.ndarray:
        """True where the surface is defined (surface.py:235)."""
        from .. import ops
        return ops.surface_mask(self._desc(), x, y)

    def values(self, x: np.ndarray, y: np.ndarray) -> np.ndarray:
        """Surface height in absolute coordinates (surface.py:137)."""
        from .. import ops
        return ops.surface_values(self._desc(), x, y)

    def __setattr__(self, key: str, val: Any) -> None:
        if key == "r":
            check_type(key, val, (float, int))
            val = float(val)
            check_above(key, val, 0)
        elif key == "parax_roc" and val is not None:
            check_type(key, val, (float, int))
            val = float(val)
        super().__setattr__(key, val)


class CircularSurface(Surface):
    """Flat disc (circular_surface.py:9-44)."""

    rotational_symmetry = True
    _kind = _capi.SURF_CIRCLE

    def __init__(self, r: float, **kwargs) -> None:
        self._lock = False
        super().__init__(r, **kwargs)
        self.parax_roc = np.inf
        self.z_min = self.z_max = self.pos[2]
        self.lock()


class RingSurface(Surface):
    """Flat annulus r_i <= r <= r_o (ring_surface.py:10-161)."""

    rotational_symmetry = True
    _kind = _capi.SURF_RING

    def __init__(self, r: float, ri: float, **kwargs) -> None:
        self._lock = False
        super().__init__(r, **kwargs)
        self.r, self.ri = r, ri
        self.parax_roc = np.inf
        self.z_min = self.z_max = self.pos[2]
        if ri >= r:
            raise ValueError("ri needs to be smaller than r.")
        self.lock()

    def _mask_host(self, x, y):
        x0, y0, _ = self.pos
        r2 = (x - x0) ** 2 + (y - y0) ** 2
        return ((self.ri - self.N_EPS) ** 2 <= r2) & (r2 <= (self.r + self.N_EPS) ** 2)

    def _desc(self):
        d = super()._desc()
        d.ri = float(self.ri)
        return d

    def hurb_props(self, x: np.ndarray, y: np.ndarray):
        """Edge distances / axis for HURB (ring_surface.py:88-121), on the GPU."""
        from .. import ops
        return ops.surface_hurb_props(self._desc(), x, y)

    def __setattr__(self, key, val):
        if key == "ri":
            check_type(key, val, (float, int))
            val = float(val)
            check_above(key, val, 0)
        super().__setattr__(key, val)


class RectangularSurface(Surface):
    """Flat rectangle, optionally rotated about z (rectangular_surface.py:10-177)."""

    rotational_symmetry = False
    _kind = _capi.SURF_RECT

    def __init__(self, dim, **kwargs) -> None:
        self._lock = False
        self._angle = 0
        super().__init__(1, **kwargs)
        self.dim = np.asarray_chkfinite(dim, dtype=np.float64)
        self.parax_roc = np.inf
        self.z_min = self.z_max = self.pos[2]
        self.lock()

    @property
    def info(self) -> str:
        return (f"{type(self).__name__}, pos = [{self.pos[0]:.5g} mm, {self.pos[1]:.5g} mm, "
                f"{self.pos[2]:.5g} mm], dim = [{self.dim[0]:.5g} mm, {self.dim[1]:.5g} mm]")

    @property
    def extent(self) -> tuple:
        sx = np.abs(self.dim[0] * np.cos(self._angle)) + np.abs(self.dim[1] * np.sin(self._angle))
        sy = np.abs(self.dim[0] * np.sin(self._angle)) + np.abs(self.dim[1] * np.cos(self._angle))
        return (self.pos[0] - sx / 2, self.pos[0] + sx / 2, self.pos[1] - sy / 2, self.pos[1] + sy / 2,
                self.z_min, self.z_max)

    @property
    def _extent(self) -> tuple:
        return -self.dim[0] / 2, self.dim[0] / 2, -self.dim[1] / 2, self.dim[1] / 2, 0., 0.

    def rotate(self, angle: float) -> None:
        self._lock = False
        self._angle += np.deg2rad(angle)
        self.lock()

    def flip(self) -> None:
        self._lock = False
        self._angle *= -1
        self.lock()

    def _mask_host(self, x, y):
        xr, yr = self._rotate_rc(x - self.pos[0], y - self.pos[1], -self._angle)
        xs, xe, ys, ye = self._extent[:4]
        e = self.N_EPS
        return (xs - e <= xr) & (xr <= xe + e) & (ys - e <= yr) & (yr <= ye + e)

    def _desc(self):
        d = super()._desc()
        d.dim[:] = [float(self.dim[0]), float(self.dim[1])]
        d.angle = float(self._angle)
        return d

    def __setattr__(self, key, val):
        if key == "dim":
            check_type(key, val, np.ndarray)
            if val.ndim != 1 or val.shape[0] != 2:
                raise TypeError("dim needs to have two elements.")
            if val[0] <= 0 or val[1] <= 0:
                raise ValueError(f"Dimensions dim need to be positive, but are {val}")
        super().__setattr__(key, val)


class SlitSurface(RectangularSurface):
    """Rectangle with a rectangular opening (slit_surface.py:12-123)."""

    _kind = _capi.SURF_SLIT

    def __init__(self, dim, dimi, **kwargs) -> None:
        super().__init__(dim, **kwargs)
        self._lock = False
        self._new_lock = False
        self.dimi = np.asarray_chkfinite(dimi, dtype=np.float64)
        self.lock()

    def _mask_host(self, x, y):
        xr, yr = self._rotate_rc(x - self.pos[0], y - self.pos[1], -self._angle)
        xs, xe, ys, ye = -self.dimi[0] / 2, self.dimi[0] / 2, -self.dimi[1] / 2, self.dimi[1] / 2
        e = self.N_EPS
        inside = (xs + e <= xr) & (xr <= xe - e) & (ys + e <= yr) & (yr <= ye - e)
        return super()._mask_host(x, y) & ~inside

    def _desc(self):
        d = super()._desc()
        d.dimi[:] = [float(self.dimi[0]), float(self.dimi[1])]
        return d

    def hurb_props(self, x: np.ndarray, y: np.ndarray):
        """Edge distances / axis for HURB (slit_surface.py:65-87), on the GPU."""
        from .. import ops
        return ops.surface_hurb_props(self._desc(), x, y)

    def __setattr__(self, key, val):
        if key == "dimi":
            check_type(key, val, np.ndarray)
            if val.ndim != 1 or val.shape[0] != 2:
                raise TypeError("dimi needs to have two elements.")
            if val[0] >= self.dim[0] or val[1] >= self.dim[1]:
                raise ValueError("Dimensions dimi must be smaller than dimension dim.")
            if val[0] <= 0 or val[1] <= 0:
                raise ValueError(f"Dimensions dimi need to be positive, but are {val}")
        super().__setattr__(key, val)


class ConicSurface(Surface):
    """Conic section z(r) = rho r^2 / (1 + sqrt(1 - (k+1) rho^2 r^2)) (conic_surface.py:10-229)."""

    rotational_symmetry = True
    _kind = _capi.SURF_CONIC

    def __init__(self, r: float, R: float, k: float, **kwargs) -> None:
        self._lock = False
        super().__init__(r, **kwargs)
        self.R, self.k = R, k
        self.parax_roc = R
        if (self.k + 1) * (self.r / self.R) ** 2 >= 1:
            raise ValueError("Surface radius r larger than radius of conic section.")
        z0 = self.pos[2]
        self.z_max = 0
        z1 = z0 + self._values_rel_host(np.array([r]), np.array([0]))[0]
        self.z_min, self.z_max = min(z0, z1), max(z0, z1)
        self.lock()

    @property
    def info(self) -> str:
        return super().info + f", R = {self.R:.5g} mm, k = {self.k:.5g}"

    def _values_rel_host(self, x, y):
        k, rho = self.k, 1 / self.R
        r2 = x ** 2 + y ** 2
        return rho * r2 / (1 + np.sqrt(1 - (k + 1) * rho ** 2 * r2))

    def flip(self) -> None:
        self._lock = False
        self.R *= -1
        self.parax_roc *= -1
        a = self.pos[2] - (self.z_max - self.pos[2])
        b = self.pos[2] + (self.pos[2] - self.z_min)
        self.z_min, self.z_max = a, b
        self.lock()

    def _desc(self):
        d = super()._desc()
        d.R, d.k = float(self.R), float(self.k)
        return d

    def __setattr__(self, key, val):
        if key in ("R", "k"):
            check_type(key, val, (float, int))
            val = float(val)
            if key == "R" and (val == 0 or not np.isfinite(val)):
                raise ValueError("R needs to be non-zero and finite. Use planar surface types for planar surfaces.")
        super().__setattr__(key, val)


class SphericalSurface(ConicSurface):
    """Sphere cap = conic with k = 0 (spherical_surface.py:7-98)."""

    sphere_projection_methods = ["Equidistant", "Orthographic", "Equal-Area", "Stereographic"]

    def __init__(self, r: float, R: float, **kwargs) -> None:
        self._lock = False
        super().__init__(r, R, k=0, **kwargs)
        self.lock()

    @property
    def info(self) -> str:
        return Surface.info.fget(self) + f", R = {self.R:.5g} mm"

    def sphere_projection(self, p: np.ndarray, projection_method: str = "Equidistant") -> np.ndarray:
        """Map points on the sphere to a plane (spherical_surface.py:36-97), on the GPU."""
        if projection_method not in self.sphere_projection_methods:
            raise ValueError(f"Invalid projection_method {projection_method}, "
                             f"must be one of {self.sphere_projection_methods}.")
        from .. import ops
        return ops.sphere_projection(self._desc(), p, projection_method)


class AsphericSurface(Surface):
    """Conic plus even polynomial a_0 r^2 + a_1 r^4 + ... (aspheric_surface.py:9-136).

    The reference builds it on FunctionSurface1D with Python callables; here the surface function and
    its derivative are closed forms evaluated on the device, and the hit search is the same Illinois
    regula falsi (surface.py:329-414).
    """

    rotational_symmetry = True
    _kind = _capi.SURF_ASPHERE

    def __init__(self, r: float, R: float, k: float, coeff, **kwargs) -> None:
        self._lock = False
        super().__init__(r, **kwargs)
        self.k = k
        self.R = R
        self.coeff = coeff
        self.parax_roc = 1 / (1 / self.R + 2 * self.coeff[0])
        # z range by sampling the profile (function_surface_2d.py:84-88)
        rn = np.linspace(0, self.r, 10000)
        zn = self._values_rel_host(rn, np.zeros_like(rn))
        mn = self._mask_host(rn, np.zeros_like(rn))
        self.z_min, self.z_max = float(zn[mn].min()), float(zn[mn].max())
        warning(f"Estimated z-bounds of {type(self).__name__}: [{self.z_min:.9g}, {self.z_max:.9g}], "
                "provide actual values for higher precision.")
        self.lock()

    @property
    def info(self) -> str:
        return super().info + f", R = {self.R:.5g} mm, k = {self.k:.5g}\ncoeff = {self.coeff}"

    @property
    def _np_coeff(self) -> np.ndarray:
        c = np.zeros(2 * len(self.coeff) + 1, dtype=np.float64)
        c[2::2] = self.coeff
        return np.flip(c)

    def _values_rel_host(self, x, y):
        r = np.sqrt(x ** 2 + y ** 2)
        rho, k = 1 / self.R, self.k
        z = rho * r ** 2 / (1 + np.sqrt(1 - (k + 1) * rho ** 2 * r ** 2))
        z += np.polyval(self._np_coeff, r)
        return z

    def flip(self) -> None:
        self._lock = False
        self.R *= -1
        self.coeff.flags.writeable = True
        self.coeff *= -1
        self.parax_roc *= -1
        a = self.pos[2] - (self.z_max - self.pos[2])
        b = self.pos[2] + (self.pos[2] - self.z_min)
        self.z_min, self.z_max = a, b
        self.lock()

    def _desc(self):
        d = super()._desc()
        d.R, d.k = float(self.R), float(self.k)
        if len(self.coeff) > _capi.OT_MAX_ASPH:
            raise _capi.BackendError(f"AsphericSurface with more than {_capi.OT_MAX_ASPH} coefficients "
                                     "is not supported by the device kernels.")
        d.ncoeff = len(self.coeff)
        for j, c in enumerate(self.coeff):
            d.coeff[j] = float(c)
        return d

    def __setattr__(self, key, val):
        if key in ("R", "k"):
            check_type(key, val, (float, int))
            val = float(val)
            if key == "R" and (val == 0 or not np.isfinite(val)):
                raise ValueError("R needs to be non-zero and finite. Use planar surface types for planar surfaces.")
        elif key == "coeff":
            check_type(key, val, (list, np.ndarray))
            val = np.asarray_chkfinite(val, dtype=np.float64)
            if not len(val):
                raise ValueError("Empty coeff list. Provide coefficients or use ConicSurface instead.")
        super().__setattr__(key, val)


class Point(BaseClass):
    """Point source shape (point.py:7-70)."""

    def __init__(self, **kwargs) -> None:
        self._lock = False
        self.pos = np.array([0., 0., 0.], dtype=np.float64)
        super().__init__(**kwargs)
        self.lock()

    def move_to(self, pos) -> None:
        self._lock = False
        self.pos = np.asarray_chkfinite(pos, dtype=np.float64)
        self.lock()

    def flip(self) -> None:
        pass

    def rotate(self, angle: float) -> None:
        pass

    @property
    def extent(self) -> tuple:
        return tuple(self.pos.repeat(2))


class Line(BaseClass):
    """Line source shape in the xy plane (line.py:9-112)."""

    def __init__(self, r: float, angle: float = 0, **kwargs) -> None:
        self._lock = False
        self.pos = np.array([0., 0., 0.], dtype=np.float64)
        self.r = r
        self.angle = angle
        super().__init__(**kwargs)
        self.lock()

    def move_to(self, pos) -> None:
        self._lock = False
        self.pos = np.asarray_chkfinite(pos, dtype=np.float64)
        self.lock()

    def flip(self) -> None:
        self._lock = False
        self.angle *= -1
        self.lock()

    def rotate(self, angle: float) -> None:
        self._lock = False
        self.angle += angle
        self.lock()

    @property
    def extent(self) -> tuple:
        ang = np.deg2rad(self.angle)
        return (self.pos[0] - self.r * np.cos(ang), self.pos[0] + self.r * np.cos(ang),
                self.pos[1] - self.r * np.sin(ang), self.pos[1] + self.r * np.sin(ang),
                self.pos[2], self.pos[2])

    def __setattr__(self, key, val):
        if key in ("r", "angle"):
            check_type(key, val, (float, int))
            val = float(val)
            if key == "r":
                check_above(key, val, 0)
        super().__setattr__(key, val)
