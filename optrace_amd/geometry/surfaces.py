"""Surface shapes of the sequential tracer.

Host-side counterpart of optrace/tracer/geometry/surface/*.py and of point.py / line.py: the reference's constructor
arguments, attributes and error texts, around one job -- describing the shape to the device
(`_desc()` -> `ot_surface`).  The per-ray operators (`find_hit`, `normals`, `mask`, `values`, `hurb_props`,
`sphere_projection`) run on the GPU through the C-ABI (include/optrace_amd.h); only scene set-up (z range of a surface,
collision checks) evaluates the NumPy forms `_mask_host` / `_values_host`.

Shapes are read-only once built (their arrays too), so every change of state is an attribute assignment inside
`_edit()`: that is what lets `Raytracer.trace` recognise an unchanged scene by a counter (base.mutation_epoch).
"""
from __future__ import annotations

import contextlib
from typing import Any

import numpy as np

from .. import _capi
from ..base import BaseClass, check_above, check_type
from .._warn import warning


def _number(key: str, val, positive: bool = False) -> float:
    check_type(key, val, (float, int))
    val = float(val)
    if positive:
        check_above(key, val, 0)
    return val


def _pair(key: str, val: np.ndarray, what: str) -> None:
    """A 1-D array of two positive lengths, or the reference's errors."""
    check_type(key, val, np.ndarray)
    if val.shape != (2,):
        raise TypeError(f"{key} needs to have two elements.")
    if val.min() <= 0:
        raise ValueError(f"Dimensions {what} need to be positive, but are {val}")


def _side_lengths(sides) -> np.ndarray:
    return np.asarray_chkfinite(sides, dtype=np.float64)


class _Shape(BaseClass):
    """What surfaces, points and lines share: a position and locked-by-default state."""

    def __init__(self, **kwargs) -> None:
        self._lock = False
        self.pos = np.zeros(3, dtype=np.float64)
        BaseClass.__init__(self, **kwargs)

    @contextlib.contextmanager
    def _edit(self):
        """Assignments are possible inside; afterwards the object and its arrays are read-only again."""
        self._lock = False
        try:
            yield
        finally:
            self.lock()

    def _place(self, pos) -> None:
        self.pos = np.asarray_chkfinite(pos, dtype=np.float64)

    def move_to(self, pos) -> None:
        with self._edit():
            self._place(pos)

    def flip(self) -> None:
        """Turn around the x axis through the position (nothing to do for symmetric flat shapes)."""

    def rotate(self, angle: float) -> None:
        """Rotate by `angle` degrees around the z axis through the position."""


class Surface(_Shape):
    """Common behaviour of all surfaces (reference: surface.py:15-496)."""

    C_EPS: float = 1e-6   #: solution precision of numerical hit finding (surface.py:17)
    N_EPS: float = 1e-10  #: comparison epsilon (surface.py:20)
    rotational_symmetry = False   #: True for surfaces that look the same after any rotate()
    _kind: int = _capi.SURF_CIRCLE

    def __init__(self, r: float, **kwargs) -> None:
        _Shape.__init__(self, **kwargs)
        self._lock = False
        self.r = r
        self.parax_roc = None
        self.z_min = self.z_max = np.nan

    def _set_flat(self) -> None:
        """z range and paraxial curvature of a plane at the current position."""
        self.z_min = self.z_max = self.pos[2]
        self.parax_roc = np.inf

    # ---- geometry bookkeeping (host) ------------------------------------------------------------
    def is_flat(self) -> bool:
        return self.z_min == self.z_max

    @property
    def info(self) -> str:
        x, y, z = self.pos
        return f"{type(self).__name__}, pos = [{x:.5g} mm, {y:.5g} mm, {z:.5g} mm], r = {self.r:.5g} mm"

    def _place(self, pos) -> None:
        """The z range moves with the centre (surface.py:95-110)."""
        dz = pos[2] - self.pos[2]
        self.z_min, self.z_max = self.z_min + dz, self.z_max + dz
        _Shape._place(self, pos)

    def _mirror_z_range(self) -> None:
        """z range after a flip: reflected at the centre's z."""
        zc = self.pos[2]
        self.z_min, self.z_max = zc - (self.z_max - zc), zc - (self.z_min - zc)

    @property
    def extent(self) -> tuple:
        """Smallest box around the surface: (x0, x1, y0, y1, z0, z1) (surface.py:113-120)."""
        x, y = self.pos[:2]
        return x - self.r, x + self.r, y - self.r, y + self.r, self.z_min, self.z_max

    @property
    def ds(self) -> float:
        """Total z extent."""
        return float(self.z_max) - float(self.z_min)

    @property
    def dn(self) -> float:
        """z extent before the centre."""
        return float(self.pos[2]) - float(self.z_min)

    @property
    def dp(self) -> float:
        """z extent behind the centre."""
        return float(self.z_max) - float(self.pos[2])

    def flip(self) -> None:
        assert self.is_flat()

    def rotate(self, angle: float) -> None:
        assert self.rotational_symmetry, "subclasses without rotational symmetry bring their own rotate()"

    @staticmethod
    def _rotate_rc(x, y, alpha):
        """(x, y) rotated by alpha radians."""
        if not alpha:
            return x, y
        c, s = np.cos(alpha), np.sin(alpha)
        return c * x - s * y, s * x + c * y

    # ---- host evaluation used by scene set-up only -----------------------------------------------
    def _values_rel_host(self, x: np.ndarray, y: np.ndarray) -> np.ndarray:
        """Sag relative to the centre."""
        return np.zeros(np.shape(x))

    def _mask_host(self, x: np.ndarray, y: np.ndarray) -> np.ndarray:
        dx, dy = x - self.pos[0], y - self.pos[1]
        return dx ** 2 + dy ** 2 <= (self.r + self.N_EPS) ** 2

    def _values_host(self, x: np.ndarray, y: np.ndarray) -> np.ndarray:
        """Surface height in absolute coordinates; outside the surface the edge value continues (surface.py:137-164)."""
        z = np.full(np.shape(x), float(self.z_max))
        if self.is_flat():
            return z
        inside = self._mask_host(x, y)
        cx, cy, cz = self.pos
        z[inside] = cz + self._values_rel_host(x[inside] - cx, y[inside] - cy)
        if not inside.all():
            edge = self._values_rel_host(np.array([self.r - self.N_EPS]), np.array([0.]))[0]
            z[~inside] = cz + edge
        return z

    def edge(self, nc: int):
        """nc points on the surface edge (surface.py:287-304)."""
        if nc < 20:
            raise ValueError("Expected at least nc=20")
        phi = np.linspace(-0.75 * np.pi, 1.25 * np.pi, nc)
        x, y = self.r * np.cos(phi), self.r * np.sin(phi)
        return self.pos[0] + x, self.pos[1] + y, self.pos[2] + self._values_rel_host(x, y)

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
        sel = slice(None) if where is None else where
        numeric = self._kind >= _capi.SURF_ASPHERE and not self.is_flat()
        return p_hit[sel], is_hit[sel], (ill[sel] if numeric else np.array([]))

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
            val = _number(key, val, positive=True)
        elif key == "parax_roc" and val is not None:
            val = _number(key, val)
        BaseClass.__setattr__(self, key, val)


class CircularSurface(Surface):
    """Flat disc (circular_surface.py:9-44)."""

    rotational_symmetry = True
    _kind = _capi.SURF_CIRCLE

    def __init__(self, r: float, **kwargs) -> None:
        Surface.__init__(self, r, **kwargs)
        self._set_flat()
        self.lock()


class RingSurface(Surface):
    """Flat annulus r_i <= r <= r_o (ring_surface.py:10-161)."""

    rotational_symmetry = True
    _kind = _capi.SURF_RING

    def __init__(self, r: float, ri: float, **kwargs) -> None:
        Surface.__init__(self, r, **kwargs)
        self.ri = ri
        self._set_flat()
        if not self.ri < self.r:
            raise ValueError("ri needs to be smaller than r.")
        self.lock()

    @property
    def info(self) -> str:
        """(The reference's string ends in the unformatted text "{self.ri:.5g} mm", ring_surface.py:52; the value here.)"""
        return f"{Surface.info.fget(self)}, ri = {self.ri:.5g} mm"

    def _mask_host(self, x, y):
        rr = (x - self.pos[0]) ** 2 + (y - self.pos[1]) ** 2
        return (rr >= (self.ri - self.N_EPS) ** 2) & Surface._mask_host(self, x, y)

    def _desc(self):
        d = Surface._desc(self)
        d.ri = float(self.ri)
        return d

    def hurb_props(self, x: np.ndarray, y: np.ndarray):
        """Edge distances / axis for HURB (ring_surface.py:88-121), on the GPU."""
        from .. import ops
        return ops.surface_hurb_props(self._desc(), x, y)

    def __setattr__(self, key, val):
        if key == "ri":
            val = _number(key, val, positive=True)
        Surface.__setattr__(self, key, val)


class RectangularSurface(Surface):
    """Flat rectangle, optionally rotated about z (rectangular_surface.py:10-177)."""

    rotational_symmetry = False
    _kind = _capi.SURF_RECT

    def __init__(self, dim, **kwargs) -> None:
        Surface.__init__(self, 1, **kwargs)
        self._angle = 0.0  # rotation about z in radians
        self.dim = _side_lengths(dim)
        self._set_flat()
        self.lock()

    @property
    def info(self) -> str:
        x, y, z = self.pos
        return (f"{type(self).__name__}, pos = [{x:.5g} mm, {y:.5g} mm, {z:.5g} mm], "
                f"dim = [{self.dim[0]:.5g} mm, {self.dim[1]:.5g} mm]")

    @property
    def extent(self) -> tuple:
        """Box around the rotated rectangle."""
        c, s = abs(np.cos(self._angle)), abs(np.sin(self._angle))
        hx = 0.5 * (c * self.dim[0] + s * self.dim[1])
        hy = 0.5 * (s * self.dim[0] + c * self.dim[1])
        x, y = self.pos[:2]
        return x - hx, x + hx, y - hy, y + hy, self.z_min, self.z_max

    @property
    def _extent(self) -> tuple:
        """The unrotated rectangle relative to its centre."""
        hx, hy = self.dim / 2
        return -hx, hx, -hy, hy, 0., 0.

    def rotate(self, angle: float) -> None:
        with self._edit():
            self._angle = self._angle + np.deg2rad(angle)

    def flip(self) -> None:
        with self._edit():
            self._angle = -self._angle

    def _in_rect(self, x, y, half, eps):
        """Inside the centred rectangle with half sides `half`, grown by eps, in the surface's own orientation."""
        u, v = self._rotate_rc(x - self.pos[0], y - self.pos[1], -self._angle)
        return (np.abs(u) <= half[0] + eps) & (np.abs(v) <= half[1] + eps)

    def _mask_host(self, x, y):
        return self._in_rect(x, y, self.dim / 2, self.N_EPS)

    def _desc(self):
        d = Surface._desc(self)
        d.dim[:] = [float(v) for v in self.dim]
        d.angle = float(self._angle)
        return d

    def __setattr__(self, key, val):
        if key == "dim":
            _pair(key, val, "dim")
        Surface.__setattr__(self, key, val)


class SlitSurface(RectangularSurface):
    """Rectangle with a rectangular opening (slit_surface.py:12-123)."""

    _kind = _capi.SURF_SLIT

    def __init__(self, dim, dimi, **kwargs) -> None:
        RectangularSurface.__init__(self, dim, **kwargs)
        self._lock = self._new_lock = False
        self.dimi = _side_lengths(dimi)
        self.lock()

    def _mask_host(self, x, y):
        opening = self._in_rect(x, y, self.dimi / 2, -self.N_EPS)
        return RectangularSurface._mask_host(self, x, y) & ~opening

    def _desc(self):
        d = RectangularSurface._desc(self)
        d.dimi[:] = [float(v) for v in self.dimi]
        return d

    def hurb_props(self, x: np.ndarray, y: np.ndarray):
        """Edge distances / axis for HURB (slit_surface.py:65-87), on the GPU."""
        from .. import ops
        return ops.surface_hurb_props(self._desc(), x, y)

    def __setattr__(self, key, val):
        if key == "dimi":
            check_type(key, val, np.ndarray)
            if val.shape == (2,) and np.any(val >= self.dim):
                raise ValueError("Dimensions dimi must be smaller than dimension dim.")
            _pair(key, val, "dimi")
        RectangularSurface.__setattr__(self, key, val)


def _curvature_radius(key: str, val) -> float:
    val = _number(key, val)
    if key == "R" and (val == 0 or not np.isfinite(val)):
        raise ValueError("R needs to be non-zero and finite. Use planar surface types for planar surfaces.")
    return val


def _conic_sag(r2, R: float, k: float):
    """z(r) = rho r^2 / (1 + sqrt(1 - (k + 1) rho^2 r^2)) with rho = 1 / R (conic_surface.py:57)."""
    rho = 1 / R
    return rho * r2 / (1 + np.sqrt(1 - (k + 1) * rho ** 2 * r2))


class ConicSurface(Surface):
    """Conic section of revolution with vertex radius R and conic constant k (conic_surface.py:10-229)."""

    rotational_symmetry = True
    _kind = _capi.SURF_CONIC

    def __init__(self, r: float, R: float, k: float, **kwargs) -> None:
        Surface.__init__(self, r, **kwargs)
        self.R, self.k = R, k
        self.parax_roc = R
        if (self.k + 1) * self.r ** 2 >= self.R ** 2:
            raise ValueError("Surface radius r larger than radius of conic section.")
        rim = self.pos[2] + float(_conic_sag(self.r ** 2, self.R, self.k))  # monotonic in r: the rim is the extreme
        self.z_min, self.z_max = sorted((float(self.pos[2]), rim))
        self.lock()

    @property
    def info(self) -> str:
        return Surface.info.fget(self) + f", R = {self.R:.5g} mm, k = {self.k:.5g}"

    def _values_rel_host(self, x, y):
        return _conic_sag(x ** 2 + y ** 2, self.R, self.k)

    def flip(self) -> None:
        with self._edit():
            self.R = -self.R
            self.parax_roc = -self.parax_roc
            self._mirror_z_range()

    def _desc(self):
        d = Surface._desc(self)
        d.R, d.k = float(self.R), float(self.k)
        return d

    def __setattr__(self, key, val):
        if key in ("R", "k"):
            val = _curvature_radius(key, val)
        Surface.__setattr__(self, key, val)


class SphericalSurface(ConicSurface):
    """Sphere cap = conic with k = 0 (spherical_surface.py:7-98)."""

    sphere_projection_methods = ["Equidistant", "Orthographic", "Equal-Area", "Stereographic"]

    def __init__(self, r: float, R: float, **kwargs) -> None:
        ConicSurface.__init__(self, r, R, 0, **kwargs)

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
        Surface.__init__(self, r, **kwargs)
        self.R, self.k, self.coeff = R, k, coeff
        self.parax_roc = 1 / (1 / self.R + 2 * self.coeff[0])
        # the z range comes from 10 000 samples of the profile (function_surface_2d.py:84-88), hence the warning
        radii = np.linspace(0, self.r, 10000)
        sag = self._values_rel_host(radii, 0 * radii)[self._mask_host(radii, 0 * radii)]
        self.z_min, self.z_max = float(sag.min()), float(sag.max())
        warning(f"Estimated z-bounds of {type(self).__name__}: [{self.z_min:.9g}, {self.z_max:.9g}], "
                "provide actual values for higher precision.")
        self.lock()

    @property
    def info(self) -> str:
        return Surface.info.fget(self) + f", R = {self.R:.5g} mm, k = {self.k:.5g}\ncoeff = {self.coeff}"

    @property
    def _np_coeff(self) -> np.ndarray:
        """Coefficients for numpy.polyval: highest power first, odd powers and the constant zero."""
        full = np.zeros(2 * len(self.coeff) + 1)
        full[2::2] = self.coeff
        return full[::-1].copy()

    def _values_rel_host(self, x, y):
        r = np.sqrt(x ** 2 + y ** 2)  # (the reference squares the root again: kept, the z range feeds the hit search)
        return _conic_sag(r ** 2, self.R, self.k) + np.polyval(self._np_coeff, r)

    def flip(self) -> None:
        with self._edit():
            self.R = -self.R
            self.coeff = -self.coeff
            self.parax_roc = -self.parax_roc
            self._mirror_z_range()

    def _desc(self):
        d = Surface._desc(self)
        d.R, d.k = float(self.R), float(self.k)
        if len(self.coeff) > _capi.OT_MAX_ASPH:
            raise _capi.BackendError(f"AsphericSurface with more than {_capi.OT_MAX_ASPH} coefficients "
                                     "is not supported by the device kernels.")
        d.ncoeff = len(self.coeff)
        d.coeff[:len(self.coeff)] = [float(c) for c in self.coeff]
        return d

    def __setattr__(self, key, val):
        if key in ("R", "k"):
            val = _curvature_radius(key, val)
        elif key == "coeff":
            check_type(key, val, (list, np.ndarray))
            val = np.array(np.asarray_chkfinite(val, dtype=np.float64))
            if val.size == 0:
                raise ValueError("Empty coeff list. Provide coefficients or use ConicSurface instead.")
        Surface.__setattr__(self, key, val)


class Point(_Shape):
    """Point source shape (point.py:7-70)."""

    def __init__(self, **kwargs) -> None:
        _Shape.__init__(self, **kwargs)
        self.lock()

    @property
    def extent(self) -> tuple:
        x, y, z = self.pos
        return x, x, y, y, z, z


class Line(_Shape):
    """Line source shape in the xy plane (line.py:9-112)."""

    def __init__(self, r: float, angle: float = 0, **kwargs) -> None:
        _Shape.__init__(self, **kwargs)
        self._lock = False
        self.r, self.angle = r, angle
        self.lock()

    def flip(self) -> None:
        with self._edit():
            self.angle = -self.angle

    def rotate(self, angle: float) -> None:
        with self._edit():
            self.angle = self.angle + angle

    @property
    def extent(self) -> tuple:
        x, y, z = self.pos
        hx, hy = self.r * np.cos(np.deg2rad(self.angle)), self.r * np.sin(np.deg2rad(self.angle))
        return x - hx, x + hx, y - hy, y + hy, z, z

    def __setattr__(self, key, val):
        if key in ("r", "angle"):
            val = _number(key, val, positive=(key == "r"))
        BaseClass.__setattr__(self, key, val)
