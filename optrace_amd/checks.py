"""Geometry validation of a tracing scene: what `Raytracer.trace` checks before it launches anything.

Contract (optrace/tracer/raytracer.py:510-664): sources exist; every lens / filter / aperture and every source lies
inside the outline box; consecutive surfaces of the sequential path do not run through each other; with HURB, only
ring and slit apertures can bend rays.  Findings are reported as one warning plus `geometry_error`, never raised.

The work is per scene, not per ray, and stays on the host: surfaces are compared on a res x res grid of their common
footprint through the NumPy sag functions the scene flattener uses anyway (`Surface._values_host / _mask_host`).
"""
from __future__ import annotations

import numpy as np

from .geometry.elements import Aperture
from .geometry.surfaces import Surface, Point, Line, RingSurface, SlitSurface

_NONE = np.array([])


def _is_curve(obj) -> bool:
    return isinstance(obj, (Point, Line))


def _height(obj, x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """z of the object above (x, y): points and lines are flat at their own z."""
    return np.full(x.shape, float(obj.pos[2])) if _is_curve(obj) else obj._values_host(x, y)


def _defined(obj, x: np.ndarray, y: np.ndarray) -> np.ndarray:
    return np.ones(x.shape, dtype=bool) if _is_curve(obj) else obj._mask_host(x, y)


def _samples(first, second, res: int):
    """x, y positions where the two objects are compared, or None if their footprints do not meet."""
    for obj in (first, second):
        if isinstance(obj, Point):
            return np.array([obj.pos[0]]), np.array([obj.pos[1]])
    for obj in (first, second):
        if isinstance(obj, Line):
            t = np.linspace(-obj.r, obj.r, 10 * res)
            # the reference feeds the angle, which is kept in degrees, to cos / sin as it is (raytracer.py:619-620): the
            # sampled segment points elsewhere than the line does.  Kept, so that geometry checks give the same verdicts
            # (tests/golden/host_objects.npz, collision/line_back)
            phi = obj.angle
            return obj.pos[0] + t * np.cos(phi), obj.pos[1] + t * np.sin(phi)
    fa, fb = np.asarray(first.extent[:4]), np.asarray(second.extent[:4])
    x0, x1 = max(fa[0], fb[0]), min(fa[1], fb[1])
    y0, y1 = max(fa[2], fb[2]), min(fa[3], fb[3])
    if x0 > x1 or y0 > y1:
        return None
    gx, gy = np.meshgrid(np.linspace(x0, x1, res), np.linspace(y0, y1, res))
    return gx.ravel(), gy.ravel()


def collision_points(first, second, res: int = 100):
    """Where does `first`, which comes earlier along the optical path, lie behind `second`?

    -> (found, x, y, z) with the sample positions of the violations (z: the height of `first` there, or of the
    surface when `first` is a point or a line).  Only positions where both objects are defined count."""
    if not (isinstance(first, Surface) or isinstance(second, Surface)):
        raise TypeError("At least one object needs to be a Surface for collision detection")
    if not (_is_curve(first) or _is_curve(second)) and first.extent[5] < second.extent[4]:
        return False, _NONE, _NONE, _NONE  # z ranges apart: nothing to sample
    xy = _samples(first, second, res)
    if xy is None:
        return False, _NONE, _NONE, _NONE
    x, y = xy
    both = _defined(first, x, y) & _defined(second, x, y)
    x, y = x[both], y[both]
    z1, z2 = _height(first, x, y), _height(second, x, y)
    bad = np.flatnonzero(z1 > z2)
    z_rep = z2 if _is_curve(first) else z1
    return bool(bad.size), x[bad], y[bad], z_rep[bad]


def _inside(box, outline, eps: float) -> bool:
    lo_ok = all(outline[k] - eps <= box[k] for k in (0, 2, 4))
    hi_ok = all(box[k] <= outline[k] + eps for k in (1, 3, 5))
    return lo_ok and hi_ok


def find_geometry_error(rt, elements: list):
    """First problem of the scene in the reference's order of inspection, or None.
    -> None | (warning text, fault positions (n, 3) or None)"""
    if not rt.ray_sources:
        return "RaySource Missing.", None
    outline, eps = rt.outline, rt.N_EPS
    last = len(elements) - 1
    for i, el in enumerate(elements):
        if not _inside(el.extent, outline, eps):
            return f"Element{i} {el} with extent {el.extent} outside outline {outline}.", None
        # surfaces that follow each other on the sequential path: front -> next front, front -> back -> next front
        pairs = [(el.front, elements[i + 1].front)] if i < last else []
        if el.has_back():
            pairs.append((el.front, el.back))
            if i < last:
                pairs.append((el.back, elements[i + 1].front))
        hit = _first_collision(pairs)
        if rt.use_hurb and i < last and isinstance(el, Aperture) and not isinstance(el.front, (RingSurface, SlitSurface)):
            return f"Ray bending for surface type {type(el.front).__name__} not implemented.", None
        if hit is not None:
            return hit
    for rs in rt.ray_sources:
        if not _inside(rs.extent, outline, eps):
            return f"RaySource {rs} with extent {rs.extent} outside outline {outline}.", None
        if rs.pos[2] >= elements[0].extent[4]:  # only sources that reach into the first element's z range
            hit = _first_collision([(rs.surface, elements[0].front)])
            if hit is not None:
                return hit
    return None


def _first_collision(pairs: list):
    for a, b in pairs:
        found, x, y, z = collision_points(a, b)
        if found:
            text = (f"Detected collision between two Surfaces at {x[0], y[0], z[0]}"
                    f" and at least {x.shape[0]} other positions.")
            return text, np.column_stack((x, y, z))
    return None
