"""Scene elements: Element, Lens, IdealLens, Aperture, Filter, Detector and Group.

Host-side counterpart of optrace/tracer/geometry/{element,lens,ideal_lens,aperture,filter,detector,group}.py: the
same constructors, attributes and error texts, written around what this package needs from them --
elements validate geometry and hand `optrace_amd.scene.CompiledScene` the surfaces to flatten into descriptor tables.

An element owns private copies of its surfaces and keeps them consistent: the reference position of a two-surface
element lies d1 behind the front vertex and d2 before the back vertex, so every move, flip or surface exchange goes
through the element (direct assignment of front / back / d1 / d2 / pos is refused once it is built).
"""
from __future__ import annotations

import contextlib
from typing import Any

import numpy as np

from ..base import BaseClass, check_type, touch
from ..refraction_index import RefractionIndex
from ..spectrum import TransmissionSpectrum
from .._warn import warning
from .surfaces import Surface, Point, Line, CircularSurface, AsphericSurface

_GUARDED = {"d1": "surface", "d2": "surface", "front": "surface", "surface": "surface", "back": "surface", "pos": "pos"}
_GUARD_TEXT = {"surface": "Use Functions set_surface to reassign a new Surface or its thickness.",
               "pos": "Use move_to(pos) to move the object"}


def _position(pos, text: str) -> np.ndarray:
    """A finite [x, y, z] as float64 array, or the errors the reference raises for it."""
    check_type("pos", pos, (list, np.ndarray))
    arr = np.asarray_chkfinite(pos, dtype=np.float64)
    if arr.shape[0] != 3:
        raise ValueError(text)
    return arr


class Element(BaseClass):
    """Object with a front and an optional back surface (element.py:38-237)."""

    abbr = "EL"
    _allow_non_2D = True

    def __init__(self, front, pos, back=None, d1: float = None, d2: float = None, **kwargs) -> None:
        self._sealed = False  # True: geometry attributes change through methods only
        self.front, self.back = front, back
        self.d1, self.d2 = d1, d2
        if back is not None:
            if None in (d1, d2):
                raise ValueError("d1 and d2 need to be specified for a Element with a back surface")
            if min(d1, d2) < 0:
                raise ValueError(f"Thicknesses d1, d2 need to be non-negative but are {d1=} and {d2=}.")
        self.move_to(pos)
        BaseClass.__init__(self, **kwargs)
        self._sealed = True

    @contextlib.contextmanager
    def _open(self):
        """Geometry attributes may be assigned inside this block."""
        self._sealed = False
        try:
            yield
        finally:
            self._sealed = True

    def has_back(self) -> bool:
        return self.back is not None

    def set_surface(self, surf: Surface) -> None:
        if self.back is not None:
            raise RuntimeError("Replacing of Surfaces only supported for objects with one surface")
        where = self.front.pos
        with self._open():
            self.front = surf  # (copied on assignment)
            self.front.move_to(where)

    def move_to(self, pos) -> None:
        target = _position(pos, "pos needs to have 3 elements.")
        if self.back is None:
            self.front.move_to(target)
            return
        dz = np.array([0., 0., 1.])
        self.front.move_to(target - self.d1 * dz)
        self.back.move_to(target + self.d2 * dz)

    @property
    def surface(self):
        return self.front

    @property
    def pos(self) -> np.ndarray:
        p = np.array(self.front.pos, dtype=np.float64)
        if self.back is not None:
            p[2] += self.d1
        return p

    @property
    def extent(self) -> tuple:
        if self.back is None:
            return self.front.extent
        f, b = np.asarray(self.front.extent), np.asarray(self.back.extent)
        lo, hi = np.minimum(f, b), np.maximum(f, b)
        return tuple(np.where([True, False] * 3, lo, hi))

    def get_desc(self, fallback: str = None) -> str:
        kinds = type(self.front).__name__ + (f" + {type(self.back).__name__}" if self.back is not None else "")
        return BaseClass.get_desc(self, f"{kinds}, z = {self.pos[2]:.04g}")

    def flip(self) -> None:
        """Turn the element around an axis parallel to x through its position."""
        if self.back is None:
            self.front.flip()
            return
        z = self.pos[2]
        with self._open():
            for s in (self.front, self.back):
                s.flip()
            # the old back surface becomes the front and vice versa, vertex distances swap with them
            self.front.move_to([self.front.pos[0], self.front.pos[1], z + self.d1])
            self.back.move_to([self.back.pos[0], self.back.pos[1], z - self.d2])
            old_front, old_d1 = self.front, self.d1
            BaseClass.__setattr__(self, "front", self.back)  # (both are this element's own copies already)
            BaseClass.__setattr__(self, "back", old_front)
            self.d1, self.d2 = self.d2, old_d1

    def rotate(self, angle: float) -> None:
        for s in (self.front, self.back):
            if s is not None:
                s.rotate(angle)

    def __setattr__(self, key: str, val: Any) -> None:
        if key in _GUARDED and self.__dict__.get("_sealed", False):
            raise RuntimeError(_GUARD_TEXT[_GUARDED[key]])
        if key in ("front", "back") and val is not None:
            check_type(key, val, (Surface, Point, Line) if self._allow_non_2D else Surface)
            val = val.copy()  # elements own private copies of their surfaces (element.py:223-231)
        elif key in ("d1", "d2") and val is not None:
            check_type(key, val, (int, float))
            val = float(val)
        BaseClass.__setattr__(self, key, val)


def _lens_thicknesses(front, back, de, d, d1, d2):
    """(d1, d2) of a lens from whichever thickness the caller gave (lens.py:58-86): the centre thickness d, the gap
    de between the z ranges of the surfaces (default 0), or d1 and d2 themselves.  A centre thickness (or a negative
    gap) that makes the z ranges overlap is split evenly."""
    d1 = None if d1 is None else float(d1)
    d2 = None if d2 is None else float(d2)
    if not (isinstance(front, Surface) and isinstance(back, Surface)):
        return d1, d2  # the type error comes from the assignment in Element
    if d is not None:
        de = d - front.dp - back.dn
        if de < 0:
            return d / 2, d / 2
    if de is not None and d1 is None and d2 is None:
        if de < 0:
            return -de / 2, -de / 2
        return front.dp + de / 2., back.dn + de / 2.
    if None in (d1, d2):
        raise ValueError("Both thicknesses d1, d2 need to be specified")
    return d1, d2


class Lens(Element):
    """Two refracting surfaces with a material index n and an optional index n2 behind (lens.py:12-126)."""

    abbr = "L"
    _allow_non_2D = False
    is_ideal = False

    def __init__(self, front: Surface, back: Surface, n: RefractionIndex, pos, de: float = 0,
                 d: float = None, d1: float = None, d2: float = None, n2: RefractionIndex = None,
                 **kwargs) -> None:
        self.n, self.n2 = n, n2
        d1, d2 = _lens_thicknesses(front, back, de, d, d1, d2)
        Element.__init__(self, front, pos, back, d1, d2, **kwargs)
        self._new_lock = True

    @property
    def d(self) -> float:
        """Thickness at the optical axis."""
        return self.d1 + self.d2

    @property
    def de(self) -> float:
        """Gap between the z ranges of the two surfaces (negative if they overlap)."""
        return float(self.back.z_min) - float(self.front.z_max)

    def __setattr__(self, key, val):
        if key == "n":
            check_type(key, val, RefractionIndex)
        elif key == "n2":
            check_type(key, val, (RefractionIndex, type(None)))
        Element.__setattr__(self, key, val)


class IdealLens(Lens):
    """Aberration-free thin lens of optical power D on a disc (ideal_lens.py:11-43)."""

    is_ideal = True

    def __init__(self, r: float, D: float, pos, n2: RefractionIndex = None, **kwargs) -> None:
        check_type("D", D, (int, float))
        np.asarray_chkfinite(D)
        if D == 0:
            raise ValueError("Optical Power needs to be non-zero")
        self.D = float(D)
        disc = CircularSurface(r=r)
        Lens.__init__(self, disc, disc, RefractionIndex("Constant", n=1), pos, d=0, n2=n2, **kwargs)


class _SingleSurface(Element):
    """Elements made of one 2-D surface."""

    _allow_non_2D = False

    def __init__(self, surface: Surface, pos, **kwargs) -> None:
        Element.__init__(self, surface, pos, **kwargs)
        self._new_lock = True


class Aperture(_SingleSurface):
    """Absorbing surface (aperture.py:8-26)."""

    abbr = "AP"


class Filter(_SingleSurface):
    """Surface with a transmission spectrum (filter.py:11-63)."""

    abbr = "F"

    def __init__(self, surface: Surface, pos, spectrum: TransmissionSpectrum, **kwargs) -> None:
        self.spectrum = spectrum
        _SingleSurface.__init__(self, surface, pos, **kwargs)

    def __call__(self, wl: np.ndarray) -> np.ndarray:
        """Transmission at the wavelengths wl."""
        return self.spectrum(wl)

    def __setattr__(self, key, val):
        if key == "spectrum":
            check_type(key, val, TransmissionSpectrum)
        Element.__setattr__(self, key, val)


class Detector(_SingleSurface):
    """Surface on which images are rendered (detector.py:11-43)."""

    abbr = "DET"

    def __setattr__(self, key, val):
        if key == "front":
            from .data_surfaces import DataSurface2D   # (data_surfaces imports surfaces: resolved at call time)
            # detector.py:39-41: data and function surfaces and their subclasses (the asphere is one in the reference)
            if isinstance(val, (DataSurface2D, AsphericSurface)):
                raise RuntimeError("Classes and subclasses of DataSurface1D, DataSurface2D, FunctionSurface2D"
                                   " are not supported as Detector surfaces.")
        Element.__setattr__(self, key, val)


class Group(BaseClass):
    """Container of elements with an ambient index n0 (group.py:18-308)."""

    _LISTS = ("lenses", "apertures", "filters", "ray_sources", "detectors", "markers", "volumes")

    def __init__(self, elements: list = None, n0: RefractionIndex = None, **kwargs) -> None:
        for name in self._LISTS:
            BaseClass.__setattr__(self, name, [])
        self.n0 = n0
        BaseClass.__init__(self, **kwargs)
        if elements is not None:
            self.add(elements)

    def __setattr__(self, key, val):
        if key == "n0":
            val = RefractionIndex("Constant", n=1) if val is None else val
            check_type(key, val, RefractionIndex)
        BaseClass.__setattr__(self, key, val)

    @property
    def _elements(self) -> list:
        return [el for name in self._LISTS for el in getattr(self, name)]

    @property
    def elements(self) -> list:
        """All elements sorted by z position (group.py:67-70)."""
        return sorted(self._elements, key=lambda el: el.pos[2])

    @property
    def pos(self):
        """Position of the first element along z; the origin for an empty group."""
        els = self.elements
        return els[0].pos if els else [0, 0, 0]

    @property
    def tracing_surfaces(self) -> list:
        """Front/back surfaces of lenses, filters and apertures in z order (group.py:84-98); an ideal lens counts
        once."""
        out = []
        for el in self.elements:
            if not isinstance(el, (Lens, Filter, Aperture)):
                continue
            out.append(el.front)
            if el.back is not None and not getattr(el, "is_ideal", False):
                out.append(el.back)
        return out

    @property
    def extent(self) -> tuple:
        boxes = np.array([el.extent for el in self._elements], dtype=np.float64).reshape(-1, 6)
        if not boxes.shape[0]:
            return 0, 0, 0, 0, 0, 0
        lo, hi = boxes.min(axis=0), boxes.max(axis=0)
        return lo[0], hi[1], lo[2], hi[3], lo[4], hi[5]

    def move_to(self, pos) -> None:
        """Shift all elements so that the group's position becomes `pos`."""
        shift = _position(pos, "pos needs to have exactly 3 elements.") - self.pos
        for el in self._elements:
            el.move_to(el.pos + shift)

    def rotate(self, angle: float, x0: float = 0, y0: float = 0) -> None:
        """Rotate the group by `angle` degrees around the axis through (x0, y0) parallel to z."""
        c, s = np.cos(np.deg2rad(angle)), np.sin(np.deg2rad(angle))
        for el in self.elements:
            x, y, z = el.pos
            el.rotate(angle)
            el.move_to([x0 + c * (x - x0) - s * (y - y0), y0 + s * (x - x0) + c * (y - y0), z])

    def flip(self, y0: float = 0, z0: float = None) -> None:
        """Turn the group around the axis through (y0, z0) parallel to x (z0: middle of the z extent).  The media
        between the lenses reverse their order with the elements."""
        order = self.elements
        if not order:
            return
        media = [self.n0] + [el.n2 for el in order if isinstance(el, Lens)]
        zc = 0.5 * (self.extent[4] + self.extent[5]) if z0 is None else z0
        self.clear()
        self.add(order[::-1])
        for el in order:
            x, y, z = el.pos
            el.flip()
            el.move_to([x, 2 * y0 - y, 2 * zc - z])
        media = [self.n0 if n is None else n for n in media[::-1]]
        self.n0 = media[0]
        for lens, n2 in zip(self.lenses, media[1:]):
            lens.n2 = n2

    def _list_for(self, el):
        from .ray_source import RaySource
        for cls, name in ((Aperture, "apertures"), (Filter, "filters"), (RaySource, "ray_sources"),
                          (Detector, "detectors"), (Lens, "lenses")):
            if isinstance(el, cls):
                return getattr(self, name)
        return None

    def add(self, el) -> None:
        """Add an element, a list of elements or the elements of another Group (whose ambient index is taken over)."""
        touch()
        if isinstance(el, list):
            for item in el:
                self.add(item)
            return
        if self.has(el):
            warning("Element already included in geometry. Make a copy to include it another time.")
            return
        if isinstance(el, Group):
            if el.n0 != self.n0:
                warning("Overwriting ambient index with index from new Group.")
                self.n0 = el.n0
            self.add(el.elements)
            return
        target = self._list_for(el)
        if target is None:
            raise TypeError(f"Unsupported element type {type(el).__name__}.")
        target.append(el)

    def remove(self, el) -> bool:
        """Remove an element (by identity), a list of elements or a Group's elements; True if anything was removed."""
        touch()
        if isinstance(el, (list, Group)):
            items = list(el) if isinstance(el, list) else el._elements
            return any([self.remove(item) for item in items])
        found = False
        for name in self._LISTS:
            lst = getattr(self, name)
            kept = [x for x in lst if x is not el]
            found = found or len(kept) != len(lst)
            lst[:] = kept
        return found

    def has(self, el) -> bool:
        return any(x is el for x in self._elements)

    def clear(self) -> None:
        touch()
        for name in self._LISTS:
            getattr(self, name)[:] = []
