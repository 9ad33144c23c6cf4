"""Scene elements: Element, Lens, IdealLens, Aperture, Filter, Detector and Group.

Host-side mirror of optrace/tracer/geometry/{element,lens,ideal_lens,aperture,filter,detector,group}.py.
These objects only hold and validate geometry; `optrace_amd.scene.compile_scene` flattens them into
the plain descriptor tables the device kernels read.
"""
from __future__ import annotations

from typing import Any

import numpy as np

from ..base import BaseClass, check_type, touch
from ..refraction_index import RefractionIndex
from ..spectrum import TransmissionSpectrum
from .._warn import warning
from .surfaces import Surface, Point, Line, CircularSurface, AsphericSurface


class Element(BaseClass):
    """Object with a front and an optional back surface (element.py:38-237)."""

    abbr = "EL"
    _allow_non_2D = True

    def __init__(self, front, pos, back=None, d1: float = None, d2: float = None, **kwargs) -> None:
        self._geometry_lock = False
        self.front = front
        self.back = back
        self.d1 = d1
        self.d2 = d2
        if self.has_back():
            if d1 is None or d2 is None:
                raise ValueError("d1 and d2 need to be specified for a Element with a back surface")
            if d1 < 0 or d2 < 0:
                raise ValueError(f"Thicknesses d1, d2 need to be non-negative but are {d1=} and {d2=}.")
        self.move_to(pos)
        super().__init__(**kwargs)
        self._geometry_lock = True

    def has_back(self) -> bool:
        return self.back is not None

    def set_surface(self, surf: Surface) -> None:
        if self.has_back():
            raise RuntimeError("Replacing of Surfaces only supported for objects with one surface")
        self._geometry_lock = False
        pos = self.front.pos
        self.front = surf.copy()
        self.front.move_to(pos)
        self._geometry_lock = True

    def move_to(self, pos) -> None:
        check_type("pos", pos, (list, np.ndarray))
        pos = np.asarray_chkfinite(pos, dtype=np.float64)
        if pos.shape[0] != 3:
            raise ValueError("pos needs to have 3 elements.")
        if not self.has_back():
            self.front.move_to(pos)
        else:
            self.front.move_to(pos - [0, 0, self.d1])
            self.back.move_to(pos + [0, 0, self.d2])

    @property
    def surface(self):
        return self.front

    @property
    def pos(self) -> np.ndarray:
        return self.front.pos + [0, 0, 0 if not self.has_back() else self.d1]

    @property
    def extent(self) -> tuple:
        if not self.has_back():
            return self.front.extent
        ext = np.zeros(6, dtype=np.float64)
        exts = np.column_stack((self.front.extent, self.back.extent))
        ext[[0, 2, 4]] = np.min(exts, axis=1)[[0, 2, 4]]
        ext[[1, 3, 5]] = np.max(exts, axis=1)[[1, 3, 5]]
        return tuple(ext)

    def get_desc(self, fallback: str = None) -> str:
        s1 = type(self.front).__name__
        if self.has_back():
            fallback = f"{s1} + {type(self.back).__name__}, z = {self.pos[2]:.04g}"
        else:
            fallback = f"{s1}, z = {self.pos[2]:.04g}"
        return super().get_desc(fallback)

    def flip(self) -> None:
        if self.has_back():
            self._geometry_lock = False
            self.back.flip()
            self.front.flip()
            zp = self.pos[2]
            self.front.move_to([*self.front.pos[:2], zp + self.d1])
            self.back.move_to([*self.back.pos[:2], zp - self.d2])
            self.front, self.back = self.back, self.front
            self.d1, self.d2 = self.d2, self.d1
            self._geometry_lock = True
        else:
            self.front.flip()

    def rotate(self, angle: float) -> None:
        self.front.rotate(angle)
        if self.has_back():
            self.back.rotate(angle)

    def __setattr__(self, key: str, val: Any) -> None:
        if self.__dict__.get("_geometry_lock", False):
            if key in ("d1", "d2", "front", "surface", "back"):
                raise RuntimeError("Use Functions set_surface to reassign a new Surface or its thickness.")
            if key == "pos":
                raise RuntimeError("Use move_to(pos) to move the object")
        if key == "front" or (key == "back" and val is not None):
            check_type(key, val, (Surface, Point, Line) if self._allow_non_2D else Surface)
            val = val.copy()  # elements own private copies of their surfaces (element.py:223-231)
        elif key in ("d1", "d2") and val is not None:
            check_type(key, val, (int, float))
            val = float(val)
        super().__setattr__(key, val)


class Lens(Element):
    """Two refracting surfaces with a material index n and an optional index n2 behind (lens.py:12-126)."""

    abbr = "L"
    _allow_non_2D = False
    is_ideal = False

    def __init__(self, front: Surface, back: Surface, n: RefractionIndex, pos, de: float = 0,
                 d: float = None, d1: float = None, d2: float = None, n2: RefractionIndex = None,
                 **kwargs) -> None:
        self.n = n
        self.n2 = n2
        d1 = float(d1) if d1 is not None else d1
        d2 = float(d2) if d2 is not None else d2

        if isinstance(front, Surface) and isinstance(back, Surface):
            if d is not None:
                de = d - front.dp - back.dn
                if de < 0:  # overlapping z extents: split the centre thickness evenly
                    d1 = d / 2
                    d2 = d / 2
            if de is not None and d1 is None and d2 is None:
                if de < 0:
                    d1 = -de / 2
                    d2 = -de / 2
                else:
                    d1 = de / 2. + front.dp
                    d2 = de / 2. + back.dn
            elif d1 is None or d2 is None:
                raise ValueError("Both thicknesses d1, d2 need to be specified")

        super().__init__(front, pos, back, d1, d2, **kwargs)
        self._new_lock = True

    @property
    def d(self) -> float:
        return self.d1 + self.d2

    @property
    def de(self) -> float:
        return float(self.back.z_min - self.front.z_max)

    def __setattr__(self, key, val):
        if key == "n2":
            check_type(key, val, (RefractionIndex, type(None)))
        if key == "n":
            check_type(key, val, RefractionIndex)
        super().__setattr__(key, val)


class IdealLens(Lens):
    """Aberration-free thin lens of optical power D on a disc (ideal_lens.py:11-43)."""

    is_ideal = True

    def __init__(self, r: float, D: float, pos, n2: RefractionIndex = None, **kwargs) -> None:
        check_type("D", D, (int, float))
        np.asarray_chkfinite(D)
        self.D = float(D)
        if not D:
            raise ValueError("Optical Power needs to be non-zero")
        super().__init__(front=CircularSurface(r=r), back=CircularSurface(r=r),
                         n=RefractionIndex("Constant", n=1), pos=pos, d=0, n2=n2, **kwargs)


class Aperture(Element):
    """Absorbing surface (aperture.py:8-26)."""

    abbr = "AP"
    _allow_non_2D = False

    def __init__(self, surface: Surface, pos, **kwargs) -> None:
        super().__init__(surface, pos, **kwargs)
        self._new_lock = True


class Filter(Element):
    """Surface with a transmission spectrum (filter.py:11-63)."""

    abbr = "F"
    _allow_non_2D = False

    def __init__(self, surface: Surface, pos, spectrum: TransmissionSpectrum, **kwargs) -> None:
        super().__init__(surface, pos, **kwargs)
        self.spectrum = spectrum
        self._new_lock = True

    def __call__(self, wl: np.ndarray) -> np.ndarray:
        return self.spectrum(wl)

    def __setattr__(self, key, val):
        if key == "spectrum":
            check_type(key, val, TransmissionSpectrum)
        super().__setattr__(key, val)


class Detector(Element):
    """Surface on which images are rendered (detector.py:11-43)."""

    abbr = "DET"
    _allow_non_2D = False

    def __init__(self, surface: Surface, pos, **kwargs) -> None:
        super().__init__(surface, pos, **kwargs)
        self._new_lock = True

    def __setattr__(self, key, val):
        if key == "front" and isinstance(val, AsphericSurface):
            raise RuntimeError("Function-defined surfaces are not supported as Detector surfaces.")
        super().__setattr__(key, val)


class Group(BaseClass):
    """Container of elements with an ambient index n0 (group.py:18-308)."""

    def __init__(self, elements: list = None, n0: RefractionIndex = None, **kwargs) -> None:
        self.lenses: list = []
        self.apertures: list = []
        self.filters: list = []
        self.detectors: list = []
        self.ray_sources: list = []
        self.markers: list = []
        self.volumes: list = []
        self.n0 = n0
        super().__init__(**kwargs)
        if elements is not None:
            self.add(elements)

    def __setattr__(self, key, val):
        if key == "n0":
            if val is None:
                val = RefractionIndex("Constant", n=1)
            check_type(key, val, RefractionIndex)
        super().__setattr__(key, val)

    @property
    def _elements(self) -> list:
        return [*self.lenses, *self.apertures, *self.filters, *self.ray_sources, *self.detectors,
                *self.markers, *self.volumes]

    @property
    def elements(self) -> list:
        """All elements sorted by z position (group.py:67-70)."""
        return sorted(self._elements, key=lambda el: el.pos[2])

    @property
    def pos(self):
        return self.elements[0].pos if len(self._elements) else [0, 0, 0]

    @property
    def tracing_surfaces(self) -> list:
        """Front/back surfaces of lenses, filters and apertures in z order (group.py:84-98)."""
        surfs = []
        for el in self.elements:
            if isinstance(el, (Lens, Filter, Aperture)):
                surfs.append(el.front)
                if el.has_back() and not isinstance(el, IdealLens):
                    surfs.append(el.back)
        return surfs

    @property
    def extent(self) -> tuple:
        els = self._elements
        if not len(els):
            return 0, 0, 0, 0, 0, 0
        ext = np.array([np.array(el.extent) for el in els])
        mx, mn = np.max(ext, axis=0), np.min(ext, axis=0)
        return mn[0], mx[1], mn[2], mx[3], mn[4], mx[5]

    def move_to(self, pos) -> None:
        check_type("pos", pos, (list, np.ndarray))
        pos = np.asarray_chkfinite(pos, dtype=np.float64)
        if pos.shape[0] != 3:
            raise ValueError("pos needs to have exactly 3 elements.")
        pos0 = self.pos
        for el in self._elements:
            el.move_to(el.pos - (pos0 - pos))

    def rotate(self, angle: float, x0: float = 0, y0: float = 0) -> None:
        if not len(self._elements):
            return
        ang = np.deg2rad(angle)
        for el in self.elements:
            xr, yr = el.pos[0] - x0, el.pos[1] - y0
            posr = [x0 + xr * np.cos(ang) - yr * np.sin(ang), y0 + xr * np.sin(ang) + yr * np.cos(ang), el.pos[2]]
            el.rotate(angle)
            el.move_to(posr)

    def flip(self, y0: float = 0, z0: float = None) -> None:
        if not len(self._elements):
            return
        els = self.elements
        ns = [self.n0] + [L.n2 for L in els if isinstance(L, Lens)]
        z0 = np.mean(self.extent[4:]) if z0 is None else z0
        self.clear()
        els.reverse()
        self.add(els)
        for el in els:
            el.flip()
            el.move_to([el.pos[0], y0 - (el.pos[1] - y0), z0 - (el.pos[2] - z0)])
        ns.reverse()
        ns = [n if n is not None else self.n0 for n in ns]
        self.n0 = ns[0]
        for n2, L in zip(ns[1:], self.lenses):
            L.n2 = n2

    def add(self, el) -> None:
        from .ray_source import RaySource
        touch()
        if not isinstance(el, list) and self.has(el):
            warning("Element already included in geometry. Make a copy to include it another time.")
            return
        if isinstance(el, Aperture):
            self.apertures.append(el)
        elif isinstance(el, Filter):
            self.filters.append(el)
        elif isinstance(el, RaySource):
            self.ray_sources.append(el)
        elif isinstance(el, Detector):
            self.detectors.append(el)
        elif isinstance(el, Lens):
            self.lenses.append(el)
        elif isinstance(el, Group):
            if self.n0 != el.n0:
                warning("Overwriting ambient index with index from new Group.")
                self.n0 = el.n0
            for eli in el.elements:
                self.add(eli)
        elif isinstance(el, list):
            for eli in el:
                self.add(eli)
        else:
            raise TypeError(f"Unsupported element type {type(el).__name__}.")

    def remove(self, el) -> bool:
        touch()
        success = False
        if isinstance(el, list):
            for eli in el.copy():
                success = self.remove(eli) or success
        elif isinstance(el, Group):
            for eli in el._elements.copy():
                success = self.remove(eli) or success
        else:
            for lst in (self.lenses, self.apertures, self.detectors, self.volumes, self.filters,
                        self.ray_sources, self.markers):
                for lel in lst.copy():
                    if lel is el:
                        lst.remove(lel)
                        success = True
        return success

    def has(self, el) -> bool:
        return any(eli is el for eli in self._elements)

    def clear(self) -> None:
        touch()
        for lst in (self.lenses, self.apertures, self.filters, self.detectors, self.ray_sources,
                    self.markers, self.volumes):
            lst[:] = []
