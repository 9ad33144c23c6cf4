"""Common parent of all geometry / spectrum classes (mirrors optrace/tracer/base_class.py:11-114)."""
from __future__ import annotations

import copy
import zlib
from typing import Any

import numpy as np


def check_type(key: str, val: Any, types) -> None:
    """TypeError in the reference's wording (property_checker.py:2-14) if `val` has the wrong type."""
    if isinstance(val, bool) and types in (float, int, (int, float), (float, int)):
        raise TypeError(f"Property '{key}' needs to be of type(s) {types}, but is bool.")
    if not isinstance(val, types):
        raise TypeError(f"Property '{key}' needs to be of type(s) {types}, but is {type(val).__name__}.")


def check_above(key: str, val, bound) -> None:
    if val <= bound:
        raise ValueError(f"Property '{key}' needs to be above {bound}, but is {val}.")


def check_not_below(key: str, val, bound) -> None:
    if val < bound:
        raise ValueError(f"Property '{key}' needs to be at least {bound}, but is {val}.")


def check_not_above(key: str, val, bound) -> None:
    if val > bound:
        raise ValueError(f"Property '{key}' needs to be at most {bound}, but is {val}.")


def check_in(key: str, val, choices) -> None:
    if val not in choices:
        raise ValueError(f"Property '{key}' needs to be one of {choices}, but is '{val}'.")


_EPOCH = [0]
_SAW_WRITEABLE = [False]  # set by crepr when it meets a large array that can still be edited in place
_WATCH = [None]           # [list or None]: while a list, crepr adds the small arrays it meets that can still be edited in
                          # place (RaySource.s, conv_pos, the outline); Raytracer.trace collects them for its shortcut


def mutation_epoch() -> int:
    """Counts attribute assignments on tracked objects (everything a trace depends on is a BaseClass whose arrays
    are read-only and whose state changes only through `__setattr__`).  While it stands still, nothing a
    Raytracer compiled or checked can have changed, so `Raytracer.trace` may skip its snapshot comparison."""
    return _EPOCH[0]


def touch() -> None:
    """Record a state change that does not pass through `__setattr__` (list edits of a Group)."""
    _EPOCH[0] += 1


def _array_token(a: np.ndarray) -> tuple:
    """Identity of a large array for change detection.  The reference keeps `id(array)` (base_class.py:40-48); an id
    alone is reused once the old array is freed and says nothing about in-place edits, so the token also carries the
    shape and a content checksum: of everything while the array is writeable, of 64 evenly spread elements once it is
    locked read-only (its content cannot change any more, the samples only tell a recycled id apart)."""
    if a.flags.writeable:
        _SAW_WRITEABLE[0] = True
    flat = a.reshape(-1) if a.flags.c_contiguous else a.ravel()
    part = flat if a.flags.writeable else flat[::max(1, flat.shape[0] // 64)]
    return id(a), a.shape, zlib.crc32(np.ascontiguousarray(part).view(np.uint8)) if part.dtype != object else 0


class BaseClass:
    """Description strings, copy, read-only locking and a compact state representation."""

    _tracked = True  #: False for result containers (ray storage, images): their attributes never feed a trace

    def __init__(self, desc: str = "", long_desc: str = "") -> None:
        self._lock = False
        self._new_lock = False
        self.desc = desc
        self.long_desc = long_desc

    def crepr(self) -> list:
        """State as nested lists / tuples of plain values; compared to detect changes since the last trace
        (base_class.py:27-58).  Small arrays go in by value, large ones as a token (`_array_token`), callables by id."""
        def plain(v):
            if isinstance(v, BaseClass):
                return v.crepr()
            if isinstance(v, np.ndarray):
                if v.size >= 20:
                    return _array_token(v)
                if v.flags.writeable and _WATCH[0] is not None:  # the reference re-reads such arrays at every trace
                    _WATCH[0].append(v)
                return tuple(v.ravel().tolist())
            if isinstance(v, list):
                return tuple(v)
            return id(v) if callable(v) else v
        return [plain(v) for v in vars(self).values()]

    def get_desc(self, fallback: str = "") -> str:
        """The short description, or `fallback` if there is none."""
        return self.desc or fallback

    def get_long_desc(self, fallback: str = "") -> str:
        """The long description, else the short one, else `fallback`."""
        return self.long_desc or self.get_desc(fallback)

    def copy(self):
        """A fully independent copy."""
        return copy.deepcopy(self)

    def lock(self) -> None:
        """Make the object (and its arrays) read-only."""
        for arr in vars(self).values():
            if isinstance(arr, np.ndarray):
                arr.setflags(write=False)
        self._lock = self._new_lock = True

    def __str__(self) -> str:
        d = {k: v for k, v in self.__dict__.items() if not k.startswith("_")}
        return f"{type(self).__name__} at {hex(id(self))} with {d}"

    def __setattr__(self, key: str, val: Any) -> None:
        if key not in ("_lock", "_new_lock"):
            d = self.__dict__
            if d.get("_new_lock", False) and not (key in d or hasattr(type(self), key)):
                raise AttributeError(f"Failed to set invalid/unknown property {key}.")
            if d.get("_lock", False) and key not in ("desc", "long_desc"):
                raise RuntimeError("Object is currently read-only. Create a new object with new properties "
                                   "or use class methods to change its properties.")
        if key in ("desc", "long_desc"):
            check_type(key, val, str)
        self.__dict__[key] = val
        if self._tracked:
            _EPOCH[0] += 1
