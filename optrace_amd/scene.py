"""Flattening of a Raytracer's Python object graph into the plain descriptor tables of the C-ABI.

The reference walks Python objects once per surface inside `Raytracer.trace` (raytracer.py:274, 307-397);
here the walk happens once on the host and the device kernels only see `ot_scene_desc`.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from .geometry.elements import Lens, Filter, Aperture
from .geometry.surfaces import RectangularSurface, RingSurface, SlitSurface
from .spectrum import LightSpectrum


class CompiledScene:
    """ctypes tables + the `ot_scene_desc` pointing at them (keeps the arrays alive)."""

    def __init__(self, rt) -> None:
        self.elements_py = tracing_elements(rt)
        self.nt = len(rt.tracing_surfaces) + 2

        pool: list = []
        media: list = []
        media_ids: dict = {}
        lines = discrete_lines(rt.ray_sources)

        def medium_id(n) -> int:
            key = id(n)
            if key not in media_ids:
                media_ids[key] = len(media)
                media.append(n._desc(pool, lines))
            return media_ids[key]

        surfaces, elements, filters = [], [], []
        n0 = medium_id(rt.n0)
        for en, el in enumerate(self.elements_py):
            e = _capi.Element()
            e.front = len(surfaces)
            surfaces.append(el.front._desc())
            e.back = e.n_lens = e.n_after = e.filter = -1
            if isinstance(el, Lens):
                e.n_after = medium_id(el.n2 or rt.n0)
                if el.is_ideal:
                    e.kind = _capi.EL_IDEAL_LENS
                    e.D = el.D
                else:
                    e.kind = _capi.EL_LENS
                    e.back = len(surfaces)
                    surfaces.append(el.back._desc())
                    e.n_lens = medium_id(el.n)
            elif isinstance(el, Filter):
                e.kind = _capi.EL_FILTER
                e.filter = len(filters)
                filters.append(el.spectrum._desc(pool, lines))
            else:
                e.kind = _capi.EL_APERTURE
                last = en == len(self.elements_py) - 1
                e.hurb = int(bool(rt.use_hurb) and not last and isinstance(el.front, (RingSurface, SlitSurface)))
            elements.append(e)

        self.surfaces = (_capi.Surface * max(len(surfaces), 1))(*surfaces)
        self.elements = (_capi.Element * max(len(elements), 1))(*elements)
        self.media = (_capi.Medium * max(len(media), 1))(*media)
        self.filters = (_capi.Filter * max(len(filters), 1))(*filters)
        self.pool = (C.c_double * max(len(pool), 1))(*pool)
        self.n_hurb = sum(e.hurb for e in elements)

        d = _capi.SceneDesc()
        d.outline[:] = [float(v) for v in rt.outline]
        d.n_surfaces, d.n_elements, d.n_media, d.n_filters = len(surfaces), len(elements), len(media), len(filters)
        d.surfaces, d.elements, d.media, d.filters = self.surfaces, self.elements, self.media, self.filters
        d.table_pool = self.pool
        d.table_pool_len = len(pool)
        d.n0 = n0
        d.no_pol = int(rt.no_pol)
        d.use_hurb = int(rt.use_hurb)
        d.hurb_factor = float(rt.HURB_FACTOR)
        # discrete spectra: the library tabulates n(lambda), n1/n2 and filter values per line (LDS tables)
        self.lines = None
        d.n_lines = 0
        if lines is not None and 1 <= len(lines) <= _capi.OT_MAX_LINES:
            self.lines = (C.c_double * len(lines))(*[float(v) for v in lines])
            d.n_lines = len(lines)
            d.lines = self.lines
        self.desc = d


def tracing_elements(rt) -> list:
    """z-sorted lenses/filters/apertures plus the invisible absorbing end aperture at the outline's
    far z face (raytracer.py:492-508)."""
    o = rt.outline
    end = Aperture(RectangularSurface(dim=[o[1] - o[0], o[3] - o[2]]),
                   pos=[(o[1] + o[0]) / 2, (o[2] + o[3]) / 2, o[5]])
    return [el for el in rt.elements if isinstance(el, (Lens, Filter, Aperture))] + [end]


def discrete_lines(sources) -> np.ndarray | None:
    """Distinct float32 wavelengths if every source has a discrete spectrum, else None."""
    out = []
    for rs in sources:
        sp = rs.spectrum
        if rs._image is not None and type(rs._image).__name__ == "RGBImage":
            return None
        if not isinstance(sp, LightSpectrum) or sp.is_continuous():
            return None
        if sp.spectrum_type == "Monochromatic":
            out.append(np.float32(sp.wl))
        else:
            out.extend(np.asarray(sp.lines, dtype=np.float32).tolist())
    return np.unique(np.array(out, dtype=np.float32)) if out else None
