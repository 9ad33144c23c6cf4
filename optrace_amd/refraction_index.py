"""Wavelength dependent refractive index of a medium.

Host-side mirror of optrace/tracer/refraction_index.py:11-265.  `__call__` evaluates n(lambda) on the
GPU (`ot_refraction_index`); inside `Raytracer.trace` the same device function is inlined into the
tracing kernel and fed from the `ot_medium` descriptor produced by `_desc`.
"""
from __future__ import annotations

from typing import Any

import numpy as np

from . import _capi
from .base import check_type, check_above, check_not_below
from .spectrum import Spectrum, wavelengths

#: Fraunhofer lines F, d, C in nm (presets/spectral_lines.py) -- default for the Abbe model
_FdC = [486.1327, 587.5618, 656.272]


class RefractionIndex(Spectrum):

    coeff_count = {"Cauchy": 4, "Conrady": 3, "Sellmeier1": 6, "Sellmeier2": 5, "Sellmeier3": 8,
                   "Sellmeier4": 5, "Sellmeier5": 10, "Herzberger": 6, "Extended": 8, "Extended2": 8,
                   "Handbook of Optics 1": 4, "Handbook of Optics 2": 4, "Schott": 6, "Extended3": 9}

    n_types = ["Abbe", "Cauchy", "Conrady", "Constant", "Data", "Extended", "Extended2", "Extended3",
               "Function", "Handbook of Optics 1", "Handbook of Optics 2", "Sellmeier1", "Sellmeier2",
               "Sellmeier3", "Sellmeier4", "Sellmeier5", "Herzberger", "Schott"]
    spectrum_types = n_types
    quantity = "Refraction Index n"
    unit = ""

    _models = {"Cauchy": _capi.N_CAUCHY, "Conrady": _capi.N_CONRADY, "Sellmeier1": _capi.N_SELLMEIER1,
               "Sellmeier2": _capi.N_SELLMEIER2, "Sellmeier3": _capi.N_SELLMEIER3,
               "Sellmeier4": _capi.N_SELLMEIER4, "Sellmeier5": _capi.N_SELLMEIER5,
               "Herzberger": _capi.N_HERZBERGER, "Extended": _capi.N_EXTENDED,
               "Extended2": _capi.N_EXTENDED2, "Extended3": _capi.N_EXTENDED3,
               "Handbook of Optics 1": _capi.N_HOO1, "Handbook of Optics 2": _capi.N_HOO2,
               "Schott": _capi.N_SCHOTT}

    def __init__(self, n_type: str = "Constant", n: float = 1.0, coeff: list = None, lines=None,
                 V: float = None, **kwargs) -> None:
        self.spectrum_type = n_type
        self.coeff = coeff
        self.V = V
        lines = lines if lines is not None else _FdC
        super().__init__(n_type, val=n, lines=lines, **kwargs)
        self._new_lock = True

    # ---- device descriptor ----------------------------------------------------------------------
    def _abbe_AB(self) -> tuple[float, float, float]:
        """A, B, d of n = A + B/(wl2 - d) from (n_c, V, lines), refraction_index.py:85-98.

        The arithmetic is spelled exactly like the reference's: `self.lines` is a float32 array there
        (spectrum.py:168), so A and B come out as float32 values and are used as such.
        """
        l = 1e-3 * np.array(self.lines)
        nc = self.val
        d = 0.014
        B = 1 / self.V * (nc - 1) / (1 / (l[0] ** 2 - d) - 1 / (l[2] ** 2 - d))
        A = nc - B / (l[1] ** 2 - d)
        return float(A), float(B), d

    def _desc(self, pool: list, lines: np.ndarray | None = None) -> _capi.Medium:
        """`ot_medium` of this index; tables are appended to `pool` (list of floats)."""
        m = _capi.Medium()
        st = self.spectrum_type
        if st == "Constant":
            m.model = _capi.N_CONSTANT
            m.c[0] = float(self.val)
        elif st == "Abbe":
            if self.V is None:
                raise TypeError("Abbe number V needs to be provided for n_type='Abbe'")
            m.model = _capi.N_ABBE
            m.c[0], m.c[1], m.c[2] = self._abbe_AB()
        elif st in self._models:
            if self.coeff is None:
                raise TypeError(f"coefficient variable 'coeff' needs to be provided for n_type='{st}'.")
            m.model = self._models[st]
            for j, c in enumerate(self.coeff):
                m.c[j] = float(c)
        elif st == "Data":
            m.model = _capi.N_DATA
            m.tab_off, m.tab_len = len(pool), len(self._wls)
            pool.extend(self._wls.tolist())
            pool.extend(self._vals.tolist())
        elif st == "Function":
            if lines is not None:
                m.model = _capi.N_LINES
                x = lines.astype(np.float64)
            else:
                m.model = _capi.N_DATA
                x = wavelengths(65537)
            v = np.asarray(self.func(x, **self.func_args), dtype=np.float64)
            m.tab_off, m.tab_len = len(pool), len(x)
            pool.extend(x.tolist())
            pool.extend(v.tolist())
        else:
            raise AssertionError(st)
        return m

    def __call__(self, wl) -> np.ndarray:
        """n at the given wavelengths [nm] (refraction_index.py:62-169), computed on the GPU."""
        from . import ops
        wl_ = np.asarray_chkfinite(wl, dtype=np.float64)
        if self.spectrum_type == "Data" and wl_.size and (wl_.min() < self._wls[0] or wl_.max() > self._wls[-1]):
            raise RuntimeError(f"Wavelength range [{wl_.min():.5g}, {wl_.max():.5g}] larger than data range"
                               f" [{self._wls[0]}, {self._wls[-1]}] for this material.")
        if self.spectrum_type == "Function":
            ns = np.asarray(self.func(wl_, **self.func_args), dtype=np.float64)
        else:
            pool: list = []
            md = self._desc(pool)
            ns = ops.refraction_index(md, np.array(pool, dtype=np.float64), wl_.reshape(-1)).reshape(wl_.shape)
        if ns.size and (nm := ns.min()) < 1:
            raise RuntimeError(f"Refraction index below 1 with value {nm:.4g} at {wl_.flat[np.argmin(ns)]:.4g}nm.")
        return ns

    def __eq__(self, other: Any) -> bool:
        if type(self) is not type(other):
            return False
        if self is other or (self.spectrum_type != "Data" and self.crepr() == other.crepr()):
            return True
        if self.spectrum_type == "Data" and other.spectrum_type == "Data":
            return bool(np.all(self._wls == other._wls) and np.all(self._vals == other._vals))
        return False

    def __ne__(self, other: Any) -> bool:
        return not self.__eq__(other)

    __hash__ = object.__hash__

    def abbe_number(self, lines: list = None) -> float:
        lines = lines if lines is not None else self.lines
        ns, nc, nl = tuple(self(lines))
        return float((nc - 1) / (ns - nl) if ns != nl else np.inf)

    def is_dispersive(self) -> bool:
        return bool(np.isfinite(self.abbe_number()))

    def __setattr__(self, key: str, val: Any) -> None:
        if key == "val":
            check_type(key, val, (int, float))
            np.asarray_chkfinite(val)
            check_not_below(key, val, 1)
        elif key == "coeff" and val is not None:
            check_type(key, val, list)
            cnt = self.coeff_count[self.spectrum_type]
            if len(val) != cnt:
                raise ValueError(f"{key} needs to be a list with exactly {cnt} numeric coefficients for mode "
                                 f"{self.spectrum_type}, but got {len(val)}.")
            val = val.copy()
        elif key == "_vals" and val is not None:
            if np.min(val) < 1:
                raise ValueError("all vals values needs to be at least 1.")
        elif key == "lines" and isinstance(val, (list, np.ndarray)):
            if len(val) != 3:
                raise ValueError("Property 'lines' for n_type='Abbe' needs to have exactly 3 elements")
            if not val[0] < val[1] < val[2]:
                raise ValueError("The values of property 'lines' need to be ascending.")
        elif key == "func" and callable(val):
            n = val(wavelengths(1000), **self.func_args)
            if n.min() < 1:
                raise ValueError("Function func needs to output values >= 1 over the whole visible range.")
        elif key == "V" and val is not None:
            check_type(key, val, (float, int))
            check_above(key, val, 0)
            np.asarray_chkfinite(val)
        super().__setattr__(key, val)
