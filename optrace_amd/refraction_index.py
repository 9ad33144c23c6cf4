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


#: coefficient formulas: name -> (number of coefficients, device model code)
_FORMULAS = {
    "Cauchy": (4, _capi.N_CAUCHY), "Conrady": (3, _capi.N_CONRADY),
    "Sellmeier1": (6, _capi.N_SELLMEIER1), "Sellmeier2": (5, _capi.N_SELLMEIER2),
    "Sellmeier3": (8, _capi.N_SELLMEIER3), "Sellmeier4": (5, _capi.N_SELLMEIER4),
    "Sellmeier5": (10, _capi.N_SELLMEIER5), "Herzberger": (6, _capi.N_HERZBERGER),
    "Extended": (8, _capi.N_EXTENDED), "Extended2": (8, _capi.N_EXTENDED2), "Extended3": (9, _capi.N_EXTENDED3),
    "Handbook of Optics 1": (4, _capi.N_HOO1), "Handbook of Optics 2": (4, _capi.N_HOO2),
    "Schott": (6, _capi.N_SCHOTT),
}


class RefractionIndex(Spectrum):

    coeff_count = {name: count for name, (count, _) in _FORMULAS.items()}
    # the reference's order (refraction_index.py:22-26): alphabetical, with the two models added last at the end
    n_types = sorted(["Abbe", "Constant", "Data", "Function", *(k for k in _FORMULAS if k not in ("Herzberger", "Schott"))]) \
        + ["Herzberger", "Schott"]
    spectrum_types = n_types
    quantity = "Refraction Index n"
    unit = ""
    _amplitude_checked = False   # `val` is an index (>= 1), `func` returns indices: own checks below
    _models = {name: code for name, (_, code) in _FORMULAS.items()}

    def __init__(self, n_type: str = "Constant", n: float = 1.0, coeff: list = None, lines=None,
                 V: float = None, **kwargs) -> None:
        self.spectrum_type = n_type     # first: the coefficient count check needs it
        self.coeff, self.V = coeff, V
        super().__init__(n_type, val=n, lines=_FdC if lines is None else lines, **kwargs)
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
        x = np.asarray_chkfinite(wl, dtype=np.float64)
        kind = self.spectrum_type
        if kind == "Data" and x.size:
            lo, hi = self._wls[0], self._wls[-1]
            if x.min() < lo or x.max() > hi:
                raise RuntimeError(f"Wavelength range [{x.min():.5g}, {x.max():.5g}] larger than data range"
                                   f" [{lo}, {hi}] for this material.")
        if kind == "Function":
            ns = np.asarray(self.func(x, **self.func_args), dtype=np.float64)
        else:
            pool: list = []
            medium = self._desc(pool)
            ns = ops.refraction_index(medium, np.array(pool, dtype=np.float64), x.reshape(-1)).reshape(x.shape)
        if ns.size:
            worst = int(np.argmin(ns))
            if ns.flat[worst] < 1:
                raise RuntimeError(f"Refraction index below 1 with value {ns.flat[worst]:.4g} at {x.flat[worst]:.4g}nm.")
        return ns

    def __eq__(self, other: Any) -> bool:
        if self is other:
            return True
        if type(other) is not type(self):
            return False
        if "Data" not in (self.spectrum_type, other.spectrum_type):
            return self.crepr() == other.crepr()
        if self.spectrum_type != other.spectrum_type:
            return False
        return bool(np.array_equal(self._wls, other._wls) and np.array_equal(self._vals, other._vals))

    def __ne__(self, other: Any) -> bool:
        return not self == other

    __hash__ = object.__hash__

    def abbe_number(self, lines: list = None) -> float:
        """(n_centre - 1) / (n_short - n_long) at the three lines; inf for a medium without dispersion."""
        n_short, n_centre, n_long = self(self.lines if lines is None else lines)
        spread = n_short - n_long
        return float((n_centre - 1) / spread) if spread else float("inf")

    def is_dispersive(self) -> bool:
        return bool(np.isfinite(self.abbe_number()))

    def _check_coeff(self, coeff: list) -> list:
        check_type("coeff", coeff, list)
        need = self.coeff_count[self.spectrum_type]
        if len(coeff) != need:
            raise ValueError(f"coeff needs to be a list with exactly {need} numeric coefficients for mode "
                             f"{self.spectrum_type}, but got {len(coeff)}.")
        return list(coeff)

    def __setattr__(self, key: str, val: Any) -> None:
        if key == "val":
            check_type(key, val, (int, float))
            np.asarray_chkfinite(val)
            check_not_below(key, val, 1)
        elif key == "V" and val is not None:
            check_type(key, val, (float, int))
            check_above(key, val, 0)
            np.asarray_chkfinite(val)
        elif key == "coeff" and val is not None:
            val = self._check_coeff(val)
        elif key == "_vals" and val is not None and np.min(val) < 1:
            raise ValueError("all vals values needs to be at least 1.")
        elif key == "lines" and isinstance(val, (list, np.ndarray)):
            if len(val) != 3:
                raise ValueError("Property 'lines' for n_type='Abbe' needs to have exactly 3 elements")
            if not (val[0] < val[1] and val[1] < val[2]):
                raise ValueError("The values of property 'lines' need to be ascending.")
        elif key == "func" and callable(val) and np.min(val(wavelengths(1000), **self.func_args)) < 1:
            raise ValueError("Function func needs to output values >= 1 over the whole visible range.")
        super().__setattr__(key, val)
