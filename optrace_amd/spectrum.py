"""Spectra: base class, light spectra of sources, transmission spectra of filters.

Host-side mirror of optrace/tracer/spectrum/{spectrum,light_spectrum,transmission_spectrum}.py.
The objects hold parameters and describe themselves to the device: a LightSpectrum becomes the
`spectrum` part of an `ot_source` (wavelength sampling happens inside the ray-generation kernel),
a TransmissionSpectrum becomes an `ot_filter`.
"""
from __future__ import annotations

import copy
import pathlib
from typing import Any, Callable

import numpy as np

from . import _capi
from .base import BaseClass, check_type, check_in, check_not_above
from .options import global_options as go

_tables = np.load(pathlib.Path(__file__).resolve().parent / "data" / "cie_tables.npz")
_illuminants = _tables["illuminants"]
_ill_names = [str(s) for s in _tables["illuminant_names"]]


def wavelengths(N: int) -> np.ndarray:
    """N equally spaced wavelengths over global_options.wavelength_range (color/tools.py:13-21)."""
    return np.linspace(*go.wavelength_range, N)


def illuminant(name: str) -> Callable[[np.ndarray], np.ndarray]:
    """Linear interpolation of a CIE standard illuminant table (color/illuminants.py:16-...)."""
    col = _ill_names.index(name)

    def f(wl):
        return np.interp(wl, _illuminants[:, 0], _illuminants[:, col], left=0, right=0)
    f.__name__ = f"{name.lower()}_illuminant"
    return f


d65_illuminant = illuminant("D65")


def blackbody(wl: np.ndarray, T: float = 6504.) -> np.ndarray:
    """Planck spectral radiance (color/tools.py:24-44)."""
    c, h, k_B = 299792458.0, 6.62607015e-34, 1.380649e-23
    wlm = 1e-9 * wl
    return 2 * h * c ** 2 / wlm ** 5 / (np.exp(h * c / (wlm * k_B * T)) - 1)


def normalized_blackbody(wl: np.ndarray, T: float = 6504.) -> np.ndarray:
    """Blackbody curve with its maximum inside the visible range scaled to 1 (color/tools.py:46-61)."""
    lo, hi = go.wavelength_range
    peak = 2897.771955e3 / T  # Wien's displacement law, in nm
    candidates = [peak] if lo <= peak <= hi else [lo, hi]
    return blackbody(wl, T) / blackbody(np.array(candidates), T).max()


def _finite(wl) -> np.ndarray:
    return np.asarray_chkfinite(wl, dtype=np.float64)


def _private(values, dtype) -> np.ndarray:
    """Checked private copy that changes only by assignment (the scene's change detection relies on that)."""
    arr = np.array(np.asarray_chkfinite(values, dtype=dtype))
    arr.setflags(write=False)
    return arr


def _visible(key: str, v) -> float:
    check_type(key, v, (int, float))
    lo, hi = go.wavelength_range
    if not lo <= v <= hi:
        raise ValueError(f"Property '{key}' needs to be inside {go.wavelength_range}, but is {float(v)}.")
    return float(v)


def _above_zero(key: str, v) -> float:
    check_type(key, v, (int, float))
    if not v > 0:
        raise ValueError(f"{key} needs to be above 0")
    return float(v)


class Spectrum(BaseClass):
    """Parametrised spectrum (spectrum.py:12-260)."""

    spectrum_types = ["Monochromatic", "Constant", "Data", "Lines", "Rectangle", "Gaussian", "Function"]
    unit = ""
    quantity = ""
    _amplitude_checked = True    # RefractionIndex stores indices in `val` and `func`: no sign checks there

    def __init__(self, spectrum_type: str = "Gaussian", val: float = 1., lines=None, line_vals=None,
                 wl: float = 550., wl0: float = 400., wl1: float = 600., wls=None, vals=None,
                 func: Callable = None, mu: float = 550., sig: float = 50., unit: str = None,
                 quantity: str = None, func_args: dict = {}, **kwargs) -> None:
        given = dict(spectrum_type=spectrum_type, lines=lines, line_vals=line_vals, func_args=func_args, func=func,
                     wl=wl, wl0=wl0, wl1=wl1, val=val, mu=mu, sig=sig, _wls=wls, _vals=vals)
        for name, value in given.items():   # func_args before func: the function is probed with them
            setattr(self, name, value)
        for name, value in (("unit", unit), ("quantity", quantity)):
            if value is not None:
                setattr(self, name, value)
        super().__init__(**kwargs)
        self._new_lock = True

    def is_continuous(self) -> bool:
        return self.spectrum_type not in ("Lines", "Monochromatic")

    def _eval_host(self, wl) -> np.ndarray:
        """Spectrum values on the host -- scene set-up (tabulating callables) only (spectrum.py:81-120)."""
        kind = self.spectrum_type
        if not self.is_continuous():
            raise RuntimeError(f"Can't call discontinuous spectrum_type '{kind}'")
        x = _finite(wl)
        if kind == "Constant":
            return np.broadcast_to(self.val, x.shape)
        if kind == "Data":
            return np.interp(x, self._wls, self._vals, left=0., right=0.)
        if kind == "Rectangle":
            return np.where((x >= self.wl0) & (x <= self.wl1), self.val, 0.)
        if kind == "Gaussian":
            return self.val * np.exp(-(x - self.mu) ** 2 / (2 * self.sig ** 2))
        if kind == "Function":
            return self.func(x, **self.func_args)
        raise AssertionError(kind)

    def __call__(self, wl) -> np.ndarray:
        return self._eval_host(wl)

    def get_desc(self, fallback: str = None) -> str:
        own = self.spectrum_type if self.spectrum_type != "Constant" else str(self.val)
        return super().get_desc(fallback=own)

    # ---- property checks, one small method per group of properties
    def _check_lines(self, key, v):
        check_type(key, v, (list, np.ndarray))
        arr = _private(v, np.float32)  # float32 like the reference (spectrum.py:168)
        if not arr.shape[0]:
            raise ValueError(f"'{key}' can't be empty.")
        if key == "line_vals":
            if arr.min() < 0:
                raise ValueError("line_vals must be all positive.")
            return arr
        lo, hi = go.wavelength_range
        if arr.min() < lo or arr.max() > hi:
            raise ValueError(f"'lines' need to be inside visible range {go.wavelength_range}.")
        if np.unique(v).size != len(v):
            raise ValueError("All elements inside of 'lines' must be unique.")
        return arr

    def _check_table(self, key, v):
        check_type(key, v, (list, np.ndarray))
        arr = _private(v, np.float64)
        if key == "_vals":
            if arr.min() < 0:
                raise ValueError("vals must be all positive")
            return arr
        lo, hi = go.wavelength_range
        if arr[0] < lo or arr[-1] > hi:
            raise ValueError("wls needs to be inside the visible range")
        steps = np.diff(arr)
        if steps.std() > 1e-4 or steps.min() <= 0 or steps[0] < 1e-6:
            raise ValueError("wls needs to be monotonically increasing with the same step size.")
        return arr

    def _check_func(self, key, f):
        if f is None:
            return None
        if not callable(f):
            raise TypeError("func needs to be callable or None")
        if self._amplitude_checked:
            probe = f(wavelengths(10000), **self.func_args)
            if np.min(probe) < 0 or not np.max(probe) > 0:
                raise RuntimeError("Function func needs to return positive values over the visible range.")
        return f

    def __setattr__(self, key: str, val: Any) -> None:
        if key == "spectrum_type":
            check_type(key, val, str)
            check_in(key, val, self.spectrum_types)
        elif key in ("quantity", "unit"):
            check_type(key, val, str)
        elif key == "func_args":
            check_type(key, val, dict)
            val = copy.deepcopy(val)
        elif key == "func":
            val = self._check_func(key, val)
        elif key in ("wl", "wl0", "wl1", "mu"):
            val = _visible(key, val)
        elif key == "sig":
            val = _above_zero(key, val)
        elif key == "val" and self._amplitude_checked:
            check_type(key, val, (int, float))
            if val < 0:
                raise ValueError("val needs to be at least 0")
            val = float(val)
        elif val is not None and key in ("lines", "line_vals"):
            val = self._check_lines(key, val)
        elif val is not None and key in ("_wls", "_vals"):
            val = self._check_table(key, val)
        super().__setattr__(key, val)


class LightSpectrum(Spectrum):
    """Spectrum of a light source (light_spectrum.py:13-300); sampling runs in the generation kernel."""

    spectrum_types = [*Spectrum.spectrum_types, "Blackbody", "Histogram"]

    def __init__(self, spectrum_type: str = "Blackbody", T: float = 5500, **sargs) -> None:
        self.T = T
        discrete = spectrum_type in ("Monochromatic", "Lines")   # powers per line, densities otherwise
        super().__init__(spectrum_type, unit="W" if discrete else "W/nm",
                         quantity="Spectral Power" + ("" if discrete else " Density"), **sargs)

    def _eval_host(self, wl) -> np.ndarray:
        kind = self.spectrum_type
        if kind == "Blackbody":
            return self.val * normalized_blackbody(_finite(wl), T=self.T)
        if kind == "Histogram":
            x = _finite(wl)
            edges, heights = self._wls, self._vals
            bin_ = np.searchsorted(edges, x, side="right") - 1    # bin i covers [edge i, edge i+1)
            inside = (bin_ >= 0) & (bin_ < len(edges) - 1)
            return np.where(inside, heights[np.clip(bin_, 0, len(heights) - 1)], 0.)
        return super()._eval_host(wl)

    def random_wavelengths(self, N: int) -> np.ndarray:
        """N wavelengths distributed like this spectrum (light_spectrum.py:81-138) -- drawn where the tracer draws them: by
        the generation kernel (stratified inverse-CDF sampling on the device, `ot_generate.hpp`), for a point source that
        carries this spectrum.  -> float64 host array (values of float32 resolution, like `RayStorage.wl_list`)."""
        from .geometry import Point
        from .geometry.ray_source import RaySource
        from . import ops
        N = int(N)
        if N < 1:
            return np.zeros(0, dtype=np.float64)
        return ops.create_rays(RaySource(Point(), spectrum=self), N, no_pol=True)[4]

    # ---- figures of a spectrum (light_spectrum.py:232-400): host arithmetic on the spectrum's own description ----
    _DENSE = 100000   # samples over the visible range where a figure has no closed form

    def _dense(self):
        grid = wavelengths(self._DENSE)
        return grid, self._eval_host(grid)

    def _integral(self, sensitivity) -> float:
        """Integral of sensitivity(wl) * spectrum over the wavelength; sums for line spectra, bin sums for histograms."""
        kind = self.spectrum_type
        if kind == "Monochromatic":
            return float(sensitivity(self.wl) * self.val)
        if kind == "Lines":
            return float(np.sum(sensitivity(self.lines) * self.line_vals))
        if kind == "Histogram":
            width = self._wls[1] - self._wls[0]
            centres = self._wls[:-1] + width / 2
            return float(np.sum(sensitivity(centres) * self._vals) * width)
        grid, values = self._dense()
        weighted = sensitivity(grid) * values
        return float(np.sum((weighted[1:] + weighted[:-1]) / 2) * (grid[1] - grid[0]))

    def power(self) -> float:
        """Power in W."""
        return self._integral(lambda wl: np.ones_like(wl, dtype=np.float64))

    def luminous_power(self) -> float:
        """Luminous power in lm (683 lm/W times the CIE 1931 y observer)."""
        table = _tables["observers"]
        return self._integral(lambda wl: 683.0 * np.interp(wl, table[:, 0], table[:, 2], left=0, right=0))

    def peak(self) -> float:
        """Highest value of the spectrum."""
        kind = self.spectrum_type
        if kind in ("Monochromatic", "Gaussian", "Rectangle", "Constant", "Blackbody"):
            return float(self.val)
        if kind == "Lines":
            return float(np.max(self.line_vals))
        if kind in ("Histogram", "Data"):
            return float(np.max(self._vals))
        return float(np.max(self._dense()[1]))

    def peak_wavelength(self) -> float:
        """Wavelength of the (first) highest value."""
        kind = self.spectrum_type
        if kind == "Monochromatic":
            return float(self.wl)
        if kind == "Lines":
            return float(self.lines[int(np.argmax(self.line_vals))])
        if kind == "Rectangle":
            return float(self.wl0)
        if kind == "Constant":
            return float(go.wavelength_range[0])
        if kind == "Gaussian":
            return float(self.mu)
        grid, values = self._dense()
        return float(grid[int(np.argmax(values))])

    def centroid_wavelength(self) -> float:
        """Power-weighted mean wavelength."""
        kind = self.spectrum_type
        middle = float(np.mean(go.wavelength_range))
        if kind == "Monochromatic":
            return float(self.wl)
        if kind == "Lines":
            lines, weights = np.array(self.lines), np.array(self.line_vals)
            return float(np.sum(weights * lines) / np.sum(weights))
        if kind == "Rectangle":
            return float(np.mean([self.wl0, self.wl1]))
        if kind == "Constant":
            return middle
        grid, values = self._dense()
        if not np.any(values > 0):
            return middle
        moment = grid * values
        return float(np.sum((moment[1:] + moment[:-1]) / 2) / np.sum((values[1:] + values[:-1]) / 2))

    def fwhm(self) -> float:
        """Full width at half maximum around the highest peak: the nearest crossings of half its height on both sides."""
        kind = self.spectrum_type
        if kind in ("Monochromatic", "Lines"):
            return 0.0
        if kind == "Rectangle":
            return float(self.wl1 - self.wl0)
        if kind == "Constant":
            return float(go.wavelength_range[1] - go.wavelength_range[0])
        grid, values = self._dense()
        top = int(np.argmax(values))
        below = values < 0.5 * values[top]
        right = np.flatnonzero(below[top:])
        left = np.flatnonzero(below[:top][::-1])
        hi = top + int(right[0]) if right.size else values.shape[0] - 1
        lo = top - int(left[0]) if left.size else 0
        return float(grid[hi] - grid[lo])

    @staticmethod
    def render(wl, w, _fill=None, **kwargs) -> "LightSpectrum":
        """Histogram spectrum (unit W/nm) of rays with wavelengths `wl` and powers `w`
        (light_spectrum.py:41-79).  `wl`, `w`: float32 device tensors (or host arrays, uploaded as they are);
        rays with weight 0 count as not selected, which is how `Raytracer._hit_detector` hands them over.
        Range search and binning run on the GPU (ot_spectrum_range / ot_spectrum_histogram).  `_fill`: wl and w are a
        compact hit list (`ot_detector_req.fill`: 1024 pieces, piece k holding `_fill[k]` entries at its front)."""
        import torch
        from ._device import require_device, ptr, stream_ptr
        lib = _capi.load_library()
        dev = require_device()
        wl = torch.as_tensor(np.ascontiguousarray(wl) if isinstance(wl, np.ndarray) else wl).to(dev, torch.float32).contiguous()
        w = torch.as_tensor(np.ascontiguousarray(w) if isinstance(w, np.ndarray) else w).to(dev, torch.float32).contiguous()
        if wl.shape != w.shape or wl.ndim != 1:
            raise ValueError("wl and w need to be one-dimensional arrays of the same length.")
        spec = LightSpectrum("Histogram", **kwargs)
        n = int(wl.shape[0])

        rng = torch.empty(2, dtype=torch.float64, device=dev)
        cnt = torch.empty(1, dtype=torch.int64, device=dev)
        if _fill is not None:
            _capi.check(lib.ot_spectrum_range_compact(n, ptr(_fill), ptr(wl), ptr(w), ptr(rng), ptr(cnt), stream_ptr()))
        else:
            _capi.check(lib.ot_spectrum_range(n, ptr(wl), ptr(w), ptr(rng), ptr(cnt), stream_ptr()))
        nz = int(cnt.item())

        # at least 51 bins, growing with sqrt(N) above that; odd, so there is a bin for the range centre
        N = max(51, np.sqrt(nz) / 2)
        N = 1 + 2 * (int(N) // 2)

        if not nz:
            spec._wls = wavelengths(N + 1)
            spec._vals = np.zeros(N, dtype=np.float64)
            return spec

        wl0, wl1 = (np.float32(v) for v in rng.cpu().numpy())
        if np.abs(wl0 - wl1) < 1:  # widen to +-1 nm inside the visible range
            wl0, wl1 = max(wl0 - 1, go.wavelength_range[0]), min(wl0 + 1, go.wavelength_range[1])
        # float32 edges exactly as np.histogram builds them for float32 data and range
        edges = np.linspace(wl0, wl1, N + 1, endpoint=True, dtype=np.float32)
        d_edges = torch.from_numpy(edges).to(dev)
        hist = torch.zeros(N, dtype=torch.float64, device=dev)
        if _fill is not None:
            _capi.check(lib.ot_spectrum_histogram_compact(n, ptr(_fill), ptr(wl), ptr(w), ptr(d_edges), N, ptr(hist),
                                                          stream_ptr()))
        else:
            _capi.check(lib.ot_spectrum_histogram(n, ptr(wl), ptr(w), ptr(d_edges), N, ptr(hist), stream_ptr()))
        spec._wls = edges
        spec._vals = hist.cpu().numpy() * (1 / (spec._wls[1] - spec._wls[0]))  # W -> W/nm
        return spec

    def _source_fields(self) -> dict:
        """Spectrum part of an `ot_source` for LightSpectrum.random_wavelengths (light_spectrum.py:81-138)."""
        st = self.spectrum_type
        if st == "Monochromatic":
            return dict(spectrum=_capi.SPEC_MONO, wl=float(self.wl))
        if st in ("Constant", "Rectangle"):
            wl0 = go.wavelength_range[0] if st == "Constant" else self.wl0
            wl1 = go.wavelength_range[1] if st == "Constant" else self.wl1
            return dict(spectrum=_capi.SPEC_UNIFORM, wl0=float(wl0), wl1=float(wl1))
        if st == "Lines":
            check_type("LightSpectrum.lines", self.lines, (np.ndarray, list))
            check_type("LightSpectrum.line_vals", self.line_vals, (np.ndarray, list))
            if not self.line_vals.sum():
                raise RuntimeError("Cumulated probability is zero.")
            keep = self.line_vals > 0
            # discrete inverse CDF: cumulative sums are taken in float32 like the reference does
            # (random.py:133-136 on the float32 line_vals)
            F = np.cumsum(self.line_vals[keep])
            tab = np.concatenate((self.lines[keep].astype(np.float64), F.astype(np.float64)))
            return dict(spectrum=_capi.SPEC_LINES, spec_tab=tab, n_spec=int(keep.sum()))
        if st == "Gaussian":
            return dict(spectrum=_capi.SPEC_GAUSSIAN, mu=float(self.mu), sig=float(self.sig),
                        wl0=float(go.wavelength_range[0]), wl1=float(go.wavelength_range[1]))
        # Data / Blackbody / Function / Histogram: linear inverse CDF of a tabulated pdf
        if st == "Data":
            x, f = self._wls, self._vals
        else:
            x = wavelengths(4000 if st == "Blackbody" else 10000)
            f = self._eval_host(x)
        f = np.asarray(f, dtype=np.float64)
        if not f.sum():
            raise RuntimeError("Cumulated probability is zero.")
        if f.min() < 0:
            raise RuntimeError("Got negative value in pdf.")
        # cumulative trapezoid with unit spacing (random.py:150: scipy cumulative_trapezoid(f, initial=0))
        F = np.concatenate(([0.], np.cumsum((f[1:] + f[:-1]) / 2)))
        return dict(spectrum=_capi.SPEC_TABLE, spec_tab=np.concatenate((np.asarray(x, dtype=np.float64), F)),
                    n_spec=len(x))

    def __setattr__(self, key, val):
        if key == "_vals" and val is not None and self.spectrum_type != "Histogram":
            # a user's data spectrum: no negative values and not zero throughout (light_spectrum.py:418-426)
            given = np.asarray_chkfinite(val, dtype=np.float64)
            if (given < 0).any():
                raise ValueError("Values below zero in LightSpectrum.")
            if not (given > 0).any():
                raise ValueError("LightSpectrum can't be constantly zero.")
        super().__setattr__(key, _above_zero(key, val) if key == "T" else val)


class TransmissionSpectrum(Spectrum):
    """Transmittance of a filter, range [0, 1] (transmission_spectrum.py:11-110)."""

    spectrum_types = ["Constant", "Data", "Rectangle", "Gaussian", "Function"]
    quantity = "Transmission T"

    def __init__(self, spectrum_type: str = "Gaussian", inverse: bool = False, **sargs) -> None:
        self.inverse = inverse
        super().__init__(spectrum_type, **sargs)

    def _eval_host(self, wl):
        v = super()._eval_host(wl)
        return v if not self.inverse else 1.0 - v

    def _desc(self, pool: list, lines: np.ndarray | None) -> _capi.Filter:
        """`ot_filter` for this spectrum; tables are appended to `pool`.  `lines` = the distinct f32
        wavelengths of all sources if every source is discrete (needed for "Function" spectra)."""
        f = _capi.Filter()
        f.inverse = int(self.inverse)
        st = self.spectrum_type
        f.val, f.wl0, f.wl1, f.mu, f.sig = float(self.val), float(self.wl0), float(self.wl1), float(self.mu), float(self.sig)
        if st == "Constant":
            f.type = _capi.T_CONSTANT
        elif st == "Rectangle":
            f.type = _capi.T_RECTANGLE
        elif st == "Gaussian":
            f.type = _capi.T_GAUSSIAN
        elif st == "Data":
            f.type = _capi.T_DATA
            f.tab_off, f.tab_len = len(pool), len(self._wls)
            pool.extend(self._wls.tolist())
            pool.extend(self._vals.tolist())
        else:  # Function: exact per line for discrete sources, fine table otherwise
            f.inverse = 0  # folded into the table
            if lines is not None:
                f.type = _capi.T_LINES
                x = lines.astype(np.float64)
            else:
                f.type = _capi.T_DATA
                x = wavelengths(65537)
            v = np.asarray(self._eval_host(x), dtype=np.float64)
            f.tab_off, f.tab_len = len(pool), len(x)
            pool.extend(x.tolist())
            pool.extend(v.tolist())
        return f

    def __setattr__(self, key, val):
        if key == "val" and isinstance(val, (int, float)):
            check_not_above(key, val, 1)
        if key == "_vals" and isinstance(val, (list, np.ndarray)):
            if np.max(val) > 1:
                raise ValueError("all elements in vals need to be in range [0, 1].")
        if key == "inverse":
            check_type(key, val, bool)
        if key == "func" and callable(val):
            if np.any(val(wavelengths(1000)) > 1):
                raise RuntimeError("Function func needs to return values in range [0, 1] over the visible range.")
        super().__setattr__(key, val)
