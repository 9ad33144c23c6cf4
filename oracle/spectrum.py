"""CPU restatement of LightSpectrum.render (optrace/tracer/spectrum/light_spectrum.py:41-79).

TEST INFRASTRUCTURE ONLY: the checker for ot_spectrum_range / ot_spectrum_histogram.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product path never does.

The reference leaves the binning itself to np.histogram; restating NumPy would pin nothing, so the weighted
histogram is written out here as the plain loop NumPy's uniform-bin path performs in float32
(numpy/lib/_histograms_impl.py, `histogram`, "fast algorithm for equal bins"), and the test-suite checks it
against the fixtures the reference produced (tests/golden/spectra.npz).
"""
from __future__ import annotations

import numpy as np

WL_RANGE = (380., 780.)  # global_options.wavelength_range default


def bin_count(w: np.ndarray) -> int:
    """light_spectrum.py:60-61: at least 51 bins, sqrt(N)/2 above that, made odd."""
    N = max(51, np.sqrt(np.count_nonzero(w)) / 2)
    return 1 + 2 * (int(N) // 2)


def render(wl: np.ndarray, w: np.ndarray):
    """(wls, vals): bin edges and power per nm of the selected rays (all rays given here are selected)."""
    wl = np.asarray(wl, dtype=np.float32)
    w = np.asarray(w, dtype=np.float32)
    N = bin_count(w)
    if not wl.shape[0]:
        return np.linspace(WL_RANGE[0], WL_RANGE[1], N + 1), np.zeros(N)
    wl0, wl1 = wl.min(), wl.max()
    if np.abs(wl0 - wl1) < 1:  # light_spectrum.py:73-74
        wl0, wl1 = max(wl0 - 1, WL_RANGE[0]), min(wl0 + 1, WL_RANGE[1])
    first, last = np.float32(wl0), np.float32(wl1)
    edges = np.linspace(first, last, N + 1, endpoint=True, dtype=np.float32)
    denom = np.float32(last - first)
    sums = np.zeros(N, dtype=np.float64)
    for x, wi in zip(wl, w):
        if not (first <= x <= last):
            continue
        idx = int(np.float32(np.float32(np.float32(x - first) / denom) * np.float32(N)))
        if idx == N:
            idx -= 1
        if x < edges[idx]:
            idx -= 1
        if x >= edges[idx + 1] and idx != N - 1:
            idx += 1
        sums[idx] += float(wi)
    return edges.astype(np.float64), sums * (1 / (float(edges[1]) - float(edges[0])))
