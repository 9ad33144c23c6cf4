"""Shared comparison helpers for the parity tests."""
from __future__ import annotations

import pathlib

import numpy as np

GOLDEN = pathlib.Path(__file__).resolve().parent / "golden"


def load(name: str):
    return np.load(GOLDEN / name, allow_pickle=False)


def assert_close(a, b, rtol, atol=0.0, what=""):
    """NaN-aware closeness with a readable failure message."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    nan_a, nan_b = np.isnan(a), np.isnan(b)
    assert np.array_equal(nan_a, nan_b), f"{what}: NaN pattern differs at {np.argwhere(nan_a != nan_b)[:5]}"
    ok = ~nan_a
    inf = np.isinf(a) & ok
    assert np.array_equal(a[inf], b[inf]), f"{what}: inf differs"
    ok &= ~inf
    err = np.abs(a[ok] - b[ok])
    tol = atol + rtol * np.abs(b[ok])
    if np.any(err > tol):
        j = np.argmax(err - tol)
        raise AssertionError(f"{what}: max violation err={err[j]:.3e} tol={tol[j]:.3e} a={a[ok][j]!r} b={b[ok][j]!r} "
                             f"({np.count_nonzero(err > tol)} of {err.size} elements)")


def sparse_to_dense(g, key):
    shape = g[f"{key}/shape"]
    img = np.zeros(tuple(shape))
    img[g[f"{key}/iy"], g[f"{key}/ix"]] = g[f"{key}/val"]
    return img


def image_rel_l1(a, b):
    """sum |a-b| / sum |b| per channel (the image-norm criterion of SURVEY 8c for binned irradiance)."""
    return np.array([np.abs(a[..., c] - b[..., c]).sum() / max(np.abs(b[..., c]).sum(), 1e-300) for c in range(a.shape[-1])])
