"""Small host-side vector helpers with the reference's names (optrace/tracer/misc.py).

They serve post-processing on host copies (RayStorage.rays_by_mask etc.); the device kernels carry
their own inlined versions.
"""
from __future__ import annotations

import os

import numpy as np


def cpu_count() -> int:
    """Logical CPU count, overridable with PYTHON_CPU_COUNT in 1..64 (misc.py:9-32)."""
    count = os.process_cpu_count() if hasattr(os, "process_cpu_count") else os.cpu_count()
    count = count or 1
    if "PYTHON_CPU_COUNT" in os.environ:
        count = int(os.environ["PYTHON_CPU_COUNT"])
    if not (1 <= count <= 64):
        raise RuntimeError(f"Invalid core count {count}, must be between 1 and 64.")
    return count


def rdot(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Row-wise scalar product for (N, 2) or (N, 3) arrays (misc.py:94-118).

    >>> rdot(np.array([[1., 2., 3.], [4., 5., 6.]]), np.array([[-1., 2., -3.], [7., 8., 9.]]))
    array([ -6., 122.])
    """
    if a.shape[1] == 3:
        return a[:, 0] * b[:, 0] + a[:, 1] * b[:, 1] + a[:, 2] * b[:, 2]
    if a.shape[1] == 2:
        return a[:, 0] * b[:, 0] + a[:, 1] * b[:, 1]
    raise RuntimeError("Invalid number of dimensions.")


def masked_assign(cond1: np.ndarray, cond2: np.ndarray) -> np.ndarray:
    """Write cond2 into the True positions of cond1 (misc.py:120-133).

    >>> masked_assign(np.array([True, False, False, True]), np.array([True, False]))
    array([ True, False, False, False])
    """
    out = np.zeros_like(cond1)
    out[cond1] = cond2
    return out


def normalize(a: np.ndarray) -> np.ndarray:
    """Unit vectors along axis 1; zero vectors give NaN (misc.py:136-150)."""
    with np.errstate(invalid="ignore"):
        return a / np.sqrt(a[:, 0] ** 2 + a[:, 1] ** 2 + a[:, 2] ** 2)[:, np.newaxis]


def cross(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Row-wise cross product of (N, 3) arrays (misc.py:152-168).

    >>> cross(np.array([[1., 2., 3.], [4., 5., 6.]]), np.array([[-1., 2., -3.], [7., 8., 9.]]))
    array([[-12.,   0.,   4.],
           [ -3.,   6.,  -3.]])
    """
    n = np.zeros_like(a, dtype=np.float64, order='F')
    n[:, 0] = a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1]
    n[:, 1] = a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2]
    n[:, 2] = a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]
    return n
