"""Small host-side vector helpers under the reference's names (optrace/tracer/misc.py:9-168).

Only post-processing on host copies uses them (RayStorage.rays_by_mask, user scripts); the device kernels carry
their own versions (csrc/ot_device.hpp: dot3, cross3, normalize3).
"""
import os

import numpy as np


def cpu_count() -> int:
    """Cores the process may use; PYTHON_CPU_COUNT (1..64) overrides the detected number."""
    detected = getattr(os, "process_cpu_count", os.cpu_count)() or 1
    n = int(os.environ.get("PYTHON_CPU_COUNT", detected))
    if n < 1 or n > 64:
        raise RuntimeError(f"Invalid core count {n}, must be between 1 and 64.")
    return n


def _rows(a: np.ndarray, cols: tuple) -> np.ndarray:
    a = np.asarray(a)
    if a.ndim != 2 or a.shape[1] not in cols:
        raise RuntimeError("Invalid number of dimensions.")
    return a


def rdot(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Scalar product of each row of a with the same row of b; rows have two or three components.

    >>> rdot(np.array([[1., 2., 3.], [4., 5., 6.]]), np.array([[0., 1., 0.], [1., 1., 1.]]))
    array([ 2., 15.])
    """
    a, b = _rows(a, (2, 3)), np.asarray(b)
    acc = a[:, 0] * b[:, 0]
    for c in range(1, a.shape[1]):  # left to right, like the spelled-out sum of the reference
        acc = acc + a[:, c] * b[:, c]
    return acc


def masked_assign(cond1: np.ndarray, cond2: np.ndarray) -> np.ndarray:
    """cond1 with its True entries replaced, in order, by the entries of cond2.

    >>> masked_assign(np.array([False, True, True, False, True]), np.array([True, False, True]))
    array([False,  True, False, False,  True])
    """
    res = np.zeros(cond1.shape, dtype=bool)
    res[np.flatnonzero(cond1)] = cond2
    return res


def normalize(a: np.ndarray) -> np.ndarray:
    """Rows of an (N, 3) array scaled to unit length; a zero row becomes NaN (no warning).

    >>> normalize(np.array([[3., 0., 4.], [0., 0., 0.]]))
    array([[0.6, 0. , 0.8],
           [nan, nan, nan]])
    """
    a = _rows(a, (3,))
    length = np.sqrt(a[:, 0] ** 2 + a[:, 1] ** 2 + a[:, 2] ** 2)
    with np.errstate(invalid="ignore", divide="ignore"):
        return a / length[:, None]


def cross(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Cross product of each row of a with the same row of b, as a Fortran-ordered (N, 3) float64 array.

    >>> cross(np.array([[1., 0., 0.], [0., 2., 0.]]), np.array([[0., 1., 0.], [0., 0., 3.]]))
    array([[0., 0., 1.],
           [6., 0., 0.]])
    """
    a, b = _rows(a, (3,)), np.asarray(b)
    out = np.empty(a.shape, dtype=np.float64, order="F")
    for c in range(3):
        i, j = (c + 1) % 3, (c + 2) % 3
        out[:, c] = a[:, i] * b[:, j] - a[:, j] * b[:, i]
    return out
