"""Exactness of the division / square-root cores of the tracing loop (csrc/ot_device.hpp: ot_div, ot_rcp3 + ot_div_r,
ot_sqrt, normalize3) against IEEE `/` and sqrt, checked on the device by the library's own harness
(`ot_selftest_arith`, csrc/ot_selftest.hpp): bit-exact hit masks rest on it.

The reference computes with NumPy's IEEE double arithmetic (conic_surface.py:126-203, raytracer.py:761-829,
misc.py:136); the cores drop the range scaling and special-value fix-up of the compiler's IEEE sequences, so the claim
to verify is: same bits for every finite operand of ordinary magnitude, and a defined, harmless behaviour outside."""
import ctypes as C

import numpy as np
import pytest
import torch

from optrace_amd import _capi
from optrace_amd._device import require_device, stream_ptr, ptr, to_dev

pytestmark = pytest.mark.gpu

OPS = {"div": 0, "sqrt": 1, "normalize3": 2, "div_shared_rcp": 3}
CLASSES = {"wide_exponents": 0, "mm_geometry": 1, "refractive_indices": 2, "direction_cosines": 3}
N_PER_CASE = 1 << 27  # 1.3e8 operand sets per (operation, class): 2.1e9 in total


@pytest.mark.parametrize("cls", list(CLASSES))
@pytest.mark.parametrize("op", list(OPS))
def test_cores_return_the_bits_of_ieee(op, cls):
    lib = _capi.load_library()
    require_device()
    mism = C.c_int64(-1)
    bad = (C.c_double * 4)()
    _capi.check(lib.ot_selftest_arith(OPS[op], CLASSES[cls], N_PER_CASE, 20260 + 17 * OPS[op] + CLASSES[cls],
                                      C.byref(mism), bad, stream_ptr()))
    assert mism.value == 0, (f"{mism.value} of {N_PER_CASE} operand sets differ from IEEE for {op} on {cls}; "
                             f"one of them: a={bad[0]!r} b={bad[1]!r} c={bad[2]!r} core={bad[3]!r}")


def _eval(op, a, b=None, c=None):
    lib = _capi.load_library()
    require_device()
    n = len(a)
    da = to_dev(np.asarray(a, dtype=np.float64), np.float64)
    db = None if b is None else to_dev(np.asarray(b, dtype=np.float64), np.float64)
    dc = None if c is None else to_dev(np.asarray(c, dtype=np.float64), np.float64)
    core = torch.empty(3 * n, dtype=torch.float64, device=da.device)
    ieee = torch.empty(3 * n, dtype=torch.float64, device=da.device)
    _capi.check(lib.ot_selftest_eval(OPS[op], n, ptr(da), ptr(db), ptr(dc), ptr(core), ptr(ieee), stream_ptr()))
    torch.cuda.synchronize()
    return core.cpu().numpy().reshape(3, n), ieee.cpu().numpy().reshape(3, n)


def test_harness_sees_a_difference_where_there_is_one():
    """The comparison is not vacuous: outside the cores' domain the two columns differ (see the next test), and the
    IEEE column agrees with NumPy on the host."""
    a = np.array([1.0, 3.0, 2.0 ** 600, 5e-324, 1.0])
    b = np.array([3.0, 7.0, 2.0 ** -600, 3.0, 0.0])
    core, ieee = _eval("div", a, b)
    with np.errstate(all="ignore"):
        np.testing.assert_array_equal(ieee[0], a / b)
    assert np.array_equal(core[0, :2], ieee[0, :2])
    assert not np.array_equal(core[0, 2:], ieee[0, 2:], equal_nan=True)


def test_documented_behaviour_outside_the_domain():
    """Zero, infinite and subnormal operands (csrc/ot_device.hpp, comments of ot_sqrt / ot_rcp3):
    division: a zero or infinite denominator gives NaN (IEEE: +-inf / 0); quotients beyond 2^+-1022 are not guaranteed;
    sqrt: 0, +inf, NaN and negative arguments behave as sqrt does; subnormal arguments are outside the domain."""
    inf, nan = np.inf, np.nan
    core, ieee = _eval("div", np.array([1.0, -2.0, 0.0, 1.0, inf, nan, 0.0]), np.array([0.0, 0.0, 0.0, inf, 2.0, 1.0, 5.0]))
    assert np.all(np.isnan(core[0, :4])), "x / 0, 0 / 0 and x / inf come out as NaN"
    assert np.isinf(ieee[0, 0]) and np.isinf(ieee[0, 1]) and np.isnan(ieee[0, 2]) and ieee[0, 3] == 0.0
    assert np.isnan(core[0, 5]) and core[0, 6] == 0.0 and ieee[0, 6] == 0.0
    core, ieee = _eval("sqrt", np.array([0.0, inf, nan, 4.0, 2.0 ** -766, 2.0 ** 1022]))
    np.testing.assert_array_equal(core[0], ieee[0])
    assert core[0, 0] == 0.0 and core[0, 1] == inf and np.isnan(core[0, 2]) and core[0, 3] == 2.0
    # normalize3 of a zero vector: NaN like misc.py:136 (0 / 0)
    core, ieee = _eval("normalize3", np.array([0.0, 3.0]), np.array([0.0, 0.0]), np.array([0.0, 4.0]))
    assert np.all(np.isnan(core[:, 0])) and np.all(np.isnan(ieee[:, 0]))
    np.testing.assert_array_equal(core[:, 1], [0.6, 0.0, 0.8])


@pytest.mark.parametrize("surface", ["flat", "conic", "sphere"])
def test_callers_treat_nan_and_inf_alike_as_no_hit(surface):
    """Rays with s_z -> 0+ (and exactly 0): the reference's t = (z0 - p_z) / s_z overflows or becomes +-inf / NaN and the
    ray does not hit (surface.py:319-327, conic_surface.py:189-200); with the cores a zero denominator gives NaN where IEEE
    gives +-inf, and every caller must reach the same verdict.  Compared with the NumPy formulas of the reference,
    restated here for these few rays."""
    import optrace_amd as ot
    if surface == "flat":
        sf = ot.CircularSurface(r=3)
    elif surface == "conic":
        sf = ot.ConicSurface(r=3, R=10, k=-0.5)
    else:
        sf = ot.SphericalSurface(r=3, R=-12)
    sf.move_to([0.1, -0.2, 5.0])
    sz = np.array([1.0, 1e-3, 1e-8, 1e-150, 1e-300, 5e-324, 0.0])
    n = len(sz)
    p = np.tile(np.array([0.3, 0.1, 1.0]), (n, 1))
    s = np.stack([np.sqrt(np.maximum(1 - sz ** 2, 0)), np.zeros(n), sz], axis=1)
    ph, hit, ill = sf.find_hit(p, s)
    assert not ill.any()
    assert hit[0] and not hit[2:].any(), "only the first rays can reach a disc of radius 3 from 4 mm before it"
    # no-hit rays: position on the plane z = z_max along the ray where that is finite; the hit flags are what matters
    assert np.all(np.isfinite(ph[hit]))
    # the verdicts do not depend on whether the quotient came out as inf or NaN: same flags for s_z = 5e-324 and 0
    assert hit[-1] == hit[-2]
