"""Oracle restatement of LightSpectrum.render against the reference's fixtures (CPU)."""
import pathlib
import sys

import numpy as np
import pytest

from helpers import load, assert_close

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from oracle import spectrum as ospec  # noqa: E402

CASES = ["c1_single_lens", "double_gauss", "mixed_geometry", "arizona_eye", "asphere", "hurb_slit_lens"]


@pytest.mark.parametrize("name", CASES)
def test_detector_spectrum_from_golden_hits(name):
    """Golden detector hits (trace_<name>.npz) -> oracle histogram == reference detector_spectrum."""
    g, sp = load(f"trace_{name}.npz"), load("spectra.npz")
    di = 0
    proj = "None" if f"det{di}/None/w" in g else "Equidistant"
    wls, vals = ospec.render(g[f"det{di}/{proj}/wl"], g[f"det{di}/{proj}/w"])
    assert_close(wls, sp[f"{name}/det{di}/all/wls"], rtol=0, atol=0, what="bin edges")
    tot = sp[f"{name}/det{di}/all/vals"].sum()
    # the reference sums float32 block-wise (np.histogram keeps the weights' dtype): 1e-6 of the total
    assert_close(vals, sp[f"{name}/det{di}/all/vals"], rtol=1e-6, atol=1e-6 * tot, what="vals")


@pytest.mark.parametrize("name", ["mixed_geometry", "arizona_eye", "double_gauss"])
def test_source_spectrum_from_golden_rays(name):
    g, sp = load(f"trace_{name}.npz"), load("spectra.npz")
    B = np.concatenate(([0], np.cumsum(g["N_list"])))
    for si in range(len(g["N_list"])):
        sl = slice(B[si], B[si + 1])
        wls, vals = ospec.render(g["wl"][sl], g["w0"][sl])
        assert_close(wls, sp[f"{name}/src{si}/wls"], rtol=0, atol=0, what="bin edges")
        tot = sp[f"{name}/src{si}/vals"].sum()
        assert_close(vals, sp[f"{name}/src{si}/vals"], rtol=1e-6, atol=1e-6 * tot, what="vals")


def test_big_bundle_bin_count_follows_sqrt_n():
    sp = load("spectra.npz")
    B = np.concatenate(([0], np.cumsum(sp["big/N_list"])))
    counts = []
    for si in range(2):
        sl = slice(B[si], B[si + 1])
        wls, vals = ospec.render(sp["big/wl"][sl], sp["big/w0"][sl])
        counts.append(vals.shape[0])
        assert_close(wls, sp[f"big/src{si}/wls"], rtol=0, atol=0, what="bin edges")
        tot = sp[f"big/src{si}/vals"].sum()
        assert_close(vals, sp[f"big/src{si}/vals"], rtol=1e-6, atol=1e-6 * tot, what="vals")
    assert counts[0] > 51 and counts[1] > counts[0] and all(c % 2 == 1 for c in counts)
