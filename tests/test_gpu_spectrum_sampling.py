"""`LightSpectrum.random_wavelengths` (light_spectrum.py:81-138; here: the generation kernel's wavelength sampler behind a
point source) and `LightSpectrum.render` held to the reference's own tests: tests/test_spectrum.py:247-363
(test_light_spectrum_random_wavelengths) and :365-435 (test_light_spectrum_render), restated with their bounds -- means and
standard deviations of every spectrum type against the closed forms (uniform, discrete, normal, truncated normal), the
rendered histogram against the illuminant it was drawn from."""
import numpy as np
import pytest
from scipy.special import erf

import optrace_amd as ot
from optrace_amd.spectrum import wavelengths

pytestmark = pytest.mark.gpu
WL0, WL1 = 380.0, 780.0


def normal_pdf(x, mu, sig):
    return np.exp(-(x - mu) ** 2 / 2 / sig ** 2) / np.sqrt(2 * np.pi)


def normal_cdf(x, mu, sig):
    return 0.5 * (1 + erf((x - mu) / (sig * np.sqrt(2))))


def truncated_mean(mu, sig, a, b):
    Z = normal_cdf(b, mu, sig) - normal_cdf(a, mu, sig)
    return mu + (normal_pdf(a, mu, sig) - normal_pdf(b, mu, sig)) / Z * sig


def truncated_std(mu, sig, a, b):
    Z = normal_cdf(b, mu, sig) - normal_cdf(a, mu, sig)
    pa, pb = normal_pdf(a, mu, sig), normal_pdf(b, mu, sig)
    return np.sqrt(sig ** 2 * (1 + ((a - mu) / sig * pa - (b - mu) / sig * pb) / Z - ((pa - pb) / Z) ** 2))


def test_wavelength_range_is_the_references():
    assert tuple(ot.global_options.wavelength_range) == (WL0, WL1)


def test_single_line_and_uniform_spectra():
    wl = ot.LightSpectrum("Monochromatic", wl=412.589).random_wavelengths(1000)
    assert wl.shape == (1000,) and np.all(wl == np.float32(412.589))
    wl = ot.LightSpectrum("Constant").random_wavelengths(100_000)
    assert np.mean(wl) == pytest.approx((WL0 + WL1) / 2, abs=0.001)
    assert np.var(wl) == pytest.approx((WL1 - WL0) ** 2 / 12, abs=0.005)
    a, b = 572.5986, 752.69
    wl = ot.LightSpectrum("Rectangle", wl0=a, wl1=b).random_wavelengths(1_000_000)
    assert wl.min() == pytest.approx(a, abs=0.001) and wl.max() == pytest.approx(b, abs=0.001)
    assert np.mean(wl) == pytest.approx((a + b) / 2, abs=0.001)
    assert np.var(wl) == pytest.approx((b - a) ** 2 / 12, abs=0.005)


def test_line_spectrum():
    lines = np.array(ot.presets.spectral_lines.F_eC_)
    vals = np.array([0.2, 1., 3])
    wl = ot.LightSpectrum("Lines", lines=lines, line_vals=vals).random_wavelengths(100_000)
    assert np.all(np.any(np.abs(wl[:, None] - lines) < 1000 * np.finfo(np.float32).eps, axis=1)), "only the lines occur"
    mean = np.sum(lines * vals) / vals.sum()
    assert np.mean(wl) == pytest.approx(mean, abs=0.005)
    assert np.std(wl) == pytest.approx(np.sqrt(np.sum((lines - mean) ** 2 * vals / vals.sum())), abs=0.005)


@pytest.mark.parametrize("mu,sig", [(572.568, 35.123), (732.568, 65.123), (392.968, 25.123), (580.128, 190.123)])
def test_gaussian_spectra_inside_and_cut_by_the_visible_range(mu, sig):
    wl = ot.LightSpectrum("Gaussian", mu=mu, sig=sig).random_wavelengths(100_000)
    if (mu, sig) == (572.568, 35.123):  # (not truncated: the reference's bound is tighter there)
        assert np.mean(wl) == pytest.approx(mu, abs=0.003) and np.std(wl) == pytest.approx(sig, abs=0.003)
    assert np.mean(wl) == pytest.approx(truncated_mean(mu, sig, WL0, WL1), abs=0.005)
    assert np.std(wl) == pytest.approx(truncated_std(mu, sig, WL0, WL1), abs=0.005)


def test_function_and_data_spectra():
    mu, sig = 392.968, 25.123
    wl = ot.LightSpectrum("Function", func=lambda x: normal_pdf(x, mu, sig)).random_wavelengths(100_000)
    assert np.mean(wl) == pytest.approx(truncated_mean(mu, sig, WL0, WL1), abs=0.005)
    assert np.std(wl) == pytest.approx(truncated_std(mu, sig, WL0, WL1), abs=0.005)
    mu, sig = 580.128, 190.123
    grid = wavelengths(1000)
    wl = ot.LightSpectrum("Data", wls=grid, vals=normal_pdf(grid, mu, sig)).random_wavelengths(100_000)
    assert np.mean(wl) == pytest.approx(truncated_mean(mu, sig, WL0, WL1), abs=0.005)
    assert np.std(wl) == pytest.approx(truncated_std(mu, sig, WL0, WL1), abs=0.005)
    a, b = 400, 600  # data over a part of the range only
    grid = np.linspace(a, b, 1000)
    spec = ot.LightSpectrum("Data", wls=grid, vals=normal_pdf(grid, mu, sig))
    wl = spec.random_wavelengths(100_000)
    assert np.mean(wl) == pytest.approx(truncated_mean(mu, sig, a, b), abs=0.005)
    assert np.std(wl) == pytest.approx(truncated_std(mu, sig, a, b), abs=0.005)
    assert np.all(spec(wl[(wl < a) | (wl > b)]) == 0)


def test_rendered_spectra():
    """tests/test_spectrum.py:365-435."""
    rng = np.random.default_rng(4)
    wl, w = rng.uniform(WL0, WL1, 10_000), rng.uniform(0, 1, 10_000)
    spec = ot.LightSpectrum.render(wl, w, desc="ABC")
    assert isinstance(spec, ot.LightSpectrum) and spec.desc == "ABC"
    assert np.sum(spec._vals) * (spec._wls[1] - spec._wls[0]) == pytest.approx(w.sum(), rel=1e-6)  # W/nm -> W
    one = ot.LightSpectrum.render(np.array([500.]), np.array([1.]))
    assert one._wls[0] != one._wls[-1], "a single wavelength gets a range"
    assert ot.LightSpectrum.render(np.array([WL0]), np.array([1.]))._wls[0] >= WL0
    assert ot.LightSpectrum.render(np.array([WL1]), np.array([1.]))._wls[-1] <= WL1
    empty = ot.LightSpectrum.render(np.array([]), np.array([]))
    assert len(empty._wls) > 1 and len(empty._wls) == len(empty._vals) + 1 and not empty._vals.any()
    line = ot.presets.spectral_lines.d
    for N in [100, 10_000, 1_000_000]:
        spec = ot.LightSpectrum.render(np.full(N, line), np.ones(N))
        lit = np.flatnonzero(spec._vals > 0)
        assert len(lit) == 1
        assert spec._wls[lit[0]] + (spec._wls[1] - spec._wls[0]) / 2 == pytest.approx(line, abs=1e-4)
    d65 = ot.presets.light_spectrum.d65
    for N in [3000, 30_000, 3_000_000]:
        wl = WL0 + (WL1 - WL0) * (np.arange(N) + rng.random(N)) / N  # stratified over the range
        spec = ot.LightSpectrum.render(wl, d65(wl))
        delta = 0.02 / np.sqrt(N / 30_000)  # noise and quantisation, ~ 1 / sqrt(N)
        centres = spec._wls[:-1] + (spec._wls[1] - spec._wls[0]) / 2
        m = spec._vals / d65(centres)
        assert np.std(m) / np.mean(m) < delta / 2
        grid = np.linspace(WL0, WL1, 1000)[1:-1]
        m = spec(grid) / d65(grid)
        assert np.std(m) / np.mean(m) < delta
        h = (spec._wls[1] - spec._wls[0]) / 2
        for at in (spec._wls[:-1] + h, spec._wls[:-1] + 1e-7, spec._wls[1:] - 1e-7):  # constant inside a bin
            assert np.abs(spec(at) - spec._vals).max() == pytest.approx(0, abs=1e-7 * spec._vals.max())
