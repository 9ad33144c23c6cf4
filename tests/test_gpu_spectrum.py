"""GPU parity of detector_spectrum / source_spectrum / source_image against the
reference's fixtures (tests/golden/spectra.npz) and the oracle restatement."""
import pathlib
import sys

import numpy as np
import pytest
import torch

import optrace_amd as ot
import scenes
from helpers import load, assert_close, sparse_to_dense, image_rel_l1
from test_gpu_parity import gpu_trace, TRACES

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from oracle import spectrum as ospec  # noqa: E402

pytestmark = pytest.mark.gpu


def check_spec(spec, sp, key):
    assert spec.spectrum_type == "Histogram" and spec.unit == "W/nm"
    assert spec._wls.dtype == np.float64 and spec._vals.dtype == np.float64
    assert_close(spec._wls, sp[f"{key}/wls"], rtol=0, atol=0, what=f"{key} bin edges")
    ref = sp[f"{key}/vals"]
    tot = ref.sum()
    # bin membership is exact; the reference sums float32 block-wise, the kernel in float64
    assert_close(spec._vals, ref, rtol=1e-6, atol=1e-6 * tot, what=f"{key} vals")


@pytest.mark.parametrize("name", [t for t in TRACES if t not in scenes.SCENES2 and t not in scenes.SCENES3])
def test_spectra_and_source_images_match_reference(name):
    g, RT = gpu_trace(name)
    sp = load("spectra.npz")
    loose = name.startswith(("asphere", "mixed"))
    with ot.global_options.no_warnings():
        for di in range(len(RT.detectors)):
            check_spec(RT.detector_spectrum(detector_index=di), sp, f"{name}/det{di}/all")
            uext = list(sp[f"{name}/det{di}/uext"])
            spec = RT.detector_spectrum(detector_index=di, extent=uext, source_index=len(RT.ray_sources) - 1)
            check_spec(spec, sp, f"{name}/det{di}/user")
        for si in range(len(RT.ray_sources)):
            check_spec(RT.source_spectrum(si), sp, f"{name}/src{si}")
            img = RT.source_image(si)
            key = f"{name}/src{si}/img"
            ref = sparse_to_dense(sp, key)
            assert img._data.shape == ref.shape
            assert_close(img.extent, sp[f"{key}/extent"], rtol=1e-12, atol=1e-12, what="source image extent")
            pw = float(sp[f"{key}/power"])
            assert abs(img.power() - pw) <= 1e-6 * pw
            assert np.all(image_rel_l1(img._data, ref) < (1e-4 if loose else 1e-6)), image_rel_l1(img._data, ref)
    assert "Spectrum" in RT.source_spectrum(0).long_desc and "RS0" in RT.source_image(0).long_desc


def test_big_bundle_bin_count():
    """40000 injected rays: bin count grows with sqrt(N) and every ray keeps its bin."""
    sp = load("spectra.npz")
    with ot.global_options.no_warnings():
        RT = scenes.mixed_geometry(ot, no_pol=True)
        init = (sp["big/p0"], sp["big/s0"].astype(np.float64), None, sp["big/w0"], sp["big/wl"])
        RT.trace(int(sp["big/p0"].shape[0]), _initial_rays=init, _N_list=sp["big/N_list"])
        assert np.array_equal(RT.rays.N_list, sp["big/N_list"])
        for si in range(2):
            spec = RT.source_spectrum(si)
            assert spec._vals.shape[0] > 51
            check_spec(spec, sp, f"big/src{si}")
        for di in range(len(RT.detectors)):
            check_spec(RT.detector_spectrum(detector_index=di), sp, f"big/det{di}/all")
            spec = RT.detector_spectrum(detector_index=di, extent=list(sp[f"big/det{di}/uext"]), source_index=1)
            check_spec(spec, sp, f"big/det{di}/user")


@pytest.mark.parametrize("n,kind", [(0, "empty"), (1, "single"), (5000, "mono"), (200_000, "wide"), (3_000_000, "huge")])
def test_render_against_oracle(n, kind):
    """LightSpectrum.render on raw arrays vs. the oracle loop; dense zero weights are skipped."""
    rng = np.random.default_rng(n + 5)
    if kind == "mono":
        wl = np.full(n, 532.25, dtype=np.float32)
    else:
        wl = rng.uniform(400, 700, n).astype(np.float32)
    w = rng.uniform(0.1, 1, n).astype(np.float32)
    drop = rng.random(n) < 0.3
    w[drop] = 0
    spec = ot.LightSpectrum.render(wl, w)
    sel = w > 0
    if kind == "huge":  # oracle loop is O(n) Python: check against NumPy's own histogram there instead
        N = ospec.bin_count(w[sel])
        vals, edges = np.histogram(wl[sel], bins=N, weights=w[sel].astype(np.float64),
                                   range=[wl[sel].min(), wl[sel].max()])
        vals = vals / (float(edges[1]) - float(edges[0]))
        wls = edges
        assert N > 51
    else:
        wls, vals = ospec.render(wl[sel], w[sel])
    assert_close(spec._wls, wls, rtol=0, atol=0, what="edges")
    assert_close(spec._vals, vals, rtol=1e-12, atol=1e-12 * max(vals.sum(), 1e-300), what="vals")
    if n:
        assert abs(spec._vals.sum() * (spec._wls[1] - spec._wls[0]) - w.astype(np.float64).sum()) <= 1e-9 * w.sum()


def test_render_global_atomic_path_and_errors():
    """More bins than fit into LDS (ot_spectrum_histogram falls back to global atomics)."""
    from optrace_amd import _capi
    from optrace_amd._device import ptr, stream_ptr
    lib = _capi.load_library()
    rng = np.random.default_rng(11)
    n, nb = 400_000, 9001
    wl = rng.uniform(380, 780, n).astype(np.float32)
    w = rng.uniform(0, 1, n).astype(np.float32)
    edges = np.linspace(np.float32(380), np.float32(780), nb + 1, dtype=np.float32)
    d = [torch.from_numpy(a).cuda() for a in (wl, w, edges)]
    hist = torch.zeros(nb, dtype=torch.float64, device="cuda")
    _capi.check(lib.ot_spectrum_histogram(n, ptr(d[0]), ptr(d[1]), ptr(d[2]), nb, ptr(hist), stream_ptr()))
    ref, _ = np.histogram(wl, bins=nb, weights=w.astype(np.float64), range=[np.float32(380), np.float32(780)])
    assert_close(hist.cpu().numpy(), ref, rtol=1e-12, atol=1e-12, what="global path")
    assert lib.ot_spectrum_histogram(n, ptr(d[0]), ptr(d[1]), ptr(d[2]), 0, ptr(hist), stream_ptr()) != 0
    assert lib.ot_spectrum_range(-1, ptr(d[0]), ptr(d[1]), ptr(hist), ptr(hist), stream_ptr()) != 0


def test_source_checks():
    with ot.global_options.no_warnings():
        RT = scenes.c1_single_lens(ot)
        with pytest.raises(RuntimeError):
            RT.source_image(0)  # nothing traced
        RT.trace(2000)
        with pytest.raises(IndexError):
            RT.source_spectrum(5)
        RT.lenses[0].move_to([0, 0, RT.lenses[0].pos[2] + 0.5])
        with pytest.raises(RuntimeError):
            RT.source_image(0)  # geometry changed


@pytest.mark.parametrize("N", [5000, 2_300_000])
def test_detector_spectrum_through_a_compact_hit_list_equals_the_dense_one(N):
    """`detector_spectrum` on long bundles takes the valid hits' weights and wavelengths from a compact list
    (`ot_detector_req.fill` without positions, `ot_spectrum_*_compact`): same bins, same edges, sums to 1e-12 -- with an
    automatic and a user extent, and for one source alone."""
    with ot.global_options.no_warnings():
        RT = scenes.double_gauss(ot, seed=11)
        RT.trace(N)
        old = RT.COMPACT_HITS_FROM
        try:
            e = RT.detector_image().extent
            cx, cy, hx, hy = (e[0] + e[1]) / 2, (e[2] + e[3]) / 2, (e[1] - e[0]) / 2, (e[3] - e[2]) / 2
            part = [cx - 0.6 * hx, cx + 0.3 * hx, cy - 0.2 * hy, cy + 0.7 * hy]  # an off-centre part of the lit area
            for kw in (dict(), dict(extent=part), dict(source_index=0)):
                type(RT).COMPACT_HITS_FROM = 1 << 60
                dense = RT.detector_spectrum(**kw)
                type(RT).COMPACT_HITS_FROM = 1
                compact = RT.detector_spectrum(**kw)
                np.testing.assert_array_equal(dense._wls, compact._wls)
                assert dense._vals.sum() > 0
                np.testing.assert_allclose(compact._vals, dense._vals, rtol=1e-12, atol=1e-15 * dense._vals.max())
        finally:
            type(RT).COMPACT_HITS_FROM = old
