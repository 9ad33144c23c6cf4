"""GPU parity tests: the HIP path (through the C-ABI, via the optrace_amd Python API) against the golden
vectors of the reference and against the CPU oracle on the same inputs.

Bars (BASELINE.json north_star): hit masks / indices / counters bit-exact; positions and directions within
1e-6 relative (checked much tighter); binned XYZ irradiance within 1e-4 in image norm.
"""
import numpy as np
import pytest

import optrace_amd as ot
from optrace_amd import _capi
from optrace_amd.scene import CompiledScene

import oracle_bridge as ob
import scenes
from helpers import load, assert_close, sparse_to_dense, image_rel_l1

pytestmark = pytest.mark.gpu

SURFACE_NAMES = ["circle", "ring", "rect", "rect_rot", "slit", "slit_rot", "sphere_pos", "sphere_neg",
                 "conic_m025", "conic_m75", "conic_p3", "conic_parab", "asphere_a", "asphere_b"]


@pytest.fixture(scope="module")
def zoo():
    with ot.global_options.no_warnings():
        return scenes.surface_zoo(ot)


@pytest.fixture(scope="module")
def leaf():
    return load("leaf_surfaces.npz")


@pytest.fixture(scope="module")
def media():
    return load("leaf_media.npz")


def test_library_is_loaded_from_tree():
    lib = _capi.load_library()
    assert lib.ot_device_count() >= 1
    assert str(_capi.library_path()).endswith("optrace_amd/csrc/liboptrace_hip.so")


@pytest.mark.parametrize("name", SURFACE_NAMES)
def test_find_hit(zoo, leaf, name):
    sf = zoo[name]
    ph, hit, ill = sf.find_hit(leaf[f"{name}/p"], leaf[f"{name}/s"])
    assert np.array_equal(hit, leaf[f"{name}/is_hit"]), "hit mask must be bit-exact"
    if len(ill):
        assert np.array_equal(ill, leaf[f"{name}/ill"])
    else:
        assert not leaf[f"{name}/ill"].any()
    assert_close(ph, leaf[f"{name}/p_hit"], rtol=1e-12, atol=1e-12, what=f"{name} p_hit")
    # against the oracle as well (same inputs)
    ph_o, hit_o, ill_o, st = ob.find_hit(sf._desc(), leaf[f"{name}/p"], leaf[f"{name}/s"])
    assert np.array_equal(hit, hit_o)
    assert_close(ph, ph_o, rtol=1e-12, atol=1e-12, what=f"{name} p_hit vs oracle")


@pytest.mark.parametrize("name", SURFACE_NAMES)
def test_mask_values_normals(zoo, leaf, name):
    sf = zoo[name]
    x, y = leaf[f"{name}/x"], leaf[f"{name}/y"]
    assert np.array_equal(sf.mask(x, y), leaf[f"{name}/mask"])
    assert_close(sf.values(x, y), leaf[f"{name}/values"], rtol=1e-14, atol=1e-15, what=f"{name} values")
    # x/r instead of cos(atan2) in the conic normals: agreement to a few ulp
    assert_close(sf.normals(x, y), leaf[f"{name}/normals"], rtol=1e-12, atol=1e-14, what=f"{name} normals")


@pytest.mark.parametrize("name", ["ring", "slit", "slit_rot"])
def test_hurb_props(zoo, leaf, name):
    a_, b_, b, inside = zoo[name].hurb_props(leaf[f"{name}/x"], leaf[f"{name}/y"])
    assert np.array_equal(inside, leaf[f"{name}/hurb_inside"])
    assert_close(a_, leaf[f"{name}/hurb_a"], rtol=1e-13, atol=1e-15, what="a_")
    assert_close(b_, leaf[f"{name}/hurb_b"], rtol=1e-13, atol=1e-15, what="b_")
    assert_close(b, leaf[f"{name}/hurb_bvec"], rtol=1e-12, atol=1e-15, what="b")


@pytest.mark.parametrize("name", list(scenes.MEDIA.keys()))
def test_refraction_index(media, name):
    ri = ot.RefractionIndex(name.split("_")[0], **scenes.MEDIA[name])
    n = ri(media["wl"])
    exact = name in ("Constant", "Abbe", "Abbe_lines", "Data", "Sellmeier1", "Sellmeier3", "Sellmeier4",
                     "Sellmeier5", "Handbook of Optics 1", "Handbook of Optics 2")
    assert_close(n, media[f"n/{name}"], rtol=4e-16 if exact else 1e-13, what=name)


def test_sphere_projection(media):
    for R in (-13.4, 7.0):
        sf = ot.SphericalSurface(r=abs(R) * 0.6, R=R)
        sf.move_to([0.3, -0.2, 5.0])
        p = media[f"proj/{R}/p"]
        for m in sf.sphere_projection_methods:
            assert_close(sf.sphere_projection(p, m), media[f"proj/{R}/{m}"], rtol=1e-11, atol=1e-13, what=f"{R} {m}")


# ---------------------------------------------------------------------------------------------------------
TRACES = list(scenes.SCENES.keys()) + ["double_gauss_nopol", "asphere_nopol"] + list(scenes.SCENES2.keys()) + list(scenes.SCENES3.keys())
ALL_SCENES = {**scenes.SCENES, **scenes.SCENES2, **scenes.SCENES3}


def build(name):
    no_pol = name.endswith("_nopol")
    base = name[:-6] if no_pol else name
    with ot.global_options.no_warnings():
        return ALL_SCENES[base][0](ot, **({"no_pol": True} if no_pol else {}))


def gpu_trace(name):
    g = load(f"trace_{name}.npz")
    RT = build(name)
    hn = g["hurb_normals"] if "hurb_normals" in g else None
    init = (g["p0"], g["s0"], g["pol0"] if not RT.no_pol else None, g["w0"], g["wl"])
    with ot.global_options.no_warnings():
        RT.trace(int(g["N"]), _initial_rays=init, _hurb_normals=hn, _N_list=g["N_list"])
    assert not RT.geometry_error
    return g, RT


@pytest.mark.parametrize("name", TRACES)
def test_trace_matches_reference(name):
    g, RT = gpu_trace(name)
    r = RT.rays
    assert r.p_list.shape == g["p_list"].shape
    assert r.p_list.flags.f_contiguous and r.p_list.dtype == np.float64
    assert r.w_list.dtype == np.float32 and r.wl_list.dtype == np.float32 and r.n_list.dtype == np.float64
    assert np.array_equal(RT._msgs, g["msgs"]), f"counters differ:\n{RT._msgs}\n{g['msgs']}"
    assert np.array_equal(r.w_list > 0, g["w_list"] > 0), "alive masks per section must be bit-exact"
    tab = name in ("freeform", "masked")  # FunctionSurface2D carried as a spline table: bounded by the tabulation residual
    assert_close(r.p_list, g["p_list"], rtol=1e-11, atol=1e-7 if tab else 1e-11, what="p_list")
    assert_close(r.n_list, g["n_list"], rtol=1e-13, what="n_list")
    loose = name.startswith(("asphere", "mixed", "freeform", "masked"))
    assert_close(r.w_list, g["w_list"], rtol=1e-6 if loose else 2e-7, atol=1e-15 if loose else 1e-30, what="w_list")
    assert_close(r.s0_list, g["s_final"], rtol=1e-10, atol=1e-8 if tab else 1e-12, what="s_final")
    if not RT.no_pol:
        assert r.pol_list.dtype == np.float32
        assert_close(r.pol_list, g["pol_list"], rtol=1e-5, atol=2e-7, what="pol_list")


@pytest.mark.parametrize("name", ["double_gauss", "asphere", "hurb_slit_lens"])
def test_trace_matches_oracle(name):
    """Same injected rays through the oracle: masks and counters bit-exact, geometry to rounding."""
    g, RT = gpu_trace(name)
    sc = CompiledScene(RT)
    rays = ob.HostRays(int(g["N"]), sc.nt, RT.no_pol)
    rays.set_initial(g["p0"], g["s0"], g["pol0"], g["w0"], g["wl"])
    msgs, st = ob.trace(sc.desc, rays, g["hurb_normals"] if "hurb_normals" in g else None)
    assert st == 0
    assert np.array_equal(msgs, RT._msgs)
    assert np.array_equal(rays.w_list > 0, RT.rays.w_list > 0)
    assert_close(RT.rays.p_list, rays.p_list, rtol=1e-11, atol=1e-11, what="p_list")
    assert_close(RT.rays.w_list, rays.w_list, rtol=1e-6, atol=1e-15, what="w_list")


@pytest.mark.parametrize("name", ["c1_single_lens", "double_gauss", "mixed_geometry", "arizona_eye", "asphere"])
def test_detector_image_matches_reference(name):
    g, RT = gpu_trace(name)
    with ot.global_options.no_warnings():
        for di, det in enumerate(RT.detectors):
            projs = [None] if not isinstance(det.surface, ot.SphericalSurface) else \
                ["Equidistant", "Orthographic", "Equal-Area", "Stereographic"]
            for proj in projs:
                key = f"det{di}/{proj}"
                ph, hw, wl, ext, projection, ill = RT._hit_detector("x", di, None, None, proj)
                n = hw.shape[0]
                hw_h = hw.cpu().numpy()
                ph_h = ph.cpu().numpy().reshape(3, n).T
                sel = hw_h > 0
                assert np.count_nonzero(sel) == g[f"{key}/w"].shape[0], "number of detector hits must be exact"
                if name.startswith(("asphere", "mixed")):  # float32 exp / tabulated Function filter upstream
                    assert_close(hw_h[sel], g[f"{key}/w"], rtol=1e-6, atol=1e-15, what="hit weights")
                else:
                    assert np.array_equal(hw_h[sel], g[f"{key}/w"])
                assert_close(ph_h[sel], g[f"{key}/ph"], rtol=1e-9, atol=1e-11, what=f"{key} ph")
                assert ill == int(g[f"{key}/ill"])
                if np.any(sel):
                    assert_close(ext, g[f"{key}/extent"], rtol=1e-9, atol=1e-11, what="auto extent")
                img = RT.detector_image(detector_index=di, projection_method=proj)
                ref = sparse_to_dense(g, f"{key}/img")
                assert img._data.shape == ref.shape
                assert_close(img.extent, g[f"{key}/img/extent"], rtol=1e-9, atol=1e-11, what="image extent")
                pw = float(g[f"{key}/img/power"])
                assert abs(img.power() - pw) <= 1e-6 * pw
                # rays sitting within rounding of a pixel edge may land in the neighbour pixel: image norm
                assert np.all(image_rel_l1(img._data, ref) < 1e-4), image_rel_l1(img._data, ref)
            # user extent + single source
            key = f"det{di}/user"
            img = RT.detector_image(detector_index=di, extent=list(g[f"{key}/uext"]), projection_method=projs[0],
                                    source_index=len(RT.ray_sources) - 1)
            ref = sparse_to_dense(g, f"{key}/img")
            pw = float(g[f"{key}/img/power"])
            assert abs(img.power() - pw) <= 1e-6 * max(pw, 1e-300)
            if pw > 0:
                assert np.all(image_rel_l1(img._data, ref) < 1e-4)


def test_render_matches_oracle_bitwise_layout():
    """ot_render_accumulate against the oracle's sequential histogram on random hits."""
    rng = np.random.default_rng(3)
    n = 50000
    p = np.zeros((n, 3))
    p[:, 0] = rng.uniform(-1.2, 2.2, n)
    p[:, 1] = rng.uniform(0.4, 1.6, n)
    w = rng.uniform(0.1, 1, n).astype(np.float32)
    w[::7] = 0
    wl = rng.uniform(350, 840, n).astype(np.float32)
    img = ot.RenderImage(extent=[-1.0, 2.0, 0.5, 1.5])
    img.render(p, w, wl)
    Ny, Nx = img._data.shape[:2]
    ref = ob.render(p[:, 0], p[:, 1], w, wl, img.extent, Nx, Ny)
    assert (Nx, Ny) == (945 * 3, 945)
    assert_close(img._data, ref, rtol=1e-12, atol=1e-18, what="histogram")
    assert abs(img.power() - ref[..., 3].sum()) < 1e-9 * ref[..., 3].sum()


# ---------------------------------------------------------------------------------------------------------
def test_full_size_properties_double_gauss():
    """Size-independent checks at a bench-like ray count (1M rays, on-device generation)."""
    with ot.global_options.no_warnings():
        RT = scenes.double_gauss(ot, seed=11)
        N = 1_000_000
        RT.trace(N)
        r = RT.rays
        w = r.w_list
        # weights never increase along a ray; dead rays stay dead and keep their position
        assert np.all(np.diff(w.astype(np.float64), axis=1) <= 1e-12)
        dead = w[:, 1:] == 0
        was_dead = w[:, :-1] == 0
        p = r.p_list
        stay = was_dead & dead
        assert np.array_equal(p[:, 1:][stay], p[:, :-1][stay])
        # counters equal the number of rays that lose their power at that section for these reasons
        lost = (w[:, :-1] > 0) & (w[:, 1:] == 0)
        assert lost.sum() == N, "every ray ends absorbed (end aperture at the latest)"
        # z never decreases for living rays
        alive = w[:, :-1] > 0
        dz = np.diff(p[:, :, 2], axis=1)
        assert np.all(dz[alive] >= -1e-9)
        # total power of the source: sum of the initial weights = sum of source powers
        assert abs(w[:, 0].astype(np.float64).sum() - 5.0) < 1e-3
        # directions are unit vectors where finite
        s = r.s0_list
        nrm = np.linalg.norm(s, axis=1)
        assert np.all(np.abs(nrm[np.isfinite(nrm)] - 1) < 1e-12)
        # polarisation is perpendicular to the ray direction in the last living section
        img = RT.detector_image()
        assert img.power() <= 5.0 and img.power() > 0.1
        # same seed -> identical rays; different seed -> different rays
        RT2 = scenes.double_gauss(ot, seed=11)
        RT2.trace(100000)
        RT3 = scenes.double_gauss(ot, seed=11)
        RT3.trace(100000)
        assert np.array_equal(RT2.rays.p_list, RT3.rays.p_list)
        RT4 = scenes.double_gauss(ot, seed=12)
        RT4.trace(100000)
        assert not np.array_equal(RT2.rays.p_list, RT4.rays.p_list)


@pytest.mark.parametrize("name,no_pol", [("double_gauss", False), ("double_gauss", True), ("c1_single_lens", False),
                                          ("mixed_lines", False)])
def test_discrete_spectrum_tables_equal_formula_path(name, no_pol):
    """Scenes whose sources are all discrete run the SPEC=2 kernel (n, n1/n2, filter T per line from LDS) in
    ot_generate_and_trace.  It must give bit-identical rays to generating the same rays (same seed) with
    ot_rays_generate and tracing them with the formula kernel of ot_trace, which the golden tests pin."""
    import ctypes as C
    import torch
    from optrace_amd._device import ptr, stream_ptr
    lib = _capi.load_library()
    N = 300_000
    with ot.global_options.no_warnings():
        if name == "mixed_lines":  # filters, ideal lens, Function index, conics: everything per line
            RT = scenes.mixed_geometry(ot, seed=77)
            RT.ray_sources[1].spectrum = ot.LightSpectrum("Lines", lines=[450., 550., 610., 680.], line_vals=[1, 2, 1, 0.5])
        else:
            RT = scenes.SCENES[name][0](ot, no_pol=no_pol, seed=77)
        RT.trace(N)  # fused: generation + LINES tables
        assert RT._scene.desc.n_lines >= 1
        fused = {k: RT.rays._dev[k].clone() for k in ("p", "s", "w", "n", "wl") + (() if no_pol else ("pol",))}
        msgs_fused = RT._msgs.copy()
        # same rays again: generate section 0 only, then the injected-ray kernel (formulas, IEEE n1/n2 per ray)
        rays = RT.rays._rays_struct()
        tab, rng = RT.rays._source_table(), RT.rays._source_ranges()
        _capi.check(lib.ot_rays_generate(tab.handle, rng, len(rng), 77, int(no_pol), C.byref(rays), stream_ptr()))
        msgs = torch.zeros(5 * RT.rays.Nt + 1, dtype=torch.int64, device="cuda")
        _capi.check(lib.ot_trace(RT._scene_handle, C.byref(rays), None, 77, ptr(msgs), stream_ptr()))
        torch.cuda.synchronize()
    assert np.array_equal(msgs.cpu().numpy()[:-1].reshape(5, -1), msgs_fused)
    for k, t in fused.items():
        a, b = t.cpu().numpy(), RT.rays._dev[k].cpu().numpy()
        if not np.array_equal(a, b, equal_nan=True):
            bad = np.nonzero(~((a == b) | (np.isnan(a) & np.isnan(b))))[0]
            raise AssertionError(f"{k} differs between the LINES and the formula kernel: {bad.shape[0]} of {a.shape[0]} "
                                 f"entries, first at {bad[:8]}, values {a[bad[:4]]} vs {b[bad[:4]]}")
