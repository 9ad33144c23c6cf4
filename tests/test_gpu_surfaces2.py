"""GPU parity of the tilted / data / function surfaces: leaf operators against the reference's
fixtures (tests/golden/leaf_surfaces2.npz) and the oracle, scene traces in test_gpu_parity (prism, freeform)."""
import numpy as np
import pytest

import optrace_amd as ot

import oracle_bridge as ob
import scenes
from helpers import load, assert_close
from test_gpu_parity import gpu_trace

pytestmark = pytest.mark.gpu

NAMES = ["tilted", "tilted_sph", "data1d", "data1d_flip", "data2d", "data2d_rot_flip", "func1d", "func2d",
         "func2d_rot", "func2d_noderiv"]
EXACT = [n for n in NAMES if not n.startswith("func")]


@pytest.fixture(scope="module")
def zoo():
    with ot.global_options.no_warnings():
        return scenes.surface_zoo2(ot)


@pytest.fixture(scope="module")
def leaf():
    return load("leaf_surfaces2.npz")


@pytest.mark.parametrize("name", NAMES)
def test_find_hit(zoo, leaf, name):
    sf = zoo[name]
    ph, hit, ill = sf.find_hit(leaf[f"{name}/p"], leaf[f"{name}/s"])
    ph_o, hit_o, ill_o, st = ob.find_hit(sf._desc(), leaf[f"{name}/p"], leaf[f"{name}/s"])
    assert st == 0
    # device against the oracle: same tables, same algorithm
    assert np.array_equal(hit, hit_o) and np.array_equal(ill, ill_o)
    assert_close(ph, ph_o, rtol=1e-11, atol=1e-11, what=f"{name} p_hit vs oracle")
    if name in EXACT:
        assert np.array_equal(hit, leaf[f"{name}/is_hit"]), "hit mask must be bit-exact"
        assert np.array_equal(ill, leaf[f"{name}/ill"])
        assert_close(ph, leaf[f"{name}/p_hit"], rtol=1e-11, atol=1e-11, what=f"{name} p_hit")
    else:
        assert np.count_nonzero(hit != leaf[f"{name}/is_hit"]) <= 2
        same = hit == leaf[f"{name}/is_hit"]
        assert_close(ph[same], leaf[f"{name}/p_hit"][same], rtol=0, atol=2e-8, what=f"{name} p_hit")


@pytest.mark.parametrize("name", NAMES)
def test_mask_values_normals(zoo, leaf, name):
    sf = zoo[name]
    x, y = leaf[f"{name}/x"], leaf[f"{name}/y"]
    assert np.array_equal(sf.mask(x, y), leaf[f"{name}/mask"])
    sd = sf._desc()
    assert_close(sf.values(x, y), ob.values(sd, x, y), rtol=1e-13, atol=1e-14, what=f"{name} values vs oracle")
    assert_close(sf.normals(x, y), ob.normals(sd, x, y), rtol=1e-11, atol=1e-13, what=f"{name} normals vs oracle")
    if name in EXACT:
        assert_close(sf.values(x, y), leaf[f"{name}/values"], rtol=1e-13, atol=1e-14, what=f"{name} values")
        assert_close(sf.normals(x, y), leaf[f"{name}/normals"], rtol=1e-11, atol=1e-13, what=f"{name} normals")
    else:
        assert_close(sf.values(x, y), leaf[f"{name}/values"], rtol=0, atol=3e-9, what=f"{name} values")
        atol = 1e-6 if name == "func2d_noderiv" else 2e-7
        assert_close(sf.normals(x, y), leaf[f"{name}/normals"], rtol=0, atol=atol, what=f"{name} normals")


@pytest.mark.parametrize("name", ["prism", "freeform"])
def test_detector_images(name):
    from helpers import sparse_to_dense, image_rel_l1
    g, RT = gpu_trace(name)
    with ot.global_options.no_warnings():
        for di in range(len(RT.detectors)):
            key = f"det{di}/None"
            ph, hw, wl, ext, projection, ill = RT._hit_detector("x", di, None, None, None)
            sel = hw.cpu().numpy() > 0
            assert np.count_nonzero(sel) == g[f"{key}/w"].shape[0]
            img = RT.detector_image(detector_index=di)
            ref = sparse_to_dense(g, f"{key}/img")
            pw = float(g[f"{key}/img/power"])
            assert abs(img.power() - pw) <= 1e-6 * pw
            assert_close(img.extent, g[f"{key}/img/extent"], rtol=1e-7, atol=1e-7, what="extent")
            assert np.all(image_rel_l1(img._data, ref) < (2e-3 if name == "freeform" else 1e-4))


def test_spline_tables_survive_scene_lifetime_and_large_bundle():
    """1 M rays through the freeform scene: finite, energy conserving, and repeatable (tables live in the scene)."""
    with ot.global_options.no_warnings():
        RT = scenes.freeform_scene(ot, seed=5)
        RT.trace(1_000_000)
        w = RT.rays.w_list
        assert np.all(np.isfinite(RT.rays.p_list)) and np.all(w[:, 1:] <= w[:, :-1] + 1e-12)
        assert RT._msgs[RT.INFOS.ILL_COND].sum() < 1e-4 * RT.rays.N  # a few bracket failures at the data grid's rim
        p1 = RT.rays.p_list[:1000].copy()
        RT.trace(1_000_000)
        assert np.array_equal(p1, RT.rays.p_list[:1000])


@pytest.mark.parametrize("R_", [3, 10.2, 100])
def test_one_shape_through_every_surface_type(R_):
    """After the reference's test_same_surface_behavior (tests/test_tracer_special.py:172-249): a paraboloid written as
    conic, asphere (two ways), function surface (two ways) and 1-D / 2-D data surfaces of several resolutions focuses
    a narrow beam at the same place (the reference's tolerance: 0.001 mm between the numeric variants, 0.2 mm to the
    sphere and to the lens-maker value)."""
    r, n = 2, 1.5
    func = lambda x, y, R: 1 / 2 / R * (x ** 2 + y ** 2)  # noqa: E731
    func2 = lambda x, y: 0.78785 + 1 / 2 / R_ * (x ** 2 + y ** 2)  # noqa: E731
    with ot.global_options.no_warnings():
        surfs = [ot.SphericalSurface(R=R_, r=r), ot.ConicSurface(R=R_, k=-1, r=r),
                 ot.FunctionSurface2D(func=func2, r=r),
                 ot.FunctionSurface2D(func=func, r=r, z_min=0, z_max=func(0, r, R_), func_args=dict(R=R_)),
                 ot.AsphericSurface(R=R_, r=r, k=-1, coeff=[0.]), ot.AsphericSurface(R=1e9, r=r, k=-1, coeff=[1 / 2 / R_])]
        for N in (900, 51, 201):
            Y, X = np.mgrid[-r:r:N * 1j, -r:r:N * 1j]
            surfs.append(ot.DataSurface2D(r=r, data=4.657165 + 1 / 2 / R_ * (X ** 2 + Y ** 2)))
            surfs.append(ot.DataSurface1D(r=r, data=4.657165 + 1 / 2 / R_ * np.linspace(0, r, N) ** 2))
        d = 0.1 + func(0, r, R_)
        f_lensmaker = 1 / ((n - 1) * (1 / R_))  # plano-convex: the thickness does not enter the focal length
        for RS_r in (0.01, 0.5):
            RT = ot.Raytracer(outline=[-3, 3, -3, 3, -10, 500], no_pol=True, seed=4)
            RT.add(ot.RaySource(ot.CircularSurface(r=RS_r), spectrum=ot.LightSpectrum("Monochromatic", wl=555),
                                divergence="None", pos=[0, 0, -3]))
            f_list = []
            for surf in surfs:
                L = ot.Lens(surf, ot.CircularSurface(r=r), n=ot.RefractionIndex("Constant", n=n), pos=[0, 0, d / 2], d=d)
                RT.add(L)
                RT.trace(100_000)
                res, _ = RT.focus_search(RT.focus_search_methods[0], 5)
                f_list.append(float(res.x))
                back_z, dv = L.back.pos[2], L.back.pos[2] - L.front.pos[2]
                RT.remove(L)
            # thick plano-convex lens, curved side first: the focus lies f (1 - (n - 1) d / (n R)) behind the flat side
            assert abs(f_list[1] - (back_z + f_lensmaker * (1 - (n - 1) * dv / (n * R_)))) < (0.002 if RS_r < 0.1 else 0.2)
            assert abs(f_list[0] - f_list[1]) < 0.2
            numeric = f_list[1:]
            assert max(numeric) - min(numeric) < 0.001, (RS_r, f_list)


def test_brewster_plate_through_three_surface_types():
    """After the reference's test_tilted_plane_different_surface_types (tests/test_tracer_special.py:318-365): a plate
    tilted at Brewster's angle, written as function, tilted and data surface, leaves every ray straight and shifts it
    sideways by the same 0.0531867 mm."""
    n = ot.RefractionIndex("Constant", n=1.55)
    b_ang = np.arctan(1.55 / 1)
    with ot.global_options.no_warnings():
        RT = ot.Raytracer(outline=[-3, 3, -3, 3, -8, 12], seed=6)
        spectrum = ot.LightSpectrum("Monochromatic", wl=550.)
        for x0, pol in ((0.5, "x"), (0, "y"), (-0.5, "Uniform")):
            RT.add(ot.RaySource(ot.CircularSurface(r=0.05), divergence="None", spectrum=spectrum, pos=[x0, 0, -4],
                                polarization=pol))
        Y, X = np.mgrid[-0.7:0.7:100j, -0.7:0.7:100j]
        surfs = [ot.FunctionSurface2D(func=lambda x, y: np.tan(b_ang) * x, r=0.7),
                 ot.TiltedSurface(r=0.7, normal=[-np.sin(b_ang), 0, np.cos(b_ang)]),
                 ot.DataSurface2D(r=0.7, data=np.tan(b_ang) * Y)]
        for surf in surfs:
            L = ot.Lens(surf, surf, d=0.2, pos=[0, 0, 0.5], n=n)
            RT.add(L)
            RT.trace(10_000)
            RT.remove(L)
            assert not RT.geometry_error
            p, s, _, _, _, _, _ = RT.rays.rays_by_mask(np.ones(RT.rays.N, dtype=bool), ret=[1, 1, 0, 0, 0, 0, 0])
            assert abs(np.mean(s[:, -2, 2]) - 1) < 1e-7                                   # still straight
            np.testing.assert_allclose(p[:, -2, 0] - p[:, -3, 0], -0.0531867, atol=1e-5, rtol=0)  # parallel shift
            # the plane of incidence is xz: x-polarised light is p-polarised and passes a Brewster plate without loss,
            # y-polarised light loses the Fresnel s-reflection at both faces
            w = RT.rays.w_list
            xsrc = slice(int(RT.rays.B_list[0]), int(RT.rays.B_list[1]))
            ysrc = slice(int(RT.rays.B_list[1]), int(RT.rays.B_list[2]))
            np.testing.assert_allclose(w[xsrc, -2], w[xsrc, 0], rtol=1e-5)
            ci, ct = np.cos(b_ang), np.cos(np.arcsin(np.sin(b_ang) / 1.55))
            Rs = ((ci - 1.55 * ct) / (ci + 1.55 * ct)) ** 2
            np.testing.assert_allclose(w[ysrc, -2], w[ysrc, 0] * (1 - Rs) ** 2, rtol=1e-5)


@pytest.mark.parametrize("k", [-1, 0])
@pytest.mark.parametrize("R", [0.1, 10, 10000])
def test_normals_of_numeric_surfaces_across_scales(k, R):
    """After the reference's test_surface_numerical_precision (tests/test_surface.py:704-766): normals of a paraboloid /
    sphere cap written as function surfaces (1e-7) and as data surfaces of 50-401 samples (1e-4), for curvature radii
    from 0.1 to 10 000 mm, offsets down to -10 000 mm and apertures of 0.2 % and 70 % of the radius.  One corner is
    looser than the reference's 1e-7: a 0.4 um wide aperture on an offset of -10 000 mm, where the rounding noise of the
    function itself (2e-12) over the coarsest sampling step limits the tabulated slopes to 5e-7."""
    def n_conic(x, y, rho):
        r = np.sqrt(x ** 2 + y ** 2)
        phi = np.arctan2(y, x)
        n_r = -rho * r / np.sqrt(1 - k * rho ** 2 * r ** 2)
        return np.column_stack((n_r * np.cos(phi), n_r * np.sin(phi), np.sqrt(1 - n_r ** 2)))

    worst = []
    with ot.global_options.no_warnings():
        for z0 in (0, -80, -10000):
            if k == -1:
                surf_f = lambda x, y: z0 + (x ** 2 + y ** 2) / R / 2  # noqa: E731
            else:
                surf_f = lambda x, y: z0 + R - np.sqrt(R ** 2 - (x ** 2 + y ** 2))  # noqa: E731
            for fr in (0.002, 0.7):
                r = fr * R
                x = np.linspace(0, r, 100)
                y = np.zeros_like(x)
                should = n_conic(x, y, 1 / R)
                tol = 5e-7 if (z0 == -10000 and r < 1e-3) else 1e-7
                e2 = np.abs(ot.FunctionSurface2D(func=surf_f, r=r).normals(x, y) - should).max()
                e1 = np.abs(ot.FunctionSurface1D(func=lambda rr: surf_f(rr, np.zeros_like(rr)), r=r).normals(x, y) - should).max()
                if max(e1, e2) > tol:
                    worst.append((z0, fr, e1, e2))
                for N in (50, 51, 400, 401):
                    Y, X = np.mgrid[-r:r:N * 1j, -r:r:N * 1j]
                    d2 = ot.DataSurface2D(data=surf_f(X.flatten(), Y.flatten()).reshape(X.shape), r=r)
                    d1 = ot.DataSurface1D(data=surf_f(np.linspace(0, r, N), np.zeros(N)), r=r)
                    ed = max(np.abs(d2.normals(x, y) - should).max(), np.abs(d1.normals(x, y) - should).max())
                    if ed > 1e-4:
                        worst.append((z0, fr, N, ed))
    assert not worst, worst


def test_hit_finding_properties_on_every_surface():
    """After the reference's test_surface_hit_finding (tests/test_surface.py:237-268), over every surface flavour of the
    device kernels: hits lie on the surface (C_EPS) and on their ray; rays that miss above valid surface coordinates
    end behind the surface; rays that start behind the surface stay where they are and do not hit."""
    rng = np.random.default_rng(11)
    p = rng.uniform(-2, -1, size=(10000, 3))
    s = rng.uniform(-1, 1, size=(10000, 3))
    s /= np.linalg.norm(s, axis=1)[:, None]
    s[:, 2] = np.abs(s[:, 2])
    s[s[:, 2] < 1e-3, 2] = 1e-3  # grazing directions: keep the ray parameter finite
    s /= np.linalg.norm(s, axis=1)[:, None]
    with ot.global_options.no_warnings():
        zoo = {**scenes.surface_zoo(ot), **scenes.surface_zoo2(ot)}
        for name, S in zoo.items():
            C = S.C_EPS
            p_hit, is_hit, _ = S.find_hit(p, s)
            z_hit = S.values(p_hit[is_hit, 0], p_hit[is_hit, 1])
            assert np.allclose(p_hit[is_hit, 2] - z_hit, 0, rtol=0, atol=C), name
            zs = S.values(p_hit[~is_hit, 0], p_hit[~is_hit, 1])
            ms = S.mask(p_hit[~is_hit, 0], p_hit[~is_hit, 1])
            assert np.all(p_hit[~is_hit, 2][ms] > zs[ms] - 1e-12), name
            t = (p_hit[:, 2] - p[:, 2]) / s[:, 2]
            assert np.allclose(p + s * t[:, None] - p_hit, 0, atol=C), name
            behind = p_hit.copy()
            behind[:, 2] = S.z_max + 2
            p2, hit2, _ = S.find_hit(behind, s)
            assert np.allclose(p2 - behind, 0) and not np.any(hit2), name


def _disc_mask(x, y):
    return (x - 0.7) ** 2 + (y + 0.4) ** 2 <= 2.2 ** 2


def _annulus(r):
    return (r >= 0.9) & (r <= 3.1)


def test_mask_func_bitmap_against_the_callable():
    """mask_func travels as a bitmap (include/optrace_amd.h, OT_SURF_FLAG_MASK_TABLE): mask / values / normals / find_hit
    on the device agree with the oracle on the same bitmap everywhere, and with the Python callable
    (function_surface_2d.py:158-191) except within one cell of the mask's edge."""
    rng = np.random.default_rng(77)
    with ot.global_options.no_warnings():
        f2 = ot.FunctionSurface2D(r=4, func=lambda x, y: 0.01 * (x ** 2 + 2 * y ** 2), mask_func=_disc_mask)
        f2.flip()
        f2.rotate(25)
        f2.move_to([0.3, -0.2, 5])
        f1 = ot.FunctionSurface1D(r=4, func=lambda r: 0.02 * r ** 2, mask_func=_annulus)
        f1.move_to([-0.1, 0.25, 3])
    n = 200_000
    for sf in (f2, f1):
        x, y = sf.pos[0] + rng.uniform(-4.5, 4.5, n), sf.pos[1] + rng.uniform(-4.5, 4.5, n)
        sd = sf._desc()
        m = sf.mask(x, y)
        assert np.array_equal(m, ob.mask(sd, x, y)), "same bitmap, same cells"
        callable_mask = sf._mask_host(x, y)
        differ = m != callable_mask
        dx, dy = x - sf.pos[0], y - sf.pos[1]
        if sf is f2:
            cell = 2 * sf.r / sf.N_MASK
            xr, yr = sf._rotate_rc(dx, dy, -sf._angle)
            edge_dist = np.abs(np.hypot(xr - 0.7, sf._sign * yr + 0.4) - 2.2)
        else:
            cell = sf.r / sf.N_MASK_1D
            rr = np.hypot(dx, dy)
            edge_dist = np.minimum(np.abs(rr - 0.9), np.abs(rr - 3.1))
        assert np.all(edge_dist[differ] <= cell), "only positions within one cell of the edge may differ"
        assert differ.mean() < 2e-3 and 0.1 < m.mean() < 0.6
        assert_close(sf.values(x, y), ob.values(sd, x, y), rtol=1e-13, atol=1e-14, what="values vs oracle")
        assert_close(sf.normals(x, y), ob.normals(sd, x, y), rtol=1e-11, atol=1e-13, what="normals vs oracle")
        assert np.all(sf.normals(x, y)[~m] == [0, 0, 1]), "no surface, no slope (function_surface_2d.py:210-214)"
        p = np.column_stack((x, y, np.full(n, sf.pos[2] - 2.0)))
        s = np.column_stack((rng.uniform(-0.05, 0.05, n), rng.uniform(-0.05, 0.05, n), np.ones(n)))
        s /= np.linalg.norm(s, axis=1)[:, None]
        ph, hit, ill = sf.find_hit(p, s)
        ph_o, hit_o, ill_o, st = ob.find_hit(sd, p, s)
        assert st == 0 and np.array_equal(hit, hit_o) and np.array_equal(ill, ill_o)
        assert_close(ph, ph_o, rtol=1e-11, atol=1e-11, what="p_hit vs oracle")
        assert np.array_equal(hit, sf.mask(ph[:, 0], ph[:, 1]) & hit), "a hit lies inside the mask"


def test_mask_func_scene_image_matches_reference():
    """The masked scene (fixture from the reference, which calls mask_func per ray): detector image."""
    from helpers import sparse_to_dense, image_rel_l1
    g, RT = gpu_trace("masked")
    with ot.global_options.no_warnings():
        ph, hw, wl, ext, projection, ill = RT._hit_detector("x", 0, None, None, None)
        assert np.count_nonzero(hw.cpu().numpy() > 0) == g["det0/None/w"].shape[0]
        img = RT.detector_image(detector_index=0)
    pw = float(g["det0/None/img/power"])
    assert abs(img.power() - pw) <= 1e-6 * pw
    assert np.all(image_rel_l1(img._data, sparse_to_dense(g, "det0/None/img")) < 2e-3)
