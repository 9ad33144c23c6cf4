"""Randomised scenes: GPU trace against the oracle on the same initial rays (masks, counters bit-exact).

The oracle is pinned to the reference by the fixtures; random geometries extend that pin to combinations the
fixtures do not contain (surface types x element order x media x sources)."""
import numpy as np
import pytest

import optrace_amd as ot
from optrace_amd.scene import CompiledScene

import oracle_bridge as ob
import scenes

pytestmark = pytest.mark.gpu


random_scene = lambda seed: scenes.random_scene(ot, seed, seed=seed)


@pytest.mark.parametrize("seed", range(24))
def test_random_scene_matches_oracle(seed):
    with ot.global_options.no_warnings():
        RT = random_scene(1000 + seed)
        N = 3000
        RT.trace(N)
        if RT.geometry_error:
            pytest.skip("random geometry collides")
    r = RT.rays
    sc = CompiledScene(RT)
    rays = ob.HostRays(N, sc.nt, RT.no_pol)
    p0 = r.p_list[:, 0]
    d = r.p_list[:, 1] - p0
    nrm = np.linalg.norm(d, axis=1)
    keep = nrm > 0  # rays stopped at the very first surface keep p_1 = p_0 only if they started on it
    s0 = np.where(keep[:, None], d / np.where(keep, nrm, 1)[:, None], [0, 0, 1.])
    rays.set_initial(p0, s0, None if RT.no_pol else r.pol_list[:, 0], r.w_list[:, 0], r.wl_list)
    msgs, st = ob.trace(sc.desc, rays, None)
    assert st == 0
    assert np.array_equal(rays.w_list > 0, r.w_list > 0), "alive masks per section"
    assert np.array_equal(msgs, RT._msgs), f"counters\n{msgs}\n{RT._msgs}"
    assert np.allclose(rays.p_list, r.p_list, rtol=1e-9, atol=1e-8)
    assert np.allclose(rays.w_list, r.w_list, rtol=2e-6, atol=1e-12)


@pytest.mark.parametrize("seed", range(8))
def test_random_scene_with_hurb_matches_oracle(seed):
    """Same, with edge diffraction on: the standard-normal draws are injected into both paths."""
    with ot.global_options.no_warnings():
        RT = random_scene(5000 + seed)
        RT.use_hurb = True
        RT.add(ot.Aperture(ot.RingSurface(r=6, ri=1.5), pos=[0, 0, -5]))
        N = 2000
        RT.trace(N)
        if RT.geometry_error:
            pytest.skip("random geometry collides")
        r = RT.rays
        p0 = r.p_list[:, 0].copy()
        d = r.p_list[:, 1] - p0
        s0 = d / np.linalg.norm(d, axis=1)[:, None]
        pol0 = None if RT.no_pol else r.pol_list[:, 0].copy()
        w0, wl = r.w_list[:, 0].copy(), r.wl_list.copy()
        sc = CompiledScene(RT)
        n_hurb = sum(1 for el in RT.apertures if isinstance(el.surface, (ot.RingSurface, ot.SlitSurface)))
        hn = np.random.default_rng(seed).standard_normal((2 * n_hurb, N))
        RT.trace(N, _initial_rays=(p0, s0, pol0, w0, wl), _hurb_normals=hn, _N_list=r.N_list)
        r = RT.rays
    rays = ob.HostRays(N, sc.nt, RT.no_pol)
    rays.set_initial(p0, s0, pol0, w0, wl)
    msgs, st = ob.trace(sc.desc, rays, hn)
    assert st == 0
    assert np.array_equal(rays.w_list > 0, r.w_list > 0)
    assert np.array_equal(msgs, RT._msgs), f"counters\n{msgs}\n{RT._msgs}"
    assert np.allclose(rays.p_list, r.p_list, rtol=1e-10, atol=1e-10)
