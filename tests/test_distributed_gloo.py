"""N > 1 path on CPU: two gloo ranks shard the rays, reduce extent / histogram / counters with the product's
`optrace_amd.distributed` helpers, and must reproduce the single-process reference image.  The per-rank
ray work is done by the CPU oracle here (no GPU in this tier); on the GPU the same helpers run over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from optrace_amd import distributed as D


def test_shard_range_matches_reference_thread_split():
    for N, W in [(10, 3), (1000003, 8), (7, 7), (5, 8), (100, 1)]:
        got = [D.shard_range(N, r, W) for r in range(W)]
        assert got[0][0] == 0 and got[-1][1] == N
        for (a, b), (c, d) in zip(got[:-1], got[1:]):
            assert b == c
        Np = int(N / W)
        assert all(b - a == Np for a, b in got[:-1])


def test_shard_source_powers_conserve_power_and_count():
    P = [1.0, 2.0, 0.5, 1.5, 1.0]
    N = 100001
    tot_c, tot_p = np.zeros(5, dtype=np.int64), np.zeros(5)
    for r in range(4):
        c, p = D.shard_source_powers(N, P, r, 4)
        tot_c += c
        tot_p += p
    assert tot_c.sum() == N
    np.testing.assert_allclose(tot_p, P, rtol=1e-12)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    import sys, pathlib
    root = pathlib.Path(__file__).resolve().parent.parent
    sys.path[:0] = [str(root), str(root / "tests")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import optrace_amd as ot
    from optrace_amd import _capi
    from optrace_amd.scene import CompiledScene
    import oracle_bridge as ob
    import scenes
    from helpers import load

    g = load("trace_double_gauss.npz")
    N = int(g["N"])
    with ot.global_options.no_warnings():
        RT = scenes.double_gauss(ot)
    sc = CompiledScene(RT)
    first, end = D.shard_range(N, rank, world)
    n = end - first
    rays = ob.HostRays(n, sc.nt, False)
    rays.set_initial(g["p0"][first:end], g["s0"][first:end], g["pol0"][first:end], g["w0"][first:end], g["wl"][first:end])
    msgs, st = ob.trace(sc.desc, rays, None)
    det = RT.detectors[0].surface._desc()
    ph, hw, ext, ill, st = ob.detector_hits(rays, 0, n, det, _capi.PROJ_NONE)
    # 1) agree on the automatic extent
    ext = D.allreduce_extent(ext)
    img = ot.RenderImage(extent=ext)
    img._fix_extent()
    Nx, Ny = img._pixel_counts()
    sel = hw > 0
    hist = torch.from_numpy(ob.render(ph[sel, 0], ph[sel, 1], hw[sel], rays.wl[sel], img.extent, Nx, Ny))
    # 2) the one data exchange
    D.allreduce_image(hist)
    # 3) counters
    msgs = D.allreduce_counters(msgs)
    if rank == 0:
        np.savez(out, hist_nz=np.argwhere(hist.numpy()[..., 3] > 0), power=float(hist[..., 3].sum()),
                 vals=hist.numpy()[hist.numpy()[..., 3] > 0], msgs=msgs, extent=img.extent, shape=np.array(hist.shape))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gloo_image_equals_single_process_reference(tmp_path):
    from helpers import load, sparse_to_dense, image_rel_l1
    out = str(tmp_path / "r0.npz")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    g = load("trace_double_gauss.npz")
    assert np.array_equal(got["msgs"], g["msgs"]), "summed counters of the shards = counters of the whole"
    ref = sparse_to_dense(g, "det0/None/img")
    assert tuple(got["shape"]) == ref.shape
    np.testing.assert_allclose(got["extent"], g["det0/None/img/extent"], rtol=1e-9, atol=1e-11)
    img = np.zeros(ref.shape)
    idx = got["hist_nz"]
    img[idx[:, 0], idx[:, 1]] = got["vals"]
    assert abs(got["power"] - float(g["det0/None/img/power"])) < 1e-9 * float(g["det0/None/img/power"])
    assert np.all(image_rel_l1(img, ref) < 1e-4)


# ---- the exchanges of sharded_iterative_render (K images at once), two gloo ranks on CPU -------------------------------
def _worker_k(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # K = 3 positions; rank 1 has no hit at position 2, nobody has one at position 3
    inf = np.inf
    raw = [np.array([[-1.0, 2.0, -3.0, 0.5], [0.1, 0.2, 0.3, 0.4], [inf, -inf, inf, -inf]]),
           np.array([[-1.5, 1.0, -2.0, 4.0], [inf, -inf, inf, -inf], [inf, -inf, inf, -inf]])][rank]
    ext = D.allreduce_extents(raw)
    stack = torch.full((3, 4, 5, 4), float(rank + 1), dtype=torch.float64)
    D.allreduce_image(stack)
    cnt = D.allreduce_counters(np.arange(10).reshape(5, 2) * (rank + 1))
    if rank == 0:
        np.savez(out, ext=ext, stack=stack.numpy(), cnt=cnt)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_k_image_extent_agreement_and_stacked_histogram_exchange(tmp_path):
    out = str(tmp_path / "k.npz")
    mp.spawn(_worker_k, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    np.testing.assert_array_equal(got["ext"][0], [-1.5, 2.0, -3.0, 4.0])   # min / max over the ranks, per image
    np.testing.assert_array_equal(got["ext"][1], [0.1, 0.2, 0.3, 0.4])     # one rank without a hit: the other's extent
    assert not np.any(np.isfinite(got["ext"][2]))                          # no hit anywhere: the caller collapses it
    assert np.all(got["stack"] == 3.0)
    np.testing.assert_array_equal(got["cnt"], np.arange(10).reshape(5, 2) * 3)
    # without a process group the helpers are the identity
    raw = np.array([[0.0, 1.0, 2.0, 3.0]])
    np.testing.assert_array_equal(D.allreduce_extents(raw), raw)
