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


# ---- allreduce_images: one message, lit window only ---------------------------------------------------------------------
def test_lit_window_and_block_view():
    pool = torch.zeros(3 * 6 * 7 * 4, dtype=torch.float64)
    hists = [pool[k * 168:(k + 1) * 168].view(6, 7, 4) for k in range(3)]
    block = D._as_one_block(hists)
    assert block is not None and block.shape == (3, 6, 7, 4) and block.data_ptr() == pool.data_ptr()
    assert D._as_one_block([hists[0], hists[2]]) is None                    # a gap
    assert D._as_one_block([hists[0], torch.zeros(6, 7, 4, dtype=torch.float64)]) is None  # another allocation
    assert D.lit_window(block).tolist() == [[6, 0, 7, 0]] * 3                # all dark: empty windows
    hists[0][2, 3, 3] = 1.0
    hists[2][4, 1, 0] = -2.0                                                # (any channel counts)
    assert D.lit_window(block).tolist() == [[2, -3, 3, -4], [6, 0, 7, 0], [4, -5, 1, -2]]
    block[1, 0, 6, 1] = 5.0
    hists[0][5, 0, 2] = 1.0
    assert D.lit_window(block).tolist() == [[2, -6, 0, -4], [0, -1, 6, -7], [4, -5, 1, -2]]
    # without a process group: nothing to exchange
    assert D.allreduce_images(hists) == dict(bytes=0, window=None)
    with pytest.raises(TypeError):
        D.allreduce_images([torch.zeros(3, 3, 4)])


def _worker_img(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(5 + rank)
    res = {}

    def images(K, Ny, Nx, y0, y1, x0, x1, pooled=True):
        if pooled:
            pool = torch.zeros(K * Ny * Nx * 4, dtype=torch.float64)
            hs = [pool[k * Ny * Nx * 4:(k + 1) * Ny * Nx * 4].view(Ny, Nx, 4) for k in range(K)]
        else:
            hs = [torch.zeros(Ny, Nx, 4, dtype=torch.float64) for _ in range(K)]
        for h in hs:
            h[y0:y1, x0:x1] = torch.from_numpy(rng.random((y1 - y0, x1 - x0, 4)))
        return hs

    # 1) pictures in a part of the extent, the ranks' windows differ: per image the union travels
    hs = images(3, 40, 50, 10 + rank, 20 + rank, 5, 15 + 2 * rank)
    hs[1][:] = 0
    hs[1][30:33, 40 - rank:45] = 1.5 + rank                                # the second image: another place
    hs[2][:] = 0                                                            # the third: dark on both ranks
    mine = [h.clone() for h in hs]
    info = D.allreduce_images(hs)
    res["win"], res["win_bytes"] = np.array(info["window"]), info["bytes"]
    res["a_sum"], res["a_mine"] = torch.stack(hs).numpy(), torch.stack(mine).numpy()
    # 2) separate allocations: stacked, exchanged, copied back
    hs = images(2, 30, 30, 3, 9, 4, 8, pooled=False)
    mine = [h.clone() for h in hs]
    info = D.allreduce_images(hs)
    res["b_bytes"] = info["bytes"]
    res["b_sum"], res["b_mine"] = torch.stack(hs).numpy(), torch.stack(mine).numpy()
    # 3) the picture fills the extent: the histograms travel whole
    hs = images(2, 16, 16, 0, 16, 1, 16)
    mine = [h.clone() for h in hs]
    info = D.allreduce_images(hs)
    res["c_window_is_none"], res["c_bytes"] = info["window"] is None, info["bytes"]
    res["c_sum"], res["c_mine"] = torch.stack(hs).numpy(), torch.stack(mine).numpy()
    # 4) nobody has a hit; 5) shapes differ: one exchange per image
    info = D.allreduce_images(images(2, 8, 8, 0, 0, 0, 0))
    res["d_bytes"], res["d_window"] = info["bytes"], np.array(info["window"])
    hs = images(1, 8, 9, 1, 3, 1, 3) + images(1, 9, 8, 2, 4, 2, 4)
    mine = [h.clone() for h in hs]
    info = D.allreduce_images(hs)
    res["e_bytes"] = info["bytes"]
    for k in range(2):
        res[f"e_sum{k}"], res[f"e_mine{k}"] = hs[k].numpy(), mine[k].numpy()
    np.savez(out + f".{rank}.npz", **res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_histogram_exchange_sends_the_lit_window_only(tmp_path):
    out = str(tmp_path / "img")
    mp.spawn(_worker_img, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = np.load(out + ".0.npz"), np.load(out + ".1.npz")
    # image 0: the union of rows 10-19 / 11-20, columns 5-14 / 5-16; image 1: rows 30-32, columns 39-44; image 2: nothing
    np.testing.assert_array_equal(r0["win"], [[10, 21, 5, 17], [30, 33, 39, 45], [0, 0, 0, 0]])
    assert int(r0["win_bytes"]) == (11 * 12 + 3 * 6) * 32 < 3 * 40 * 50 * 32
    assert int(r0["b_bytes"]) == 2 * 6 * 4 * 32
    assert bool(r0["c_window_is_none"]) and int(r0["c_bytes"]) == 2 * 16 * 16 * 32
    assert int(r0["d_bytes"]) == 0 and not r0["d_window"].any() and r0["d_window"].shape == (2, 4)
    assert int(r0["e_bytes"]) == (8 * 9 + 9 * 8) * 32
    for key in ("a", "b", "c"):
        want = r0[key + "_mine"] + r1[key + "_mine"]
        np.testing.assert_array_equal(r0[key + "_sum"], want)               # two summands: no rounding order to speak of
        np.testing.assert_array_equal(r1[key + "_sum"], want)               # identical on every rank
    for k in range(2):
        np.testing.assert_array_equal(r0[f"e_sum{k}"], r0[f"e_mine{k}"] + r1[f"e_mine{k}"])
