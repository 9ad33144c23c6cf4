"""N > 1 path through the HIP product code: ranks trace their ray shards on the device with
`distributed.sharded_detector_image`, reduce extent / histogram / counters, and must reproduce what one process
gets when it traces the same shards (same seeds) one after the other and adds the images -- the reference's own
composition rule for independent ray chunks (raytracer.py:1235-1267, ray split ray_storage.py:147-171).

Two gloo ranks share the one device of the test box (RCCL refuses two ranks on one GPU; the reductions then run
on host copies -- `distributed._device_collectives`); a world-size-1 NCCL group exercises the device-side
collectives.  `bench.py --gpus 2` must start its own ranks and report n_gpus = 2."""
import json
import os
import pathlib
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parent.parent
N_RAYS = 200_001  # odd: the last rank takes the remainder
BASE_SEED = 4242


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


_WORKER = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
root, backend, out, extent, N, base_seed = sys.argv[1:7]
sys.path[:0] = [root, root + "/tests"]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
if backend == "nccl":
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
else:
    dist.init_process_group(backend)
import optrace_amd as ot
from optrace_amd import distributed as D
import scenes
with ot.global_options.no_warnings():
    RT = scenes.double_gauss(ot)
    seed_before = RT.seed
    img = D.sharded_detector_image(RT, int(N), extent=None if extent == "auto" else [-45., 45., -45., 45.],
                                   base_seed=int(base_seed))
    assert RT.seed == seed_before, "the caller's seed is restored"
    msgs = RT._msgs.copy()
    powers = [rs.power for rs in RT.ray_sources]
    assert powers == [1.0] * 5, "source powers untouched"
    empty = None
    if extent == "auto":  # no ray of any rank reaches a detector far off axis: extent = detector centre
        RT.add(ot.Detector(ot.RectangularSurface(dim=[1, 1]), pos=[1500, 0, 150]))
        empty = D.sharded_detector_image(RT, 20000, detector_index=1, base_seed=7)
if rank == 0:
    np.savez(out, data=img._data, extent=np.asarray(img.extent), msgs=msgs,
             empty_extent=np.asarray(empty.extent) if empty is not None else np.zeros(4),
             empty_power=empty.power() if empty is not None else 0.0)
dist.barrier()
dist.destroy_process_group()
"""


def _run_ranks(tmp_path, world, backend, extent):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    out = tmp_path / f"img_{world}_{backend}_{extent}.npz"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script),
           str(ROOT), backend, str(out), extent, str(N_RAYS), str(BASE_SEED)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return np.load(out)


def _single_process(world, extent):
    """The same shards traced one after the other in this process, images added."""
    import optrace_amd as ot
    from optrace_amd import distributed as D
    import scenes
    with ot.global_options.no_warnings():
        RT = scenes.double_gauss(ot)
        shards = [D.shard_range(N_RAYS, r, world) for r in range(world)]
        ext = extent
        if extent is None:
            lo = np.array([np.inf, np.inf])
            hi = -lo
            for r, (a, b) in enumerate(shards):
                RT.seed = BASE_SEED + r
                RT.trace(b - a, _power_scale=(b - a) / N_RAYS)
                e = RT._hit_detector("Detector Image", 0, None, None, "Equidistant")[3]
                lo, hi = np.minimum(lo, e[[0, 2]]), np.maximum(hi, e[[1, 3]])
            ext = [lo[0], hi[0], lo[1], hi[1]]
        total, msgs = None, 0
        for r, (a, b) in enumerate(shards):
            RT.seed = BASE_SEED + r
            RT.trace(b - a, _power_scale=(b - a) / N_RAYS)
            img = RT.detector_image(extent=list(ext))
            total = img._data.copy() if total is None else total + img._data
            msgs = msgs + RT._msgs
    return total, np.asarray(img.extent), msgs


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,backend,extent", [(2, "gloo", "user"), (2, "gloo", "auto"), (1, "nccl", "user")])
def test_sharded_detector_image_equals_single_process_composition(tmp_path, world, backend, extent):
    got = _run_ranks(tmp_path, world, backend, extent)
    ref, ref_extent, ref_msgs = _single_process(world, None if extent == "auto" else [-45., 45., -45., 45.])
    assert np.array_equal(got["msgs"], ref_msgs), "summed counters of the ranks = counters of the shards"
    np.testing.assert_allclose(got["extent"], ref_extent, rtol=1e-12, atol=1e-12)
    assert got["data"].shape == ref.shape
    assert np.array_equal(got["data"][..., 3] > 0, ref[..., 3] > 0), "same pixels lit"
    # f64 sums in a different order only
    np.testing.assert_allclose(got["data"], ref, rtol=1e-9, atol=1e-12 * ref.max())
    # the image carries the full source power minus what the system absorbs
    assert 0.3 < got["data"][..., 3].sum() <= 5.0
    if extent == "auto":
        assert got["empty_power"] == 0.0
        assert np.all(np.isfinite(got["empty_extent"]))
        assert abs(got["empty_extent"][:2].mean() - 1500) < 1.0  # collapsed onto the detector centre, then widened


# ---- sharded iterative render: BASELINE config 4 as stated (trace once per rank, K positions, one histogram exchange) ----
_WORKER_ITER = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
root, backend, out, extent, N, base_seed = sys.argv[1:7]
sys.path[:0] = [root, root + "/tests"]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
if backend == "nccl":
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
else:
    dist.init_process_group(backend)
import optrace_amd as ot
from optrace_amd import distributed as D
import scenes
with ot.global_options.no_warnings():
    RT = scenes.double_gauss(ot)
    RT.ITER_RAYS_STEP = 30_000  # several chunks per rank
    z0 = RT.detectors[0].pos[2]
    pos = [[0, 0, z0 - 3], [0, 0, z0], [0, 0, z0 + 3]]
    ext = None if extent == "auto" else [-45., 45., -45., 45.]
    traces = []
    trace0 = RT.trace
    def counting_trace(*a, **k):
        traces.append(k.get("N", a[0] if a else None))
        return trace0(*a, **k)
    RT.__dict__["trace"] = counting_trace  # (instance attribute, no tracked assignment)
    imgs = D.sharded_iterative_render(RT, int(N), pos=pos, extent=ext, base_seed=int(base_seed))
    n_local = D.shard_range(int(N), rank, world)
    n_local = n_local[1] - n_local[0]
    assert sum(traces) == n_local, "every rank traces its shard exactly once, whatever the number of positions"
if rank == 0:
    np.savez(out, data=np.stack([im._data for im in imgs]), extent=np.stack([np.asarray(im.extent) for im in imgs]),
             msgs=RT._msgs, sent=D.last_exchange["bytes"], window=np.array(D.last_exchange["window"] or [[-1] * 4] * 3))
dist.barrier()
dist.destroy_process_group()
"""


def _single_process_iter(world, extent):
    """The shards rendered one after the other in this process (same seeds, same chunks), images added.  The common
    automatic extents come from the two-step hit lists of every shard's last chunk (the stored one, traced first)."""
    import optrace_amd as ot
    from optrace_amd import distributed as D
    import scenes
    with ot.global_options.no_warnings():
        RT = scenes.double_gauss(ot)
        RT.ITER_RAYS_STEP = 30_000
        z0 = RT.detectors[0].pos[2]
        pos = [[0, 0, z0 - 3], [0, 0, z0], [0, 0, z0 + 3]]
        shards = [D.shard_range(N_RAYS, r, world) for r in range(world)]
        agreed = None
        if extent is None:
            lo = np.full((3, 2), np.inf)
            hi = -lo
            for r, (a, b) in enumerate(shards):
                RT.seed = BASE_SEED + r
                # (the extents are those of the shard's stored LAST chunk, which a render with render-only chunks traces first)
                plan = RT._chunk_plan(b - a, len(RT.tracing_surfaces) + 2, True)
                RT.trace(plan[-1], _chunk=len(plan) - 1, _power_scale=(b - a) / N_RAYS)
                for k, p in enumerate(pos):
                    e = RT._hit_detectors("Detector Image", [dict(detector_index=0, extent=None, pos=p,
                                                                  projection_method="Equidistant")])[0][3]
                    lo[k], hi[k] = np.minimum(lo[k], e[[0, 2]]), np.maximum(hi[k], e[[1, 3]])
            agreed = np.stack([lo[:, 0], hi[:, 0], lo[:, 1], hi[:, 1]], axis=1)
        total, msgs, exts = None, 0, None
        for r, (a, b) in enumerate(shards):
            RT.seed = BASE_SEED + r
            imgs = RT.iterative_render(b - a, pos=pos, extent=extent, _power_scale=(b - a) / N_RAYS,
                                       _agree_extents=None if agreed is None else (lambda raw: agreed))
            d = np.stack([im._data for im in imgs])
            total = d if total is None else total + d
            msgs = msgs + RT._msgs
            exts = np.stack([np.asarray(im.extent) for im in imgs])
    return total, exts, msgs


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,backend,extent", [(2, "gloo", "user"), (2, "gloo", "auto"), (1, "nccl", "auto")])
def test_sharded_iterative_render_equals_single_process_composition(tmp_path, world, backend, extent):
    """K = 3 detector positions, several chunks per rank, user and automatic extents (raytracer.py:1134-1279)."""
    script = tmp_path / "worker_iter.py"
    script.write_text(_WORKER_ITER)
    out = tmp_path / f"iter_{world}_{backend}_{extent}.npz"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script),
           str(ROOT), backend, str(out), extent, str(N_RAYS), str(BASE_SEED)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    got = np.load(out)
    ref, ref_ext, ref_msgs = _single_process_iter(world, None if extent == "auto" else [-45., 45., -45., 45.])
    assert np.array_equal(got["msgs"], ref_msgs), "summed counters of the ranks = counters of the shards"
    np.testing.assert_allclose(got["extent"], ref_ext, rtol=1e-12, atol=1e-12)
    assert got["data"].shape == ref.shape and got["data"].shape[0] == 3
    assert np.array_equal(got["data"][..., 3] > 0, ref[..., 3] > 0), "same pixels lit"
    np.testing.assert_allclose(got["data"], ref, rtol=1e-9, atol=1e-12 * ref.max())
    for k in range(3):  # every position sees the full source power minus what the system absorbs
        assert 0.3 < got["data"][k, ..., 3].sum() <= 5.0
    if world > 1 and extent == "user":  # five spots in a row across a 90 x 90 mm extent: only their window was exchanged
        sent = 0
        for k, (y0, y1, x0, x1) in enumerate(got["window"]):  # per image its own window
            lit = np.argwhere(got["data"][k, ..., 3] > 0)
            assert y0 <= lit[:, 0].min() and lit[:, 0].max() < y1 and x0 <= lit[:, 1].min() and lit[:, 1].max() < x1
            sent += (y1 - y0) * (x1 - x0) * 32
        assert int(got["sent"]) == sent < 0.7 * got["data"].size * 8
    if world == 1:
        assert int(got["sent"]) == 0


@pytest.mark.timeout(900)
def test_bench_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` without a torch.distributed environment starts two ranks itself (they share the
    one device here, so the collectives fall back to gloo) and rank 0 reports n_gpus = 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                        "--rays", "1000000", "--skip-cpu"], env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert out["config"]["ranks_per_device"] == (2 if torch.cuda.device_count() == 1 else 1)
    assert out["value"] > 0 and out["roofline"]["kernel_ms"] > 0
    # both ranks' 1 M-ray bundles: 5 sources of unit power, part of it absorbed at the stop
    assert 1.0 < out["detector"]["image_power_all_ranks"] <= 10.0
    # configs 4 and 5 as stated, sharded over the two ranks: the renders with their exchanges
    for key, n_img in (("c4_sharded", 6), ("c5_sharded", 1)):
        blk = out[key]
        assert blk["n_gpus"] == 2 and blk["scaling"] == "strong" and blk["positions"] == n_img
        assert blk[key + "_ms"] > 0 and blk["histogram_allreduce_ms"] > 0
        assert len(blk["image_power"]) == n_img and all(p > 0 for p in blk["image_power"])
    assert "configs" not in out, "the per-configuration block is rank 0's at one GPU"


@pytest.mark.timeout(900)
def test_bench_line_carries_every_baseline_config():
    """One GPU: the driver's line has the `configs` block (C1 at 1e7 rays, C3, C4, C5, A1 at full size: trace kernel time and
    roofline fraction, detector image with an automatic and a given extent, the chunked render) and both sharded blocks."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "3", "--warmup", "2", "--skip-cpu"], env=env,
                       capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and 0.3 < out["roofline"]["frac"] < 1.0
    cfgs = out["configs"]
    assert set(cfgs) == {"C1_single_lens_1e7", "C3_arizona_eye_rgb_5e7", "C4_image_render_2e8", "C5_hurb_slit_lens_1e8",
                         "A1_double_gauss_aspheric_1e7"}
    for name, c in cfgs.items():
        assert "error" not in c, (name, c)
        assert c["trace_ms"] > 0 and 0.05 < c["roofline_frac"] < 1.0
        assert c["algorithmic_bytes"] == c["rays"] * (c["sections"] * (48 if c["pol"] else 36) + 28)
        for key in ("detector_image_auto", "detector_image_user"):
            assert c[key]["ms"] > 0 and 0 < c[key]["frac"] < 1.0
        assert c["iterative_render"]["ms"] >= c["trace_ms"] * 0.5
    assert cfgs["C4_image_render_2e8"]["iterative_render"]["positions"] == 6
    assert out["c4_sharded"]["n_gpus"] == 1 and out["c5_sharded"]["rays_total"] == 100_000_000
