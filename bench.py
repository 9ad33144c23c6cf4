#!/usr/bin/env python3
"""Headline benchmark: ray-surface-intersections/s of Raytracer.trace() on the double-Gauss scene.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--rays R] [--no-pol] [--backend nccl|gloo]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one call of `Raytracer.trace(N)` -- the call the reference times in tests/benchmark.py:81-86 -- on one
bundle of synthetic rays: on-device ray generation from the five point sources, all 15 tracing surfaces, all
section stores, the event counters back on the host.  Workload = BASELINE.json configs[1]: double_gauss.py
geometry, 10 M rays per GPU, 3 wavelengths (FDC lines), polarisation on.  Every step draws a fresh seed.

`--gpus N` without a torch.distributed environment starts the N ranks itself (a child `torch.distributed.run`,
before this process makes any GPU call); launched by torch.distributed.run it is one of the ranks.  Rays are
sharded, every rank traces its own bundle (weak scaling), no data-path collective inside the step; after the
timed region the detector histograms are all-reduced once (RCCL) to exercise the exchange step.  With fewer
devices than ranks the ranks share devices and the collectives run over gloo on host copies (a rehearsal of the
multi-rank code path, labelled as such in the output).

Rank 0 prints ONE JSON line (contract in the task description) carrying `roofline` (dominant kernel: algorithmic
bytes / HIP-event kernel time against the 8 TB/s HBM peak, plus the f64 issue fraction) and `cpu_baseline` (the
CPU oracle, a scalar port of the reference, timed on a bounded sample on the host cores).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import pathlib
import subprocess
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
# f64 mul/add issue rate of the whole chip, wave64 instructions per second, measured with
# tools/experiments/op_rate.hip (profiles/r1/op_rate.txt: 4 independent chains, 8 waves per SIMD)
F64_ISSUE_PEAK = 536.86e9


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=30)  # the first ~10 launches run while the clocks settle
    ap.add_argument("--rays", type=int, default=10_000_000, help="rays per GPU and step")
    ap.add_argument("--no-pol", action="store_true")
    ap.add_argument("--skip-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--skip-configs", action="store_true", help="skip the `configs` block (the other BASELINE configs "
                                                                "and config 4's sharded render, after the timed region)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo: collectives on "
                                                      "host copies, also picked when ranks have to share a device)")
    return ap.parse_args()


def spawn_ranks(args) -> int:
    """Parent of a `--gpus N` run: N ranks as children of one torch.distributed.run.  Nothing here touches the GPU
    (device_count does not initialise it), so the children start from a clean process."""
    import socket
    import torch
    backend = args.backend
    if backend == "nccl" and torch.cuda.device_count() < args.gpus:
        backend = "gloo"  # RCCL refuses two ranks on one device; rehearse the path on host copies instead
        print(f"bench.py: {torch.cuda.device_count()} device(s) for {args.gpus} ranks: ranks share devices, "
              "collectives over gloo", file=sys.stderr)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(ROOT / "bench.py"),
           "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--rays", str(args.rays), "--backend", backend, "--cpu-seconds", str(args.cpu_seconds)]
    cmd += ["--no-pol"] if args.no_pol else []
    cmd += ["--skip-cpu"] if args.skip_cpu else []
    cmd += ["--skip-configs"] if args.skip_configs else []
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


def cpu_baseline(RT, scene, seconds: float) -> dict:
    """Time the CPU oracle (kind "port": oracle/oracle.c, scalar restatement of the reference's NumPy path,
    pinned to it by tests/test_oracle_golden.py) on a bounded sample of the same workload: rays generated
    by the device kernel for this scene, traced through all 15 surfaces on the host cores (rays split across
    threads like the reference splits them, ray_storage.py:147-171; one core is timed as well)."""
    import numpy as np
    import oracle_bridge as ob  # checker / baseline only
    M = scene.nt - 2
    r = RT.rays
    n_avail = r.N
    threads = int(os.environ.get("OT_CPU_THREADS", min(os.cpu_count() or 1, 16)))

    d, N, nt = r._dev, r.N, scene.nt
    pv = d["p"].view(3, nt, r._Np)  # element (ray, section, component) lives at ray + stride * (section + nt * component)

    def run(n, th):
        rays = ob.HostRays(n, scene.nt, RT.no_pol)
        # only the first two sections of the first n rays leave the device (the storage is 8.4 GB)
        p0 = pv[:, 0, :n].t().cpu().numpy()
        dirs = pv[:, 1, :n].t().cpu().numpy() - p0
        s0 = dirs / np.linalg.norm(dirs, axis=1)[:, None]
        pol0 = None if RT.no_pol else d["pol"].view(3, nt, r._Np)[:, 0, :n].t().cpu().numpy()
        rays.set_initial(p0, s0, pol0, d["w"][:n].cpu().numpy(), d["wl"][:n].cpu().numpy())
        t0 = time.perf_counter()
        ob.trace(scene.desc, rays, None, threads=th)
        return time.perf_counter() - t0

    n1 = min(1_000_000, n_avail)
    t1 = run(n1, 1)  # one core
    n = min(200_000, n_avail)
    t = run(n, threads)
    n2 = int(min(n_avail, max(n, n * seconds / max(t, 1e-6))))
    if n2 > n:
        t, n = run(n2, threads), n2
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": n * M / t, "unit": "ray-surface-intersections/s", "cores": threads, "kind": "port",
            "cpu_model": model, "host_cores_available": os.cpu_count(), "value_one_core": n1 * M / t1,
            "sample": f"{n} rays x {M} surfaces of the same scene (device-generated rays), {t:.1f} s on {threads} "
                      f"threads; {n1} rays on 1 thread in {t1:.2f} s"}


def raw_launches(RT, lib, N, steps, warmup=10):
    """`steps` asynchronous ot_generate_and_trace launches on the resident storage of RT, no host work in between:
    (mean wall ms per launch, mean kernel ms from the library's events of the last launch)."""
    import torch
    from optrace_amd import _capi
    from optrace_amd._device import ptr, stream_ptr
    nt = RT.rays.Nt
    rays, tab, rng = RT.rays._rays_struct(), RT._source_cache[1], RT.rays._source_ranges()
    msgs = torch.zeros(5 * nt + 1, dtype=torch.int64, device=RT.rays._dev["p"].device)
    for i in range(warmup + steps):
        if i == warmup:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        _capi.check(lib.ot_generate_and_trace(RT._scene_handle, tab.handle, rng, len(rng), 500 + i, C.byref(rays),
                                              ptr(msgs), stream_ptr()))
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


def secondary_no_pol(ot, scenes, lib, N, steps):
    """The bench scene with no_pol=True through the same API: {value, ms_per_step, roofline_frac}."""
    import torch
    with ot.global_options.no_warnings():
        RT = scenes.double_gauss(ot, no_pol=True, seed=None)
        for i in range(10 + steps):
            if i == 10:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            RT.trace(N)
    t = (time.perf_counter() - t0) / steps
    nt = RT.rays.Nt
    b = N * (nt * 36 + 28)
    return {"value": N * (nt - 2) / t, "ms_per_step": 1e3 * t, "roofline_frac": b / t / 1e9 / HBM_PEAK_GBS,
            "algorithmic_bytes_per_launch": b}


def _median(v):
    v = sorted(v)
    return v[len(v) // 2]


def _release(torch, lib):
    """Between two configurations: the library's kept scratch and torch's cached blocks go back to the driver."""
    import gc
    from optrace_amd import _capi
    gc.collect()
    _capi.check(lib.ot_scratch_trim())
    torch.cuda.empty_cache()


def _wall_ms(torch, f, reps=5):
    """Median wall time of f() in ms, synchronised on both sides, after one untimed call."""
    f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    return _median(ts)


def measure_config(ot, torch, lib, build, N, user_extent=None, positions=None):
    """One BASELINE configuration at its full ray count on this GPU, after the timed region:
    `trace_ms` = duration of the trace kernel from the library's HIP events (median of 5 after a short settle),
    `roofline_frac` = SURVEY 8(d) bytes of `RayStorage` / trace_ms / 8 TB/s; `detector_image_ms` (wall time of the call,
    median of 5) with an automatic and with a given extent, `frac` against N * 56 B + Ny * Nx * 32 B; `iterative_render` =
    the chunked render of N rays end to end (one position with the extent found above, or `positions` with `user_extent`:
    config 4's six)."""
    from optrace_amd import _capi
    with ot.global_options.no_warnings():
        RT = build(ot)
        RT.trace(min(N, 100_000))
        _capi.check(lib.ot_scene_set_timing(RT._scene_handle, 1))
        RT._kernel_ms_log = log = []
        for _ in range(12 if N <= 10_000_000 else 3):  # settle: the clocks under this load
            RT.trace(N)
        del log[:]
        walls = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            RT.trace(N)
            torch.cuda.synchronize()
            walls.append(1e3 * (time.perf_counter() - t0))
        trace_ms, wall_ms = _median(log), _median(walls)
        RT._kernel_ms_log = None
        nt = RT.rays.Nt
        M = nt - 2
        b = N * (nt * (36 if RT.no_pol else 48) + 28)
        out = {"rays": N, "surfaces": M, "sections": nt, "pol": not RT.no_pol, "trace_ms": trace_ms,
               "trace_call_ms": wall_ms, "algorithmic_bytes": b, "roofline_frac": b / (trace_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
               "ray_surface_intersections_per_s": N * M / (trace_ms * 1e-3)}
        if RT.detectors:
            def det_entry(ms, img):
                Ny, Nx = img._dev.shape[:2]
                bd = N * 56 + Ny * Nx * 32
                return {"ms": ms, "image": [int(Ny), int(Nx)], "frac": bd / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "rays_per_s": N / (ms * 1e-3)}
            img = RT.detector_image(_keep_on_device=True)
            ext = [float(v) for v in (user_extent if user_extent is not None else img._extent0)]
            ms = _wall_ms(torch, lambda: RT.detector_image(_keep_on_device=True))
            out["detector_image_auto"] = det_entry(ms, img)
            img = RT.detector_image(extent=ext, _keep_on_device=True)
            ms = _wall_ms(torch, lambda: RT.detector_image(extent=ext, _keep_on_device=True))
            out["detector_image_user"] = det_entry(ms, img)
            out["detector_image_user"]["extent"] = ext
            del img
        if RT.detectors:  # the chunked render end to end: trace of every chunk (render-only but the last) + binning
            n_pos = 1 if positions is None else len(positions)
            kw = dict(extent=ext) if positions is None else dict(pos=positions, extent=[list(user_extent)] * n_pos)
            RT.iterative_render(N, **kw)  # untimed: allocator pools at this chunk size
            torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                RT.iterative_render(N, **kw)
                torch.cuda.synchronize()
                ts.append(1e3 * (time.perf_counter() - t0))
            out["iterative_render"] = {"ms": _median(ts), "positions": n_pos, "extent": "user",
                                       "rays_per_s": N / (_median(ts) * 1e-3)}
        _capi.check(lib.ot_scene_set_timing(RT._scene_handle, 0))
    del RT
    _release(torch, lib)
    return out


def configs_block(ot, torch, lib, scenes):
    """The BASELINE configurations bench.py's headline does not time (C1 at 1e7 rays, C3, C4, C5) and the aspheric
    double Gauss (A1), one after the other on this GPU, every storage released before the next is built."""
    table = [
        ("C1_single_lens_1e7", lambda o: scenes.c1_single_lens(o, seed=1), 10_000_000, None, None),
        ("C3_arizona_eye_rgb_5e7", scenes.c3_arizona_eye_rgb, 50_000_000, None, None),
        ("C4_image_render_2e8", scenes.c4_image_render, 200_000_000, [-8., 8., -8., 8.], scenes.C4_POSITIONS),
        ("C5_hurb_slit_lens_1e8", lambda o: scenes.hurb_slit_lens(o, seed=51), 100_000_000, None, None),
        ("A1_double_gauss_aspheric_1e7", lambda o: scenes.double_gauss(o, seed=1, aspheric=True), 10_000_000,
         [-45., 45., -45., 45.], None),
    ]
    out = {}
    for name, build, N, ext, positions in table:
        t0 = time.perf_counter()
        try:
            out[name] = measure_config(ot, torch, lib, build, N, ext, positions)
        except Exception as err:  # an out-of-memory box must not take the headline line with it
            out[name] = {"error": f"{type(err).__name__}: {err}"[:300]}
            _release(torch, lib)
        out[name]["block_s"] = time.perf_counter() - t0
    return out


def c4_sharded_block(ot, torch, lib, scenes, dist, use_dist, backend, dev, world, steps=3):
    """BASELINE config 4 AS STATED: image_render_many_rays.py, 2e8 rays sharded over the ranks, six detector positions,
    one RCCL reduce of the six stacked histograms -- `distributed.sharded_iterative_render` (every rank traces its shard
    once, in render-only chunks plus one stored chunk, and bins each chunk into all positions in one pass).  Strong
    scaling: the 2e8 rays are the job's.  -> dict on every rank (the caller prints rank 0's)."""
    return sharded_block(ot, torch, lib, dist, use_dist, backend, dev, world, scenes.c4_image_render, 200_000_000,
                         scenes.C4_POSITIONS, [[-8., 8., -8., 8.]] * len(scenes.C4_POSITIONS),
                         "image_render_many_rays.py geometry, 2e8 rays sharded over the ranks, six detector positions "
                         "(user extents), distributed.sharded_iterative_render", "c4_sharded_ms", steps)


def c5_sharded_block(ot, torch, lib, scenes, dist, use_dist, backend, dev, world, steps=3):
    """BASELINE config 5 as stated: hurb_apertures.py's slit + lens, HURB on, polarisation tracked, 1e8 rays sharded over
    the ranks, the detector image with an AUTOMATIC extent: the ranks agree on the extent of their first chunks' hits (one
    MIN and one MAX all-reduce of 2 doubles each, raytracer.py:1042-1049, 1262) before they bin, then one histogram
    reduce."""
    return sharded_block(ot, torch, lib, dist, use_dist, backend, dev, world, lambda o: scenes.hurb_slit_lens(o, seed=51),
                         100_000_000, None, None,
                         "hurb_apertures.py slit + lens, HURB on, polarisation on, 1e8 rays sharded over the ranks, one "
                         "detector image with an automatic extent (extent agreement + histogram reduce), "
                         "distributed.sharded_iterative_render", "c5_sharded_ms", steps)


def sharded_block(ot, torch, lib, dist, use_dist, backend, dev, world, build, N, pos, ext, workload, key, steps):
    from optrace_amd import _capi
    from optrace_amd import distributed as D
    on_host = use_dist and backend != "nccl"

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    with ot.global_options.no_warnings():
        RT = build(ot)
        RT.trace(100_000)
        _capi.check(lib.ot_scene_set_timing(RT._scene_handle, 1))
        D.sharded_iterative_render(RT, N, pos=pos, extent=ext, base_seed=1)  # untimed: tables, allocator pools, communicator
        RT._kernel_ms_log = log = []
        ts = []
        for k in range(steps):
            sync()
            t0 = time.perf_counter()
            imgs = D.sharded_iterative_render(RT, N, pos=pos, extent=ext, base_seed=100 + 10 * k)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        sync()
        trace_ms = sum(log) / steps  # this rank's trace kernels per render (all its chunks)
        RT._kernel_ms_log = None
        # the exchange alone, as the render does it (the window of lit pixels of all histograms in one message) and whole
        tr, tw = [], []
        for _ in range(3):
            sync()
            t1 = time.perf_counter()
            D.allreduce_images([im._dev for im in imgs])
            torch.cuda.synchronize()
            tr.append(1e3 * (time.perf_counter() - t1))
        sent = dict(D.last_exchange)
        stack = torch.stack([im._dev for im in imgs])
        for _ in range(3):
            sync()
            t1 = time.perf_counter()
            D.allreduce_image(stack)
            torch.cuda.synchronize()
            tw.append(1e3 * (time.perf_counter() - t1))
        power = [float(im.power()) for im in imgs]
        shapes = [list(im._dev.shape) for im in imgs]
        _capi.check(lib.ot_scene_set_timing(RT._scene_handle, 0))
    t = torch.tensor([_median(ts), trace_ms, -trace_ms, _median(tr), _median(tw)], dtype=torch.float64,
                     device="cpu" if on_host else dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    t = t.cpu().tolist()
    del RT, imgs, stack
    _release(torch, lib)
    return {"workload": workload, "rays_total": N, "n_gpus": world,
            "scaling": "strong", "positions": len(shapes), key: 1e3 * t[0], "rays_per_s": N / t[0],
            "trace_ms_per_rank_min_max": [-t[2], t[1]], "histogram_allreduce_ms": t[3],
            "histogram_bytes_sent": sent["bytes"], "lit_window": sent["window"],
            "whole_histograms_allreduce_ms": t[4],
            "histogram_bytes": int(sum(sh[0] * sh[1] * 32 for sh in shapes)), "image_shapes": shapes,
            "image_power": power, "backend": backend if use_dist else None}


def committed_profile(name: str, pol: bool, N: int):
    """Newest profiles/<round>/<name> recorded for this configuration, or (None, None)."""
    best = (None, None)
    for f in sorted((ROOT / "profiles").glob("*/" + name)):
        try:
            d = json.loads(f.read_text())
            if d.get("rays", N) == N and d.get("pol", pol) == pol:
                best = (d, str(f.relative_to(ROOT)))
        except Exception:
            pass
    return best


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # launched by torch.distributed.run (also with one rank: the same RCCL path as the N > 1 runs)
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    ndev = max(torch.cuda.device_count(), 1)
    backend = args.backend
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl" and world > ndev:
            backend = "gloo"
        import datetime
        limit = datetime.timedelta(seconds=300)  # a rank that never arrives fails the run instead of hanging it
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank % ndev), timeout=limit)
        else:
            dist.init_process_group(backend=backend, timeout=limit)
    local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import optrace_amd as ot
    from optrace_amd import _capi
    from optrace_amd import distributed as D
    import scenes

    lib = _capi.load_library()
    N = args.rays
    with ot.global_options.no_warnings():
        RT = scenes.double_gauss(ot, no_pol=args.no_pol, seed=None)
        np.random.seed(1000 + rank)  # an unseeded tracer draws a fresh seed per trace from NumPy's generator
        RT.trace(min(N, 100_000))    # compiles the scene, uploads the tables
    assert not RT.geometry_error
    scene = RT._scene
    nt = scene.nt
    M = nt - 2
    _capi.check(lib.ot_scene_set_timing(RT._scene_handle, 1))
    ms = C.c_double()

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # Set-up: the first collective of a process group builds its communicator (milliseconds to seconds).  Done here, the
    # barrier in front of the timed region is a warm one (~50 us).  That matters beyond its own duration: an idle gap of
    # more than ~1 ms sends the GPU's power controller through a transient -- the first launch after it runs boosted, the
    # next twenty throttled (2.2 ms falling back to 1.6; tools/nccl_after_effect.py, profiles/r3/idle_gap_transient.txt) --
    # and a cold barrier right before the timed steps cost them 7 %.
    if use_dist:
        for _ in range(2):
            barrier()

    # Set-up, untimed: let the clocks settle.  The first launches of this size run while the GPU ramps its clock under
    # the f64 load and runs into its power limit (the driver calls with --warmup 5, fewer than that takes; the very first
    # launches are the FAST ones, 1.56 against 1.61 ms): full-size traces until the kernel time the library measures, averaged
    # over five launches, moves by less than 0.5 % against the five before, at most 40 (~70 ms).
    # With several ranks the phase runs in LOCKSTEP: after every launch the ranks exchange one flag (a warm 50 us collective)
    # and go on until ALL have settled.  They then reach the warm-up steps and the barrier in front of the timed region within
    # a fraction of a step of one another -- a rank that settled early would otherwise wait there idle for tens of
    # milliseconds and walk into the transient described above.
    settle_launches, settle_hist = 0, []
    flag = torch.zeros(1, dtype=torch.int32, device="cpu" if (use_dist and backend != "nccl") else dev)
    with ot.global_options.no_warnings():
        while settle_launches < 40:
            RT.trace(N)
            _capi.check(lib.ot_scene_last_trace_ms(RT._scene_handle, C.byref(ms)))
            settle_hist.append(ms.value)
            settle_launches += 1
            settled = False
            if settle_launches >= 10:  # means of the last five launches and of the five before within 0.5 %
                a5, b5 = sum(settle_hist[-5:]) / 5, sum(settle_hist[-10:-5]) / 5
                settled = abs(a5 - b5) < 0.005 * b5
            if use_dist:
                flag.fill_(1 if settled else 0)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                settled = bool(flag.item())
            if settled:
                break

    with ot.global_options.no_warnings():
        for i in range(args.warmup):
            RT.trace(N)
        barrier()
        kernel_ms = []
        t0 = time.perf_counter()
        for i in range(args.steps):
            RT.trace(N)  # synchronous like the reference's: returns when the rays and the counters are there
            _capi.check(lib.ot_scene_last_trace_ms(RT._scene_handle, C.byref(ms)))  # events around the kernel
            kernel_ms.append(ms.value)
        torch.cuda.synchronize()
        t_local = time.perf_counter() - t0  # this rank's K steps (all ranks left the opening barrier together; the MAX over
        barrier()                           # ranks below is the job's time -- the closing barrier itself, an RCCL collective
                                            # of ~1.5 ms, is not part of the steps)
    kernel_ms = float(np.mean(kernel_ms))
    _capi.check(lib.ot_scene_set_timing(RT._scene_handle, 0))

    on_host = use_dist and backend != "nccl"  # collectives on host copies
    t = torch.tensor([t_local], dtype=torch.float64, device="cpu" if on_host else dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    t_max = float(t.item())

    # ---- after the timed region: raw launches, detector image + the one exchange step (histogram all-reduce) ----
    raw_ms = raw_launches(RT, lib, N, args.steps)
    with ot.global_options.no_warnings():
        RT.trace(N)  # (the raw launches bypassed the tracer's bookkeeping)
        det_extent = [-45., 45., -45., 45.]
        RT.detector_image(extent=det_extent, _keep_on_device=True)  # first call pays one-off table upload / lazy init
        torch.cuda.synchronize()
        td0 = time.perf_counter()
        img = RT.detector_image(extent=det_extent, _keep_on_device=True)
        torch.cuda.synchronize()
    t_det = time.perf_counter() - td0
    hist = img._dev
    t_red = 0.0
    if use_dist:
        torch.cuda.synchronize()
        tr0 = time.perf_counter()
        D.allreduce_images([hist])  # the one exchange step: all-reduce of the (Ny, Nx, 4) f64 histogram (its lit window)
        torch.cuda.synchronize()
        t_red = time.perf_counter() - tr0
        cnt = D.allreduce_counters(RT._msgs, device=None if on_host else dev)
        assert cnt.sum() >= RT._msgs.sum()
    total_power = float(hist[..., 3].sum().item())

    other = None
    if world == 1 and not args.no_pol:  # same scene without polarisation tracking, outside the timed region
        other = secondary_no_pol(ot, scenes, lib, N, args.steps)

    # ---- the other BASELINE configurations, after the timed region and with the headline's storage released ----
    cpu = None
    if rank == 0 and not args.skip_cpu and world == 1:  # CPU leg on rank 0 at N = 1 only (reads the first sections of the
        cpu = cpu_baseline(RT, scene, args.cpu_seconds)  # headline's storage: before that is released)
    cfgs = sharded = sharded5 = None
    if not args.skip_configs:
        del img, hist
        RT.rays.__init__()
        _release(torch, lib)
        if world == 1:
            cfgs = configs_block(ot, torch, lib, scenes)
        sharded = c4_sharded_block(ot, torch, lib, scenes, dist if use_dist else None, use_dist, backend, dev, world)
        sharded5 = c5_sharded_block(ot, torch, lib, scenes, dist if use_dist else None, use_dist, backend, dev, world)

    if rank == 0:
        pol = not args.no_pol
        bytes_per_ray = nt * (48 if pol else 36) + 28  # compulsory RayStorage traffic of trace()
        b_trace = N * bytes_per_ray
        achieved = b_trace / (kernel_ms * 1e-3) / 1e9
        pmc, pmc_file = committed_profile("trace_kernel_pmc.json", pol, N)
        sq, sq_file = committed_profile("trace_kernel_sq.json", pol, N)
        traffic = None
        if pmc:  # FETCH_SIZE doubled: the gfx950 correction MI355X_MICROARCH.md prescribes; KB -> bytes
            traffic = (2 * float(np.median(pmc["FETCH_SIZE"])) + float(np.median(pmc["WRITE_SIZE"]))) * 1024
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": None if pmc_file is None else f"{pmc_file} (rocprofv3 --pmc passes of this command, "
                                                                   "not collected in this run)",
                "kernel": "trace_kernel (ot_generate_and_trace_host inside Raytracer.trace)", "kernel_ms": kernel_ms,
                "kernel_ms_source": "HIP events recorded by the library right around the kernel on its launch stream, "
                                    "every timed step",
                "algorithmic_bytes_per_launch": b_trace}
        if sq and N == 10_000_000:
            valu = sq["SQ_INSTS_VALU"] / sq["SQ_WAVES"]
            rate = valu * (N / 64) / (kernel_ms * 1e-3)
            roof.update({"valu_insts_per_wave": valu, "salu_insts_per_wave": sq["SQ_INSTS_SALU"] / sq["SQ_WAVES"],
                         "valu_frac": rate / F64_ISSUE_PEAK, "valu_peak_wave_insts_per_s": F64_ISSUE_PEAK,
                         "valu_source": f"{sq_file} (SQ_INSTS_VALU / SQ_WAVES); peak: profiles/r1/op_rate.txt"})
        out = {
            "metric": "ray-surface-intersections/s",
            "value": world * N * M * args.steps / t_max,
            "unit": "ray-surface-intersections/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "settle_launches": settle_launches,  # untimed set-up before the warm-up steps (clock ramp), see above
            "settle_kernel_ms_first_last": [settle_hist[0], settle_hist[-1]],
            "ms_per_step": 1e3 * t_max / args.steps,
            "ms_per_surface_per_Mray": 1e3 * t_max / args.steps / M / (N / 1e6),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "double_gauss.py geometry (15 tracing surfaces, 5 point sources, FDC lines), "
                                   f"{N} rays per GPU, polarisation {'on' if pol else 'off'}, Raytracer.trace(N) with "
                                   "on-device generation",
                       "rays_per_gpu": N, "surfaces": M, "sections": nt, "parallelism": f"ray-sharded x{world}",
                       "backend": backend if use_dist else None,
                       "ranks_per_device": -(-world // ndev)},
            "roofline": roof,
            "api_overhead_ms": 1e3 * t_local / args.steps - kernel_ms,
            "raw_kernel": {"ms_per_launch": raw_ms, "value": N * M / (raw_ms * 1e-3),
                           "what": "back-to-back asynchronous ot_generate_and_trace launches, no host work between"},
            "detector": {"rays_per_s": N / t_det, "ms": 1e3 * t_det, "allreduce_ms": 1e3 * t_red,
                         "image_power_all_ranks": total_power},
        }
        if cfgs is not None:
            out["configs"] = cfgs
        if sharded is not None:
            out["c4_sharded"] = sharded
        if sharded5 is not None:
            out["c5_sharded"] = sharded5
        if other is not None:
            out["no_pol"] = other  # BASELINE config C2 is quoted with polarisation on and off: the other setting
        if cpu is not None:
            out["cpu_baseline"] = cpu
            out["cpu_baseline"]["gpu_over_cpu_core"] = out["value"] / out["cpu_baseline"]["value_one_core"]
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
