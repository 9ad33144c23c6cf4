#!/usr/bin/env python3
"""Headline benchmark: ray-surface-intersections/s of Raytracer.trace() on the double-Gauss scene.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--rays R] [--no-pol]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one bundle of synthetic rays: on-device ray generation from
the five point sources, all 15 tracing surfaces, all section stores (what `RT.trace(N)` covers in
tests/benchmark.py:81-86 of the reference).  Workload = BASELINE.json configs[1]: double_gauss.py geometry,
10 M rays per GPU, 3 wavelengths (FDC lines), polarisation on.  Ray storage is allocated once and stays
resident in HBM; every step uses a fresh seed.  Multi-GPU: rays are sharded, every rank traces its own
bundle (weak scaling), no data-path collective inside the step; after the timed region the detector
histograms are all-reduced once over RCCL to exercise the exchange step.

Rank 0 prints ONE JSON line (contract in the task description) carrying `roofline` (dominant kernel:
algorithmic bytes / HIP-event kernel time against the 8 TB/s HBM peak) and `cpu_baseline` (the CPU oracle,
a scalar port of the reference, timed on a bounded sample on the host cores).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=30)  # the first ~10 launches run while the clocks settle
    ap.add_argument("--rays", type=int, default=10_000_000, help="rays per GPU and step")
    ap.add_argument("--no-pol", action="store_true")
    ap.add_argument("--skip-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                                                      "the multi-rank code path on a box with fewer GPUs than ranks)")
    return ap.parse_args()


def cpu_baseline(RT, scene, seconds: float) -> dict:
    """Time the CPU oracle (kind "port": oracle/oracle.c, scalar restatement of the reference's NumPy path,
    pinned to it by tests/test_oracle_golden.py) on a bounded sample of the same workload: rays generated
    by the device kernel for this scene, traced through all 15 surfaces on the host cores (rays split across
    threads like the reference splits them, ray_storage.py:147-171; one core is timed as well)."""
    import oracle_bridge as ob  # checker / baseline only
    M = scene.nt - 2
    r = RT.rays
    n_avail = r.N
    threads = int(os.environ.get("OT_CPU_THREADS", min(os.cpu_count() or 1, 16)))

    d, N, nt = r._dev, r.N, scene.nt
    pv = d["p"].view(3, nt, N)  # element (ray, section, component) lives at ray + N * (section + nt * component)

    def run(n, th):
        rays = ob.HostRays(n, scene.nt, RT.no_pol)
        # only the first two sections of the first n rays leave the device (the storage is 8.4 GB)
        p0 = pv[:, 0, :n].t().cpu().numpy()
        dirs = pv[:, 1, :n].t().cpu().numpy() - p0
        s0 = dirs / np.linalg.norm(dirs, axis=1)[:, None]
        pol0 = None if RT.no_pol else d["pol"].view(3, nt, N)[:, 0, :n].t().cpu().numpy()
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


def secondary_no_pol(ot, scenes, lib, N, dev, steps):
    """The bench scene with no_pol=True: {value, ms_per_step, roofline_frac} from `steps` launches after 10 warm-up ones."""
    from optrace_amd import _capi
    from optrace_amd._device import ptr, stream_ptr
    with ot.global_options.no_warnings():
        RT = scenes.double_gauss(ot, no_pol=True, seed=77)
        RT._geometry_checks()
        scene = RT._compile()
        nt = scene.nt
        RT.rays.init(RT.ray_sources, N, nt, True)
    rays, tab, rng = RT.rays._rays_struct(), RT.rays._source_table(), RT.rays._source_ranges()
    msgs = torch.zeros(5 * nt + 1, dtype=torch.int64, device=dev)
    for i in range(10 + steps):
        if i == 10:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        _capi.check(lib.ot_generate_and_trace(RT._scene_handle, tab.handle, rng, len(rng), 500 + i, C.byref(rays),
                                              ptr(msgs), stream_ptr()))
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / steps
    b = N * (nt * 36 + 28)
    return {"value": N * (nt - 2) / t, "ms_per_step": 1e3 * t, "roofline_frac": b / t / 1e9 / HBM_PEAK_GBS,
            "algorithmic_bytes_per_launch": b}


def measured_traffic(pol: bool, N: int):
    """HBM bytes per launch of the trace kernel from the committed rocprofv3 PMC passes (separate --pmc
    FETCH_SIZE / WRITE_SIZE runs of this same command, profiles/<round>/trace_kernel_pmc.json; FETCH_SIZE
    doubled as the gfx950 correction of MI355X_MICROARCH.md prescribes).  None if no profile matches."""
    best = None
    for f in sorted((ROOT / "profiles").glob("*/trace_kernel_pmc.json")):
        try:
            d = json.loads(f.read_text())
            if d.get("rays") == N and d.get("pol") == pol:
                best = (2 * float(np.median(d["FETCH_SIZE"])) + float(np.median(d["WRITE_SIZE"]))) * 1024
        except Exception:
            pass
    return best


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # launched by torch.distributed.run (also with one rank: the same RCCL path as the N > 1 runs)
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        ndev = torch.cuda.device_count()
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank % ndev))
        else:
            dist.init_process_group(backend=args.backend)
        local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    on_host = use_dist and args.backend != "nccl"  # gloo rehearsal: collectives on host copies

    import optrace_amd as ot
    from optrace_amd import _capi
    from optrace_amd._device import ptr, stream_ptr
    from optrace_amd import distributed as D
    import scenes

    lib = _capi.load_library()
    N = args.rays
    with ot.global_options.no_warnings():
        RT = scenes.double_gauss(ot, no_pol=args.no_pol, seed=1000 + rank)
        RT._geometry_checks()
        assert not RT.geometry_error
        scene = RT._compile()
        nt = scene.nt
        M = nt - 2
        RT.rays.init(RT.ray_sources, N, nt, RT.no_pol)
    rays = RT.rays._rays_struct()
    tab = RT.rays._source_table()
    rng = RT.rays._source_ranges()
    msgs = torch.zeros(5 * nt + 1, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream()

    def step(seed):
        _capi.check(lib.ot_generate_and_trace(RT._scene_handle, tab.handle, rng, len(rng), seed, C.byref(rays),
                                              ptr(msgs), stream_ptr()))

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(17 + i)
    barrier()
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev0[i].record(stream)  # same stream the kernel is launched on
        step(1000 + rank + 7919 * i)
        ev1[i].record(stream)
    barrier()
    t_local = time.perf_counter() - t0
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)]))

    t = torch.tensor([t_local], dtype=torch.float64, device="cpu" if on_host else dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    t_max = float(t.item())

    # ---- after the timed region: detector image + the one exchange step (histogram all-reduce) ----------
    RT.rays.lock()
    RT._last_trace_snapshot = RT.tracing_snapshot()
    RT._msgs = msgs.cpu().numpy()[:-1].reshape(5, nt)
    det_extent = [-45., 45., -45., 45.]
    with ot.global_options.no_warnings():
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
        if on_host:
            h = hist.cpu()
            D.allreduce_image(h)
            hist.copy_(h)
        else:
            D.allreduce_image(hist)  # the one exchange step: RCCL all-reduce of the (Ny, Nx, 4) f64 histogram
        torch.cuda.synchronize()
        t_red = time.perf_counter() - tr0
        cnt = D.allreduce_counters(RT._msgs, device=None if on_host else dev)
        assert cnt.sum() >= RT._msgs.sum()
    total_power = float(hist[..., 3].sum().item())

    other = None
    if world == 1 and not args.no_pol:  # same scene without polarisation tracking, outside the timed region
        other = secondary_no_pol(ot, scenes, lib, N, dev, args.steps)

    if rank == 0:
        pol = not args.no_pol
        bytes_per_ray = nt * (48 if pol else 36) + 28  # compulsory RayStorage traffic of trace()
        b_trace = N * bytes_per_ray
        achieved = b_trace / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "ray-surface-intersections/s",
            "value": world * N * M * args.steps / t_max,
            "unit": "ray-surface-intersections/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * t_max / args.steps,
            "ms_per_surface_per_Mray": 1e3 * t_max / args.steps / M / (N / 1e6),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "double_gauss.py geometry (15 tracing surfaces, 5 point sources, FDC lines), "
                                   f"{N} rays per GPU, polarisation {'on' if pol else 'off'}, on-device generation",
                       "rays_per_gpu": N, "surfaces": M, "sections": nt, "parallelism": f"ray-sharded x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(pol, N),
                         "kernel": "trace_kernel (ot_generate_and_trace)", "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": b_trace},
            "detector": {"rays_per_s": N / t_det, "ms": 1e3 * t_det, "allreduce_ms": 1e3 * t_red,
                         "image_power_all_ranks": total_power},
        }
        if world == 1 and not args.no_pol:
            out["no_pol"] = other  # BASELINE config C2 is quoted with polarisation on and off: the other setting
        if not args.skip_cpu and world == 1:  # CPU leg on rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(RT, scene, args.cpu_seconds)
            out["cpu_baseline"]["gpu_over_cpu_core"] = out["value"] / out["cpu_baseline"]["value_one_core"]
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
