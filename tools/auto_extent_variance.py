#!/usr/bin/env python3
"""Wall time of repeated detector_image(extent=None) calls on C4, C5, C2 in one process (looks for outliers).
With `prof` as argument the per-call kernel verdict is printed too (RT._last_render_path is not recorded; use rocprofv3)."""
import pathlib, sys, time
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests"), str(ROOT / "tools")]
sys.argv = [sys.argv[0], "NONE"]
import torch
import optrace_amd as ot
import scenes
import bench_configs as bc
import os
if os.environ.get("DENSE"):
    ot.Raytracer.COMPACT_HITS_FROM = 1 << 60

def run(name, RT, N, reps=8):
    RT.trace(N)
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if os.environ.get("SPLIT"):
            spec = dict(detector_index=0, source_index=None, extent=None, projection_method="Equidistant",
                        compact=RT.rays.N >= RT.COMPACT_HITS_FROM)
            hits = RT._hit_detectors("Detector Image", [spec])[0]
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            RT._image_from_hits(hits, 0, None, None)
            torch.cuda.synchronize()
            ts.append(1e3 * (t1 - t0)); ts.append(1e3 * (time.perf_counter() - t1))
            continue
        RT.detector_image(_keep_on_device=True)
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    print(f"{name:4s} " + " ".join(f"{t:7.2f}" for t in ts), flush=True)

with ot.global_options.no_warnings():
    for rnd in range(2):
        RT = bc.c4(ot); run("C4", RT, 200_000_000); del RT; torch.cuda.empty_cache()
        RT = bc.c3(ot); run("C3", RT, 50_000_000); del RT; torch.cuda.empty_cache()
        RT = scenes.hurb_slit_lens(ot, seed=51); run("C5", RT, 100_000_000); del RT; torch.cuda.empty_cache()
        RT = scenes.double_gauss(ot, seed=1); run("C2", RT, 10_000_000); del RT; torch.cuda.empty_cache()
