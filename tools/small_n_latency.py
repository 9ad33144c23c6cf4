#!/usr/bin/env python3
"""Latency of the path at the reference's own sizes (BASELINE config 1: 1e5 rays through one lens; 1e6): trace, detector image with
automatic and given extent, spectrum, iterative_render -- wall time per call, median of 20."""
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import torch

import optrace_amd as ot
import scenes


def med(f, n=20):
    f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    return sorted(ts)[n // 2]


with ot.global_options.no_warnings():
    for N in (100_000, 1_000_000):
        RT = scenes.c1_single_lens(ot, seed=1)
        t_trace = med(lambda: RT.trace(N))
        t_auto = med(lambda: RT.detector_image())
        e = [float(v) for v in RT.detector_image()._extent0]
        t_user = med(lambda: RT.detector_image(extent=e))
        t_data = med(lambda: RT.detector_image()._data)
        t_spec = med(lambda: RT.detector_spectrum())
        t_src = med(lambda: RT.source_image())
        t_iter = med(lambda: RT.iterative_render(N), n=10)
        print(f"C1 N={N:>9,d}: trace {t_trace:.3f} ms | detector_image auto {t_auto:.3f}  given extent {t_user:.3f}  + host copy "
              f"{t_data:.3f} | detector_spectrum {t_spec:.3f} | source_image {t_src:.3f} | iterative_render {t_iter:.3f}", flush=True)
