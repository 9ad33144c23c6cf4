#!/usr/bin/env python3
"""detector_image(extent=None): the one-pass form (`Raytracer._auto_image_one_pass`) against the chain hit list ->
binning, on C4 / C5 / C2 at full size.  Prints both times and the largest difference of the two images."""
import os
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests"), str(ROOT / "tools")]
import numpy as np
import torch

import optrace_amd as ot
import scenes

sys.argv = [sys.argv[0], "NONE"]
import bench_configs as bc


def timeit(f, n=int(os.environ.get("AB_N", "5"))):
    f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3, float(np.median(ts)) * 1e3


def run(name, RT, N):
    RT.trace(N)
    applied = []
    orig = RT._auto_image_one_pass

    def spy(*a, **k):
        img = orig(*a, **k)
        applied.append(img is not None)
        return img

    RT._auto_image_one_pass = spy
    out = {}
    for label, frm in (("one pass", 1), ("chain", 1 << 60))[:1 if os.environ.get("AB_ONLY_ONE_PASS") else 2]:
        ot.Raytracer.AUTO_ONE_PASS_FROM = frm
        out[label] = timeit(lambda: RT.detector_image(_keep_on_device=True))
        out[label + " image"] = RT.detector_image()
    if os.environ.get("AB_KNOWN"):
        e = [float(v) for v in out["one pass image"]._extent0]
        t = timeit(lambda: RT.detector_image(extent=e, _keep_on_device=True))
        h = RT._hit_detectors("", [dict(detector_index=0, source_index=None, extent=None, compact=True)])[0]
        print(f"{name:4s} extent given: {t[0]:6.2f} ms (median {t[1]:6.2f});  valid hits {int(h[2][1].sum().item()):,d} of {N:,d} rays", flush=True)
    if "chain" not in out:
        print(f"{name:4s} N={N:11,d}  one pass {out['one pass'][0]:6.2f} ms (median {out['one pass'][1]:6.2f})  applied={all(applied) and bool(applied)}", flush=True)
        return
    a, b = out["one pass image"], out["chain image"]
    same_extent = np.array_equal(a.extent, b.extent)
    A, B = a._data, b._data
    diff = np.abs(A - B).max() / np.abs(B).max() if A.shape == B.shape else float("nan")
    lit = A.shape == B.shape and np.array_equal(A[..., 3] != 0, B[..., 3] != 0)
    print(f"{name:4s} N={N:11,d}  one pass {out['one pass'][0]:6.2f} ms (median {out['one pass'][1]:6.2f})   chain "
          f"{out['chain'][0]:6.2f} ms (median {out['chain'][1]:6.2f})   applied={all(applied) and bool(applied)}  "
          f"shape={A.shape}  same extent={same_extent}  same pixels lit={lit}  max rel diff={diff:.2e}", flush=True)


with ot.global_options.no_warnings():
    import os
    sel = os.environ.get("AB_CONFIGS", "C4,C5,C2").split(",")
    if os.environ.get("AB_MARGINS"):
        import ast
        ot.Raytracer.AUTO_MARGINS = ast.literal_eval(os.environ["AB_MARGINS"])
    if "C4" in sel:
        run("C4", bc.c4(ot), 200_000_000)
        torch.cuda.empty_cache()
    if "C5" in sel:
        run("C5", scenes.hurb_slit_lens(ot, seed=51), 100_000_000)
        torch.cuda.empty_cache()
    if "C2" in sel:
        run("C2", scenes.double_gauss(ot, seed=1), 10_000_000)
    if "IMG" in sel:  # crossover on the extended-image scene of the tests
        from test_gpu_fused_detector import image_scene
        for n in (2_000_000, 5_000_000, 10_000_000, 20_000_000, 40_000_000):
            RT = image_scene(N=1000)
            run(f"IMG", RT, n)
            del RT
            torch.cuda.empty_cache()
