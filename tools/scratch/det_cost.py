import sys, pathlib, time
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "tests"), str(ROOT / "tools")]
import numpy as np, torch
import optrace_amd as ot, scenes
sys.argv = [sys.argv[0], "NONE"]
import bench_configs as bc

def timeit(f, n=3):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3

with ot.global_options.no_warnings():
    RT = bc.c4(ot); N = 200_000_000
    RT.trace(N); torch.cuda.synchronize()
    print("C4 traced", flush=True)
    for ext in ([-8, 8, -8, 8], None):
        ms = timeit(lambda: RT.detector_image(extent=ext, _keep_on_device=True))
        print(f"C4 detector_image extent={ext}: {ms:.2f} ms  {N / ms * 1e3:.3e} rays/s  {N * 56 / ms / 1e6:.0f} GB/s algorithmic", flush=True)
    ms = timeit(lambda: RT.detector_spectrum())
    print(f"C4 detector_spectrum: {ms:.2f} ms")
    del RT; torch.cuda.empty_cache()
    RT = bc.c3(ot); N = 50_000_000
    RT.trace(N); torch.cuda.synchronize()
    for ext in (None,):
        ms = timeit(lambda: RT.detector_image(extent=ext, _keep_on_device=True))
        print(f"C3 detector_image (sphere, Equidistant) extent={ext}: {ms:.2f} ms  {N / ms * 1e3:.3e} rays/s  {N * 56 / ms / 1e6:.0f} GB/s algorithmic", flush=True)
    del RT; torch.cuda.empty_cache()
    RT = scenes.hurb_slit_lens(ot, seed=51); N = 100_000_000
    RT.trace(N); torch.cuda.synchronize()
    ms = timeit(lambda: RT.detector_image(_keep_on_device=True))
    print(f"C5 detector_image auto extent: {ms:.2f} ms  {N / ms * 1e3:.3e} rays/s  {N * 56 / ms / 1e6:.0f} GB/s algorithmic", flush=True)
