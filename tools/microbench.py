#!/usr/bin/env python3
"""Times the individual C-ABI entry points on the bench scene (GPU box only)."""
import ctypes as C
import sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import torch
import optrace_amd as ot
from optrace_amd import _capi
from optrace_amd._device import ptr, stream_ptr
import scenes

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
scene_name = sys.argv[2] if len(sys.argv) > 2 else "double_gauss"
no_pol = "nopol" in sys.argv
lib = _capi.load_library()
with ot.global_options.no_warnings():
    RT = scenes.SCENES[scene_name][0](ot, no_pol=no_pol, seed=1)
    RT._geometry_checks()
    sc = RT._compile()
    RT.rays.init(RT.ray_sources, N, sc.nt, RT.no_pol)
rays = RT.rays._rays_struct(); tab = RT.rays._source_table(); rng = RT.rays._source_ranges()
msgs = torch.zeros(5 * sc.nt + 1, dtype=torch.int64, device="cuda")

def timeit(f, reps=30):
    for _ in range(12): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

gen = lambda: _capi.check(lib.ot_rays_generate(tab.handle, rng, len(rng), 5, int(RT.no_pol), C.byref(rays), stream_ptr()))
fused = lambda: _capi.check(lib.ot_generate_and_trace(RT._scene_handle, tab.handle, rng, len(rng), 5, C.byref(rays), ptr(msgs), stream_ptr()))
t_gen = timeit(gen)
s0 = RT.rays._dev["s"].clone()
def trace_only():
    RT.rays._dev["s"].copy_(s0)
    _capi.check(lib.ot_trace(RT._scene_handle, C.byref(rays), None, 5, ptr(msgs), stream_ptr()))
t_copy = timeit(lambda: RT.rays._dev["s"].copy_(s0))
t_trace = timeit(trace_only) - t_copy
t_fused = timeit(fused)
M = sc.nt - 2
print(f"{scene_name} N={N} M={M} pol={not RT.no_pol}: generate {t_gen:.3f} ms | trace(injected) {t_trace:.3f} ms | fused {t_fused:.3f} ms"
      f" | {N*M/t_fused/1e6:.1f} G ray-surf/s")

# ---- detector stage -------------------------------------------------------------------------------------
if "det" in sys.argv:
    fused(); torch.cuda.synchronize()
    RT.rays.lock(); RT._last_trace_snapshot = RT.tracing_snapshot()
    import time
    ext = None if "auto" in sys.argv else [-45., 45., -45., 45.]
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with ot.global_options.no_warnings():
            img = RT.detector_image(extent=ext, _keep_on_device=True)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print(f"detector_image rep{rep}: {1e3*(t1-t0):.2f} ms  power {float(img._dev[...,3].sum()):.5f} shape {tuple(img._dev.shape)}")
