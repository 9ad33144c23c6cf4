#!/usr/bin/env python3
"""Does an RCCL collective slow the trace kernels that follow it?  One rank, NCCL backend: kernel ms (library events) of
full-size traces before a dist.barrier(), right after it, and after a pause.  Run under torch.distributed.run."""
import ctypes as C, os, pathlib, sys, time
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import torch, torch.distributed as dist
torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
import optrace_amd as ot
from optrace_amd import _capi
import scenes
lib = _capi.load_library()
ms = C.c_double()
with ot.global_options.no_warnings():
    RT = scenes.double_gauss(ot, seed=None)
    RT.trace(100_000)
    _capi.check(lib.ot_scene_set_timing(RT._scene_handle, 1))

    def run(n):
        out = []
        for _ in range(n):
            RT.trace(10_000_000)
            _capi.check(lib.ot_scene_last_trace_ms(RT._scene_handle, C.byref(ms)))
            out.append(round(ms.value, 3))
        return out
    run(40)
    print("before barrier      ", run(10))
    dist.barrier(); torch.cuda.synchronize()
    print("after dist.barrier()", run(30))
    time.sleep(0.5)
    print("after 0.5 s pause   ", run(10))
    t = torch.ones(1024, device="cuda")
    dist.all_reduce(t); torch.cuda.synchronize()
    print("after all_reduce    ", run(10))
    torch.cuda.synchronize(); time.sleep(0.05)
    print("after 50 ms pause   ", run(10))
    run(30)
    for gap in (0.0002, 0.001, 0.003, 0.01):
        run(30)
        time.sleep(gap)
        print(f"after {1e3*gap:5.1f} ms pause  ", run(8))
    run(30)
    dist.barrier(); torch.cuda.synchronize()
    print("after a WARM barrier", run(8))
    run(30)
    t0 = time.perf_counter(); dist.barrier(); torch.cuda.synchronize(); print("warm barrier takes ms", 1e3 * (time.perf_counter() - t0))
dist.destroy_process_group()
