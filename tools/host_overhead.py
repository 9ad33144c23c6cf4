import sys, time, ctypes as C
sys.path[:0]=['/root/repo','/root/repo/tests','/root/repo/tools']
sys.argv=[sys.argv[0],'NONE']
import torch, optrace_amd as ot
from optrace_amd import _capi
import bench_configs as bc
lib=_capi.load_library()
ms=C.c_double()
for name in ("C2 double gauss","C3","C4","A2"):
    key=[k for k in bc.CONFIGS if k.startswith(name)][0]
    build,N=bc.CONFIGS[key]
    with ot.global_options.no_warnings():
        RT=build(ot)
        for _ in range(15): RT.trace(N)
        _capi.check(lib.ot_scene_set_timing(RT._scene_handle,1))
        ws,ks=[],[]
        for _ in range(15):
            torch.cuda.synchronize(); t0=time.perf_counter(); RT.trace(N); torch.cuda.synchronize(); ws.append(1e3*(time.perf_counter()-t0))
            _capi.check(lib.ot_scene_last_trace_ms(RT._scene_handle,C.byref(ms))); ks.append(ms.value)
    print(f"{key:30s} wall min {min(ws):.3f} med {sorted(ws)[7]:.3f}   kernel min {min(ks):.3f} med {sorted(ks)[7]:.3f}   host {sorted(ws)[7]-sorted(ks)[7]:.3f} ms")
    del RT; torch.cuda.empty_cache()
