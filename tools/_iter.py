import pathlib, sys, time
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests"), str(ROOT / "tools")]
import torch
import optrace_amd as ot
sys.argv = [sys.argv[0], "NONE"]
import bench_configs as bc
with ot.global_options.no_warnings():
    RT, N = bc.c4(ot), 200_000_000
    pos = [[0, 0, z] for z in (30, 32, 34, 36, 38, 39.5)]
    for ext in ([[-8, 8, -8, 8]] * 6, None):
        RT.iterative_render(N, pos=pos, extent=ext); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); RT.iterative_render(N, pos=pos, extent=ext); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        print(f"C4 iterative_render 6 positions extent={'user' if ext else 'auto'}: {min(ts):.2f} ms")
