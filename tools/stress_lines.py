#!/usr/bin/env python3
"""Stress: fused LINES kernel vs generate + formula kernel on the same seed, many repetitions; prints any difference."""
import ctypes as C
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import numpy as np
import torch

import optrace_amd as ot
from optrace_amd import _capi
from optrace_amd._device import ptr, stream_ptr
import scenes

lib = _capi.load_library()
N = 300_000
bad_total = 0
junk = []
for rep in range(40):
    with ot.global_options.no_warnings():
        RT = scenes.double_gauss(ot, no_pol=False, seed=77)
        RT.trace(N)
        fused = {k: RT.rays._dev[k].clone() for k in ("p", "s", "w", "n", "wl", "pol")}
        rays = RT.rays._rays_struct()
        tab, rng = RT.rays._source_table(), RT.rays._source_ranges()
        # poison s before regenerating: stale values show up
        _capi.check(lib.ot_rays_generate(tab.handle, rng, len(rng), 77, 0, C.byref(rays), stream_ptr()))
        s_gen = RT.rays._dev["s"].clone()
        msgs = torch.zeros(5 * RT.rays.Nt + 1, dtype=torch.int64, device="cuda")
        _capi.check(lib.ot_trace(RT._scene_handle, C.byref(rays), None, 77, ptr(msgs), stream_ptr()))
        torch.cuda.synchronize()
    for k, t in fused.items():
        a, b = t.cpu().numpy(), RT.rays._dev[k].cpu().numpy()
        bad = np.nonzero(~((a == b) | (np.isnan(a) & np.isnan(b))))[0]
        if bad.shape[0]:
            bad_total += 1
            g = s_gen.cpu().numpy()
            print(f"rep {rep} {k}: {bad.shape[0]} differ, idx {bad[:6]} (mod N {bad[:6] % N}), fused {a[bad[:3]]} formula {b[bad[:3]]}"
                  + (f" generated {g[bad[:3]]}" if k == "s" else ""), flush=True)
            if k == "s":
                w = RT.rays._dev["w"].view(RT.rays.Nt, N)[:, torch.from_numpy(bad[:3] % N).cuda()].cpu().numpy()
                print("   weights along those rays:", w.T)
    junk.append(torch.empty(50_000_000 + rep * 1000, dtype=torch.float64, device="cuda").fill_(float(rep)))  # churn the allocator
    if len(junk) > 3:
        junk.pop(0)
print("repetitions with differences:", bad_total)
