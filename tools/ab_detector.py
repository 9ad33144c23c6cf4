#!/usr/bin/env python3
"""A/B of library builds for the detector stage: C4 traced once per arm, detector_image with a user extent timed.
Usage: ab_detector.py name=path.so ..."""
import os, pathlib, subprocess, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent
code = r'''
import sys, time, pathlib
ROOT = pathlib.Path(%r)
sys.path[:0] = [str(ROOT), str(ROOT / "tests"), str(ROOT / "tools")]
sys.argv = [sys.argv[0], "NONE"]
import torch, optrace_amd as ot, bench_configs as bc
build, N = bc.CONFIGS[[k for k in bc.CONFIGS if k.startswith("C4")][0]]
with ot.global_options.no_warnings():
    RT = build(ot); RT.trace(N)
    for ext in ([-8., 8., -8., 8.], None):
        for _ in range(3): RT.detector_image(extent=ext, _keep_on_device=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): RT.detector_image(extent=ext, _keep_on_device=True)
        torch.cuda.synchronize(); print("extent", "user" if ext else "auto", "%%.3f ms" %% ((time.perf_counter() - t0) * 100))
''' % str(ROOT)
for rnd in range(2):
    for arm in sys.argv[1:]:
        name, path = arm.split("=", 1)
        env = dict(os.environ, OPTRACE_AMD_LIB=str((ROOT / path).resolve()))
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print(rnd, name, " | ".join(l for l in out.stdout.splitlines() if l.startswith("extent")) or out.stderr[-500:], flush=True)
