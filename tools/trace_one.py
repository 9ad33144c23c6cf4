#!/usr/bin/env python3
"""One fused detector image of C4 (2e8 rays) after a warm-up call: run under rocprofv3 --kernel-trace to get the kernel
sequence of a single call."""
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests"), str(ROOT / "tools")]
import torch

import optrace_amd as ot

sys.argv = [sys.argv[0], "NONE"]
import bench_configs as bc

with ot.global_options.no_warnings():
    RT, N = bc.c4(ot), 200_000_000
    RT.trace(N)
    for _ in range(3):
        RT.detector_image(extent=[-8, 8, -8, 8])
        torch.cuda.synchronize()
