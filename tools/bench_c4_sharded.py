#!/usr/bin/env python3
"""BASELINE config 4 as stated: image_render_many_rays.py (RGB image source -> biconvex lens -> detector, no_pol), rays
sharded over the ranks, six detector positions, one RCCL histogram reduce -- `distributed.sharded_iterative_render`.

    python tools/bench_c4_sharded.py [--rays-per-gpu 25000000] [--steps 5] [--extent user|auto] [--backend nccl|gloo]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/bench_c4_sharded.py ...

One step = one sharded render of N = ranks x rays-per-gpu rays (weak scaling; BASELINE quotes 2e8 rays on 8 GPUs = 2.5e7 per
GPU): every rank traces its shard once, bins every chunk into all six positions, then the histograms are all-reduced.
Rank 0 prints one JSON line: rays/s over the whole job, ms per render, and the split trace+binning / exchange as measured on
rank 0.  (bench.py stays the headline benchmark -- config 2's Raytracer.trace; this is the tool for config 4's number when a
multi-GPU node is at hand.  With fewer devices than ranks use --backend gloo: the ranks share devices, collectives on host
copies.)"""
import argparse
import json
import os
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests"), str(ROOT / "tools")]
ap = argparse.ArgumentParser()
ap.add_argument("--rays-per-gpu", type=int, default=25_000_000)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--extent", default="user", choices=["user", "auto"])
ap.add_argument("--backend", default="nccl")
args = ap.parse_args()
sys.argv = [sys.argv[0], "NONE"]  # (bench_configs parses its own argv)

import numpy as np
import torch
import torch.distributed as dist

rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
torch.cuda.set_device(local)
use_dist = "RANK" in os.environ and "MASTER_PORT" in os.environ
if use_dist:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(args.backend)

import optrace_amd as ot
from optrace_amd import distributed as D
import bench_configs as bc

N = world * args.rays_per_gpu
pos = [[0, 0, z] for z in (30., 32., 34., 36., 38., 39.5)]  # image_render_many_rays.py:39-41: a sweep through the image plane
ext = [[-8., 8., -8., 8.]] * len(pos) if args.extent == "user" else None


def sync():
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()


with ot.global_options.no_warnings():
    RT = bc.c4(ot)
    D.sharded_iterative_render(RT, N, pos=pos, extent=ext, base_seed=1)  # warm-up: scene, tables, allocator pools
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        imgs = D.sharded_iterative_render(RT, N, pos=pos, extent=ext, base_seed=100 + 10 * k)
    sync()
    t = (time.perf_counter() - t0) / args.steps
    # the exchange alone, for the record: the lit window of the six histograms once more
    sync()
    t1 = time.perf_counter()
    D.allreduce_images([im._dev for im in imgs])
    sync()
    t_red = time.perf_counter() - t1
tt = torch.tensor([t], dtype=torch.float64)
if use_dist:
    tt = tt.cuda() if args.backend == "nccl" else tt
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
if rank == 0:
    print(json.dumps({"metric": "rays rendered/s (config 4: six detector positions, ray-sharded, one histogram reduce)",
                      "value": N / float(tt.item()), "unit": "rays/s", "n_gpus": world, "rays_total": N,
                      "rays_per_gpu": args.rays_per_gpu, "ms_per_render": 1e3 * float(tt.item()), "extent": args.extent,
                      "positions": len(pos), "image_shapes": [list(im._dev.shape) for im in imgs],
                      "histogram_allreduce_ms": 1e3 * t_red, "image_power": [float(im.power()) for im in imgs],
                      "backend": args.backend if use_dist else None, "scaling": "weak"}), flush=True)
if use_dist:
    dist.barrier()
    dist.destroy_process_group()
