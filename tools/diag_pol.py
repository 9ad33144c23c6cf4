#!/usr/bin/env python3
"""Diagnostic: norm of the polarisation vectors of living rays per section, C2 at 1e7 rays."""
import pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import numpy as np, torch
import optrace_amd as ot, scenes
N = 10_000_000
with ot.global_options.no_warnings():
    RT = scenes.double_gauss(ot, seed=11)
    RT.trace(N)
d = RT.rays._dev
pol = d["pol"].view(3, 17, N)
w = d["w"].view(17, N)
for sec in range(17):
    live = w[sec] > 0
    pn = (pol[:, sec].double() ** 2).sum(dim=0).sqrt()
    dev = (pn - 1).abs()
    dev = torch.where(live, dev, torch.zeros_like(dev))
    nanc = int((torch.isnan(pn) & live).sum())
    worst = int(torch.argmax(torch.nan_to_num(dev, nan=1e9)))
    print(sec, "max dev", float(torch.nan_to_num(dev, nan=1e9).max()), "n>1e-4", int((dev > 1e-4).sum()), "nan", nanc,
          "worst ray", worst, pol[:, sec, worst].tolist(), float(w[sec, worst]))
    if float(torch.nan_to_num(dev, nan=1e9).max()) > 1e-4:
        s = d["s"].view(3, N)[:, worst].tolist()
        print("   s_final", s, "pol history", [pol[:, k, worst].tolist() for k in range(17)], "w", w[:, worst].tolist())
        print("   p", d["p"].view(3, 17, N)[:, :, worst].T.tolist())
        break
