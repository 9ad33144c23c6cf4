#!/usr/bin/env python3
"""A/B of library builds on ONE box: runs bench.py once per library and round (interleaved, so that clock drift of the
box hits all arms alike) and prints kernel ms / step ms per arm.  Usage: ab_bench.py [--rounds R] name=path.so ..."""
import json
import os
import pathlib
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
args = sys.argv[1:]
rounds = 2
if args and args[0] == "--rounds":
    rounds, args = int(args[1]), args[2:]
extra = []
if "--" in args:
    k = args.index("--")
    args, extra = args[:k], args[k + 1:]
arms = [a.split("=", 1) for a in args]
res = {n: [] for n, _ in arms}
for r in range(rounds):
    for name, path in arms:
        env = dict(os.environ, OPTRACE_AMD_LIB=str((ROOT / path).resolve()))
        out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--skip-cpu", "--steps", "40", "--warmup", "30"] + extra,
                             env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(name, "FAILED", out.stderr[-800:])
            continue
        d = json.loads(line[-1])
        res[name].append((d["roofline"]["kernel_ms"], d["ms_per_step"], d["no_pol"]["ms_per_step"] if "no_pol" in d else 0))
        print(f"round {r} {name:12s} kernel {d['roofline']['kernel_ms']:.4f} ms  step {d['ms_per_step']:.4f} ms  "
              f"no_pol step {res[name][-1][2]:.4f}", flush=True)
for name, v in res.items():
    if v:
        print(f"{name:12s} mean kernel {sum(x[0] for x in v)/len(v):.4f}  min {min(x[0] for x in v):.4f}   "
              f"no_pol {sum(x[2] for x in v)/len(v):.4f}")
