#!/usr/bin/env python3
"""Register / scratch / occupancy table of every kernel in the HIP library (cross-compiles, no GPU needed).

usage: tools/resource_usage.py [file.hip ...] [-D...]      (default: every translation unit of optrace_amd/csrc)
One line per kernel: VGPRs, AGPRs, SGPR / VGPR spills, scratch bytes per lane, waves per SIMD, static LDS.
"""
import pathlib
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = pathlib.Path(__file__).resolve().parent.parent / "optrace_amd" / "csrc"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-munsafe-fp-atomics",
         "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null"]
KEYS = [("VGPRs", r"\bVGPRs: (\d+)"), ("AGPRs", r"AGPRs: (\d+)"), ("sgpr_spill", r"SGPRs Spill: (\d+)"),
        ("vgpr_spill", r"VGPRs Spill: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
        ("waves", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")]


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    out = r.stdout.split("\n")
    return [re.sub(r"\(.*$", "", o) for o in out[:len(names)]]


def usage(path, extra):
    r = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *extra, str(path)], capture_output=True, text=True, cwd=CSRC)
    rows, cur = [], None
    for line in r.stderr.split("\n"):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        if cur is None:
            continue
        for k, pat in KEYS:
            m = re.search(pat, line)
            if m and k not in cur:
                cur[k] = int(m.group(1))
    return rows


def main():
    extra = [a for a in sys.argv[1:] if a.startswith("-")]
    files = [pathlib.Path(a) for a in sys.argv[1:] if not a.startswith("-")]
    if not files:
        files = sorted(CSRC.glob("*.hip"))
    files = [f if f.is_absolute() or f.exists() else CSRC / f for f in files]
    with ThreadPoolExecutor(4) as ex:
        res = list(ex.map(lambda f: usage(f.resolve(), extra), files))
    print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'sSpill':>6s} {'vSpill':>6s} {'scratch':>7s} {'waves':>5s} {'LDS':>6s}")
    for f, rows in zip(files, res):
        print(f"# {f.name}")
        names = demangle([r["name"] for r in rows])
        for r, n in zip(rows, names):
            print(f"{n[:70]:70s} {r.get('VGPRs', -1):5d} {r.get('AGPRs', 0):5d} {r.get('sgpr_spill', 0):6d} "
                  f"{r.get('vgpr_spill', 0):6d} {r.get('scratch', 0):7d} {r.get('waves', -1):5d} {r.get('lds', 0):6d}")


if __name__ == "__main__":
    main()
