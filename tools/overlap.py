#!/usr/bin/env python3
"""Share of a file's code lines (> 25 characters, comments and blank lines left out) that occur verbatim somewhere in the
reference package -- the copy check of the review.  Build container only (reads /root/reference).
Usage: overlap.py [files...]   (default: every .py under optrace_amd/)"""
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
REF = pathlib.Path("/root/reference/optrace")


def code_lines(path):
    out = []
    for line in path.read_text(errors="ignore").splitlines():
        t = line.strip()
        if len(t) > 25 and not t.startswith("#"):
            out.append(t)
    return out


ref = set()
for f in REF.rglob("*.py"):
    ref.update(code_lines(f))
files = [pathlib.Path(a) for a in sys.argv[1:]] or sorted((ROOT / "optrace_amd").rglob("*.py"))
show = "-v" in sys.argv
for f in files:
    if str(f) == "-v":
        continue
    lines = code_lines(f)
    hit = [t for t in lines if t in ref]
    if lines:
        print(f"{100 * len(hit) / len(lines):5.1f} %  {len(hit):4d} / {len(lines):4d}  {f.relative_to(ROOT) if f.is_absolute() else f}")
    if show:
        for t in hit:
            print("      ", t)
