#!/usr/bin/env python3
"""Soak run of tests/test_gpu_random_scenes.py over many more seeds than the suite holds (GPU trace against the oracle on the
same rays: alive masks and counters bit-exact).  Usage: soak_random_scenes.py [first_seed] [count]"""
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import _pytest.outcomes
import test_gpu_random_scenes as T

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 500
ok = skipped = 0
bad = []
for seed in range(first, first + count):
    for fn in (T.test_random_scene_matches_oracle, T.test_random_scene_with_hurb_matches_oracle):
        try:
            fn(seed)
            ok += 1
        except _pytest.outcomes.Skipped:
            skipped += 1
        except AssertionError as e:
            bad.append((fn.__name__, seed, str(e)[:200]))
    if (seed - first) % 50 == 49:
        print(f"seeds {first}..{seed}: {ok} ok, {skipped} skipped (colliding geometry), {len(bad)} mismatches", flush=True)
print(f"TOTAL {ok} ok, {skipped} skipped, {len(bad)} mismatches")
for b in bad:
    print("MISMATCH", b)
sys.exit(1 if bad else 0)
