#!/usr/bin/env python3
"""Soak run of the render-only chunks over random systems (tests/scenes.py::random_scene, with and without HURB): per seed
(1) the records of a render-only trace against the stored sections (tests/test_gpu_render_only.py::check_tail_records),
(2) `iterative_render` with random chunk sizes, merged last chunk, 1-3 positions, user or automatic extents, against the same
chunks through the ray storage (same pixels lit, sums to 1e-7, counters equal).
Usage: soak_render_only.py [first_seed] [count]"""
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import numpy as np
import _pytest.outcomes

import optrace_amd as ot
import scenes
import test_gpu_render_only as T
from test_gpu_fused_detector import same_image

first = int(sys.argv[1]) if len(sys.argv) > 1 else 7000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
ok = skipped = 0
bad = []


def render_case(seed, hurb):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(20_000, 300_000))
    n_chunks = int(rng.integers(2, 5))  # N // n chunks, the last one takes the rest (the reference's rule, raytracer.py:1216-1217)
    N = n * n_chunks + int(rng.integers(0, n))
    out = {}
    for mode in (True, False):
        RT = scenes.random_scene(ot, seed, seed=seed, use_hurb=hurb)
        RT.add(ot.Detector(ot.RectangularSurface(dim=[20, 20]), pos=[0, 0, 110]))
        RT.trace(1000)
        if RT.geometry_error:
            raise _pytest.outcomes.Skipped("collides")
        di = len(RT.detectors) - 1
        K = int(np.random.default_rng(seed + 1).integers(1, 4))
        pos = [[0, 0, 110 - 3 * k] for k in range(K)]
        ext = [[-9., 9., -9., 9.]] * K if seed % 2 else None
        if ext is None and not mode:  # (automatic extents: those the render-only form found, from its stored last chunk)
            ext = [list(im._extent0) for im in out[True][0]]
        with T.settings(ITER_RAYS_STEP=n, ITER_RENDER_ONLY=mode, ITER_EXTENT_RAYS=1 << 60):
            imgs = RT.iterative_render(N, detector_index=di, pos=pos, extent=ext)
        out[mode] = (imgs, RT._msgs.copy(), RT.rays.N)
    (a, ma, na), (b, mb, nb) = out[True], out[False]
    assert np.array_equal(ma, mb), "counters"
    assert na == nb
    for x, y in zip(a, b):
        if y._data[..., 3].sum() == 0:
            assert x._data[..., 3].sum() == 0
            continue
        same_image(x, y, tol=1e-7)


with ot.global_options.no_warnings():
    for seed in range(first, first + count):
        for hurb in (False, True):
            for what in ("records", "render"):
                try:
                    if what == "records":
                        T.check_tail_records(scenes.random_scene(ot, seed, seed=seed, use_hurb=hurb),
                                             int(np.random.default_rng(seed).integers(1, 200_000)))
                    else:
                        render_case(seed, hurb)
                    ok += 1
                except _pytest.outcomes.Skipped:
                    skipped += 1
                except AssertionError as e:
                    bad.append((what, seed, hurb, str(e)[:300]))
        if (seed - first) % 20 == 19:
            print(f"seeds {first}..{seed}: {ok} ok, {skipped} skipped (colliding geometry), {len(bad)} mismatches", flush=True)
print(f"TOTAL {ok} ok, {skipped} skipped, {len(bad)} mismatches")
for b in bad:
    print("MISMATCH", b)
sys.exit(1 if bad else 0)
