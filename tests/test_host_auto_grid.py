"""Host logic of the one-pass image with an automatic extent (`Raytracer._auto_grid`, CPU only): the provisional tile
grid laid over the extent E0 of a sample must serve ANY final extent E that contains E0 -- a tile may span at most 61
pixels of E's image (its LDS window has 64) -- with the reference's extent and pixel-count rules
(render_image.py:224-255 `__fix_extent`, :383-387 pixel counts) in between."""
import numpy as np
import pytest

import optrace_amd as ot
from optrace_amd.render_image import RenderImage

MARGINS = ot.Raytracer.AUTO_MARGINS


def final_grid(extent, limit):
    img = RenderImage(extent=np.array(extent, dtype=np.float64))
    img._limit = limit
    img._fix_extent()
    Nx, Ny = img._pixel_counts()
    return img, Nx, Ny


@pytest.mark.parametrize("limit", [None, 3.0, 40.0])
def test_tiles_fit_the_windows_of_any_larger_extent(limit):
    rng = np.random.default_rng(5)
    planned = checked = 0
    for _ in range(3000):
        cx, cy = rng.uniform(-50, 50, 2)
        sx0 = 10 ** rng.uniform(-3, 2)
        sy0 = sx0 * 10 ** rng.uniform(-0.75, 0.75)  # ratios up to 5.6 either way
        e0 = np.array([cx - sx0 / 2, cx + sx0 / 2, cy - sy0 / 2, cy + sy0 / 2])
        plan = ot.Raytracer._auto_grid(e0, limit, None, MARGINS)
        ratio = max(sx0, sy0) / min(sx0, sy0)
        s0 = final_grid(e0, limit)[0].s
        ratio_fixed = max(s0) / min(s0)  # side ratio of the sample's own image (`limit` widens both sides)
        if ratio > RenderImage.MAX_IMAGE_RATIO / 1.2:
            assert plan is None, "very elongated sample extents are left to the hit-list chain"
            continue
        assert plan is not None
        planned += 1
        (X0, Y0, tw, th, tx, ty), tw2, th2 = plan
        assert (tw, th) == (tw2, th2) and tx * ty <= 2048 and tx >= 1 and ty >= 1
        # the grid covers the sample extent
        assert X0 <= e0[0] and Y0 <= e0[2] and X0 + tx * tw >= e0[1] and Y0 + ty * th >= e0[3]
        # any final extent containing e0 whose image's side ratio grew by less than the 20 % the planner allows for
        for _ in range(6):
            gx, gy = 1 + rng.uniform(0, 1.0) * rng.integers(0, 2), 1 + rng.uniform(0, 1.0) * rng.integers(0, 2)
            sx, sy = sx0 * gx, sy0 * gy
            ox, oy = rng.uniform(0, sx - sx0), rng.uniform(0, sy - sy0)
            E = [e0[0] - ox, e0[0] - ox + sx, e0[2] - oy, e0[2] - oy + sy]
            img, Nx, Ny = final_grid(E, limit)
            if max(img.s) / min(img.s) > 1.2 * ratio_fixed:  # (beyond the planner's allowance: the host check declines)
                continue
            checked += 1
            assert tw * Nx / img.s[0] <= 60.001, (e0, E, Nx, tw)  # (side lengths are differences of coordinates: ~1e-11 relative here)
            assert th * Ny / img.s[1] <= 60.001, (e0, E, Ny, th)
    assert planned > 2000 and checked > 4000


def test_degenerate_sample_extents_are_declined():
    for e0 in ([1.0, 1.0, -2.0, 3.0], [0.0, 4.0, 2.0, 2.0], [0.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 4.5]):
        assert ot.Raytracer._auto_grid(np.array(e0), None, None, MARGINS) is None
    # a square: 5 % margin per side = 19 x 19 tiles, the most the tile kernel with line buffers takes (csrc: OT_LB_MAXK)
    grid, tw, th = ot.Raytracer._auto_grid(np.array([-1.0, 1.0, -1.0, 1.0]), None, None, MARGINS)
    assert (grid[4], grid[5]) == (19, 19) and grid[0] == pytest.approx(-1.1) and grid[1] == pytest.approx(-1.1)
    # without that entry: 50 % would take 33 x 33 tiles, more than the 1024 of the plain kernel's faster form; 30 % fits
    grid, _, _ = ot.Raytracer._auto_grid(np.array([-1.0, 1.0, -1.0, 1.0]), None, None, ((0.5, 1024), (0.3, 1024)))
    assert (grid[4], grid[5]) == (27, 27) and grid[0] == pytest.approx(-1.6)
    assert tw == pytest.approx(60 * 2.0 / 945) and th == tw


def test_margin_preference_and_tile_budget():
    # ratio 4 image: 4725 x 945 pixels -> 79 x 16 tiles without margin; the 1024-tile forms do not fit, a 2048 one does
    grid, tw, th = ot.Raytracer._auto_grid(np.array([0.0, 4.0, 0.0, 1.0]), None, None, MARGINS)
    assert 1024 < grid[4] * grid[5] <= 2048
    assert ot.Raytracer._auto_grid(np.array([0.0, 4.0, 0.0, 1.0]), None, None, ((0.5, 1024),)) is None
