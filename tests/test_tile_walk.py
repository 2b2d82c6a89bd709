"""The XCD-aware tile order of the persistent tile kernels (csrc/conv_frag.h: tile_walk), restated: whatever the grid and the tile count, every tile is
computed by exactly one workgroup, XCD x (workgroups b with b % 8 == x) owns one contiguous range, and the workgroups of an XCD walk it side by side
(at any step they hold consecutive tiles).  The kernels' outputs are compared bit for bit elsewhere (-m gpu); this pins the arithmetic."""
import re
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def tile_walk(G, b, ntile):
    if G & 7:
        return b, G, ntile
    x = b & 7
    lo = (x * ntile) >> 3
    end = ((x + 1) * ntile) >> 3
    return lo + (b >> 3), G >> 3, end


def test_the_restatement_is_the_header():
    src = open(os.path.join(ROOT, "skyeye-aerial-object-detection-using-yolo_amd", "csrc", "conv_frag.h")).read()
    body = src[src.index("void tile_walk("):]
    body = body[:body.index("\n}\n")]
    for piece in ("if (G & 7) { first = b; step = G; end = ntile; return; }", "const int x = b & 7;", "(((long)x * ntile) >> 3)",
                  "(((long)(x + 1) * ntile) >> 3)", "step = G >> 3;", "first = lo + (b >> 3);"):
        assert piece in body, piece


@pytest.mark.parametrize("G,ntile", [(512, 6400), (512, 12800), (512, 520), (256, 200), (512, 513), (8, 3), (8, 64), (64, 64), (250, 1000), (7, 50),
                                     (512, 1600), (768, 9216), (1024, 1601), (16, 1), (512, 0)])
def test_every_tile_once_and_contiguous_ranges(G, ntile):
    seen = {}
    for b in range(G):
        t, step, end = tile_walk(G, b, ntile)
        while t < end:
            assert t not in seen, (t, b, seen.get(t))
            seen[t] = b
            t += step
    assert sorted(seen) == list(range(ntile))
    if G % 8 == 0:
        for x in range(8):
            mine = sorted(t for t, b in seen.items() if b % 8 == x)
            assert mine == list(range(mine[0], mine[0] + len(mine))) if mine else True          # one contiguous range per XCD
            assert abs(len(mine) - ntile / 8) < 1                                                  # balanced to within one tile
        # the first round of an XCD's workgroups: consecutive tiles
        for x in range(8):
            first = [tile_walk(G, b, ntile) for b in range(x, G, 8)]
            starts = [t for t, _, end in first if t < end]
            assert starts == list(range(starts[0], starts[0] + len(starts))) if starts else True
