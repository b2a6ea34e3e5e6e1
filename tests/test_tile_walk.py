"""Host-side check of the tile orders the persistent kernels use (csrc/common.h xcd_walk, csrc/gemm.hip's XCD-major workgroup
number, csrc/conv.hip k_wgrad4_os's run / row split): every tile exactly once, whatever the grid.  The formulas are restated here
(pure integer arithmetic) - the GPU parity tests check the kernels' results, this checks the invariant they rely on."""
import re
from pathlib import Path

import pytest

CSRC = Path(__file__).resolve().parents[1] / "dcase2019_task4_amd" / "csrc"


def xcd_walk(w, grid, n_tiles):
    """common.h xcd_walk: (first, end, step) of workgroup w."""
    if grid % 8 == 0:
        chunk = (n_tiles + 7) >> 3
        lo = (w & 7) * chunk
        hi = min(lo + chunk, n_tiles)
        return lo + (w >> 3), hi, grid >> 3
    return w, n_tiles, grid


@pytest.mark.parametrize("grid", [1, 3, 8, 24, 60, 64, 117, 240, 256, 512])
@pytest.mark.parametrize("n_tiles", [1, 7, 8, 9, 40, 240, 471, 960, 961, 2800])
def test_xcd_walk_visits_every_tile_once(grid, n_tiles):
    grid = min(grid, n_tiles)                       # the launchers cap the grid at the tile count
    seen = []
    for w in range(grid):
        first, end, step = xcd_walk(w, grid, n_tiles)
        seen.extend(range(first, end, step))
    assert sorted(seen) == list(range(n_tiles))


def test_xcd_walk_keeps_an_xcd_on_consecutive_tiles_and_the_xcds_level():
    grid, n_tiles = 256, 960                        # block-1 convolutions at B = 24: 3.75 tiles per workgroup
    per_xcd = {}
    for w in range(grid):
        first, end, step = xcd_walk(w, grid, n_tiles)
        per_xcd.setdefault(w & 7, []).extend(range(first, end, step))
    for x, tiles in per_xcd.items():
        tiles.sort()
        assert tiles == list(range(tiles[0], tiles[0] + len(tiles)))       # one contiguous run of the image
        assert len(tiles) == n_tiles // 8                                  # same load on every XCD


def test_the_restated_formula_is_the_one_in_common_h():
    src = (CSRC / "common.h").read_text()
    body = src[src.index("__device__ __forceinline__ TileWalk xcd_walk"):]
    body = body[:body.index("\n}\n")]
    assert re.search(r"chunk = \(n_tiles \+ 7\) >> 3, lo = \(w & 7\) \* chunk", body)
    assert re.search(r"return \{lo \+ \(w >> 3\), hi, g >> 3\}", body)
    assert re.search(r"return \{w, n_tiles, g\}", body)


@pytest.mark.parametrize("gx,gy,gz", [(2, 12, 64), (8, 12, 64), (1, 3, 16), (3, 3, 8), (4, 12, 32)])
def test_gemm_xcd_major_workgroup_number_is_a_permutation(gx, gy, gz):
    total = gx * gy * gz
    seen = set()
    for lin in range(total):
        l2 = (lin & 7) * (total >> 3) + (lin >> 3) if total % 8 == 0 else lin
        seen.add((l2 % gx, (l2 // gx) % gy, l2 // (gx * gy)))
    assert len(seen) == total


@pytest.mark.parametrize("n_tiles", [2, 20, 64, 100, 240, 700])
def test_wgrad4_os_runs_and_rows_cover_every_tile_four_times(n_tiles):
    n_runs = min(64, (n_tiles + 7) & ~7)
    count = {}
    for w in range(4 * n_runs):
        run, row = (w & 7) + 8 * (w >> 5), (w >> 3) & 3
        assert run < n_runs
        for t in range(run, n_tiles, n_runs):
            count.setdefault(t, set()).add(row)
    assert sorted(count) == list(range(n_tiles))
    assert all(rows == {0, 1, 2, 3} for rows in count.values())
    # the four rows of a run sit on ONE XCD (workgroup number % 8)
    for run in range(n_runs):
        ws = [w for w in range(4 * n_runs) if (w & 7) + 8 * (w >> 5) == run]
        assert len(ws) == 4 and len({w % 8 for w in ws}) == 1
