"""The work lists of the cut form of the STAGED engine's fp64 product (hqp_amd/csrc/sk_table.hpp, host code): whatever
shares the plan gives the workgroups, every k-slab of every tile is computed exactly once, a tile's pieces park in
slots of their own in the order of their k ranges, and nobody's list is longer than the stride says."""
import numpy as np
import pytest

from hqp_amd import ipmatrix

CASES = [(1600, 313, 512), (820, 313, 512), (780, 313, 512), (321, 313, 512), (511, 40, 512), (513, 64, 512), (1000, 200, 512),
         (2000, 313, 512), (700, 100, 512), (330, 313, 512), (5000, 70, 512), (600, 33, 512), (257, 1000, 512), (150, 128, 208),
         (104, 313, 208), (1600, 313, 256)]


@pytest.mark.parametrize("tiles,nslab,grid", CASES)
def test_every_slab_of_every_tile_once(tiles, nslab, grid):
    got = ipmatrix.sk_table(tiles, nslab, grid)
    assert got is not None
    units, pieces, wa, wb = got
    assert units.shape[0] == grid and wa >= wb >= 0
    cover = [[] for _ in range(tiles)]
    slots = {}
    for b in range(grid):
        ended = False
        for (t, s0, s1, slot0, np_, j) in units[b]:
            if t < 0:
                ended = True
                continue
            assert not ended, "a unit behind the end mark"
            assert 0 <= t < tiles and 0 <= s0 < s1 <= nslab and 0 <= j < np_
            cover[t].append((s0, s1, j))
            if np_ > 1:
                assert 0 <= slot0 and slot0 + np_ <= pieces
                assert slots.setdefault(t, (slot0, np_)) == (slot0, np_)
            else:
                assert (s0, s1) == (0, nslab)
        assert ended, "no end mark"
    used = np.zeros(pieces, dtype=int)
    for t in range(tiles):
        c = sorted(cover[t])
        at = 0
        for q, (s0, s1, j) in enumerate(c):
            assert s0 == at and j == q, (t, c)
            at = s1
        assert at == nslab, (t, c)
        if t in slots:
            assert slots[t][1] == len(c)
            used[slots[t][0]:slots[t][0] + slots[t][1]] += 1
        else:
            assert len(c) == 1
    assert (used == 1).all()
    # whole tiles first: the first workgroup of a CU (blockIdx < grid / 2) gets at least as many as the second
    work = np.array([[(u[2] - u[1]) for u in units[b] if u[0] >= 0] for b in range(grid)], dtype=object)
    tot = np.array([sum(w) for w in work])
    assert tot[:grid // 2].min() >= wa * nslab and tot[grid // 2:].min() >= wb * nslab
    assert tot.sum() == tiles * nslab


def test_headline_shapes_take_whole_tiles():
    # W of the headline (40 x 40 tiles): four whole tiles for the first workgroup of a CU, two and a quarter for the second;
    # G (820 lower tiles): two, and one and a quarter for 208 of the second ones
    u, pieces, wa, wb = ipmatrix.sk_table(1600, 313, 512)
    assert (wa, wb, pieces) == (4, 2, 256)
    u, pieces, wa, wb = ipmatrix.sk_table(820, 313, 512)
    assert (wa, wb, pieces) == (2, 1, 208)
