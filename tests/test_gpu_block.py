"""The pivot-block kernel on single dense blocks (k_factor_blk, hqp_amd/csrc/factor_blk.hip.h), through the C ABI's
debug entry: P A P' = L D L' with the pivot rule of hqp/spBKP.C:392, 431-438, 471, 480 restricted to the block,
and M = L^-1, for every block count and start offset the panels can meet."""
import numpy as np
import pytest

from tests import blockcheck as bc

pytestmark = pytest.mark.gpu

SIZES = [1, 2, 5, 15, 16, 17, 31, 32, 33, 47, 48, 64, 65, 80, 100, 127, 128, 129, 150, 160, 161, 176, 191, 192]


@pytest.mark.parametrize("kind", ["spd", "qd", "indef", "kkt0", "tiny"])
def test_block_factorisation_of_every_size(kind):
    worst = (0.0, 0.0)
    for p in SIZES:
        A = bc.make_block(kind, p, 100 + p)
        out = bc.factor_block(A, variant=0)
        err, inv, ok, growth = bc.check_block(A, out)
        c = out["counters"]
        assert ok, (kind, p, c)
        if c[2] == 0:  # nothing perturbed: the factors reproduce the block
            assert err < 1e-11 * max(1.0, growth) ** 2, (kind, p, err, growth, c)
        assert inv < 1e-11 * max(1.0, growth) ** 2, (kind, p, inv, growth)
        if kind in ("spd", "qd"):
            assert c[1] == 0 and c[3] == 0, (kind, p, c)  # no 2x2 pivot, no slow step
        if kind in ("kkt0", "indef") and p >= 33:
            assert c[3] > 0, (kind, p, c)  # slow steps
        worst = (max(worst[0], err if c[2] == 0 else 0.0), max(worst[1], inv))
    print(kind, "worst", worst)


@pytest.mark.parametrize("kind", ["qd", "indef", "kkt0"])
def test_both_pivot_block_kernels_agree(kind):
    """k_factor_diag (rounds 1-3) and k_factor_blk apply the same pivot rule: same pivot order, same pivot
    types, factors equal to rounding."""
    for p in (20, 64, 100, 128):
        A = bc.make_block(kind, p, 7 + p)
        new, old = bc.factor_block(A, variant=0), bc.factor_block(A, variant=1)
        assert (new["lperm"] == old["lperm"]).all(), (kind, p)
        assert (new["ptype"] == old["ptype"]).all(), (kind, p)
        L1, L2 = np.tril(new["L"], -1), np.tril(old["L"], -1)
        assert np.abs(L1 - L2).max() <= 1e-9 * max(1.0, np.abs(L2).max()), (kind, p)


def test_sixteen_wavefront_variant_on_small_blocks():
    for p in (16, 40, 128):
        A = bc.make_block("indef", p, 3 * p)
        out = bc.factor_block(A, variant=2)
        err, inv, ok, growth = bc.check_block(A, out)
        assert ok and err < 1e-11 * growth ** 2 and inv < 1e-11 * growth ** 2, (p, err, inv)


@pytest.mark.parametrize("kind", ["qd", "indef", "kkt0", "tiny"])
def test_the_instances_of_twelve_wavefronts_agree(kind):
    """Five / six blocks per wavefront (fronts of up to 160 / 176 pivots, what run_factor launches for them) against eight
    (up to 192): the same pivot sequence and the same factor to the last bit - the blocks are dealt to the wavefronts
    differently, the arithmetic of a block is the same."""
    for p in (129, 144, 160, 176):
        A = bc.make_block(kind, p, 7 * p)
        b = bc.factor_block(A, variant=2)
        for variant in (0, 3):  # (0: five blocks per wavefront up to 160 pivots, six beyond; 3: six)
            a = bc.factor_block(A, variant=variant)
            err, inv, ok, growth = bc.check_block(A, a)
            assert ok and err < 1e-11 * growth ** 2 and inv < 1e-11 * growth ** 2, (kind, p, variant, err, inv)
            assert (a["lperm"] == b["lperm"]).all() and (a["ptype"] == b["ptype"]).all(), (kind, p, variant)
            assert np.array_equal(a["L"], b["L"]) and np.array_equal(a["W"], b["W"]), (kind, p, variant)
