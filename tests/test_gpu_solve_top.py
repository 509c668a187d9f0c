"""GPU (-m gpu): the top levels of the tree solve in one launch (k_solve_top, hqp_amd/csrc/solve_top.hip.h) against
the per-level kernels on the same factorisation data, against the CPU oracle, and repeated (the protocol words of
the launch reset themselves)."""
import numpy as np
import pytest

from hqp_amd import ipmatrix, problems
from oracle import oracleapi

pytestmark = pytest.mark.gpu


def _solve(cls, prog, st, **kw):
    M = cls(**kw)
    M.init(prog)
    d = [np.zeros(k) for k in (prog.n, prog.me, prog.m, prog.m)]
    M.factor(prog, st[0], st[1])
    res = M.solve(prog, *st, *d)
    return M, d, res


@pytest.mark.parametrize("n,band,cls", [(3000, 20, ipmatrix.IpRedSpBKP), (6000, 40, ipmatrix.IpSpBKP),
                                        (12000, 80, ipmatrix.IpRedSpBKP), (9000, 85, ipmatrix.IpSpBKP)])
def test_fused_top_against_the_per_level_sweeps(n, band, cls, monkeypatch):
    """Same system, same factorisation kernels; the solve with the fused top and with HQPKKT_NO_SOLVE_TOP: both below
    the residual tolerance, solutions equal to rounding (the order of summation differs), and the fused launch is
    really in use (introspection 31)."""
    prog = problems.banded_qp(n, band, seed=7)
    st = problems.ip_state(prog, seed=3)
    A, da, ra = _solve(cls, prog, st)
    top = A.debug(31)
    assert top[0] >= 3 and top[1] < A.stats()["n_levels"], top
    monkeypatch.setenv("HQPKKT_NO_SOLVE_TOP", "1")
    B, db, rb = _solve(cls, prog, st)
    assert B.debug(31)[0] == 0
    assert ra <= 1e-10 and rb <= 1e-10, (ra, rb)
    for x, yv in zip(da, db):
        if len(x):
            assert np.abs(x - yv).max() <= 1e-9 * max(1.0, np.abs(yv).max())
    # repeated solves on one handle: the protocol words are back to zero after every launch
    for _ in range(5):
        d2 = [np.zeros(k) for k in (prog.n, prog.me, prog.m, prog.m)]
        r2 = A.solve(prog, *st, *d2)
        assert r2 == ra
        for x, yv in zip(d2, da):
            assert np.array_equal(x, yv)  # reproducible from run to run


@pytest.mark.parametrize("n,band,cls", [(6000, 40, ipmatrix.IpSpBKP), (20000, 60, ipmatrix.IpRedSpBKP)])
def test_split_sweeps_give_the_bits_of_the_fused_launch(n, band, cls, monkeypatch):
    """The two sweeps as launches of their own (forward leaves first, backward root first: the form in use, for any
    number of fronts) against both sweeps in one launch (HQPKKT_SOLVE_TOP_FUSED, at most 128 fronts, all resident):
    same arithmetic, same bits; repeated solves too."""
    prog = problems.banded_qp(n, band, seed=8)
    st = problems.ip_state(prog, seed=2)
    B, db, rb = _solve(cls, prog, st)
    monkeypatch.setenv("HQPKKT_SOLVE_TOP_FUSED", "1")
    A, da, ra = _solve(cls, prog, st)
    assert B.debug(31)[6] == 1 and B.debug(31)[0] == A.debug(31)[0] >= 3
    if A.debug(31)[6] == 0:  # (the larger system is split anyway)
        assert ra == rb
        for x, yv in zip(da, db):
            assert np.array_equal(x, yv)
    for _ in range(3):
        d2 = [np.zeros(k) for k in (prog.n, prog.me, prog.m, prog.m)]
        assert B.solve(prog, *st, *d2) == rb
        for x, yv in zip(d2, db):
            assert np.array_equal(x, yv)
    assert rb <= 1e-10


def test_fused_top_against_the_oracle():
    prog = problems.banded_qp(1600, 32, seed=11)
    st = problems.ip_state(prog, seed=5)
    O = oracleapi.OracleIpMatrix("SpBKP")
    O.init(prog)
    O.factor(st[0], st[1])
    osol, ores = O.solve(*st)
    M, d, res = _solve(ipmatrix.IpSpBKP, prog, st)
    assert M.debug(31)[0] >= 3
    assert res <= 1e-10
    assert O.residuum(*st, *d) <= ores + 1e-10 * max(1.0, max(np.abs(v).max() for v in d if len(v)))


@pytest.mark.parametrize("K", [60, 400, 2000])
def test_whole_tree_sweeps_on_trees_of_small_fronts(K, monkeypatch):
    """Double-integrator DOCP (fronts of a few pivots, a tree of a dozen levels): each sweep of the solve is one
    launch over all levels (k_solve_fwd_small<true> / k_solve_bwd_small<true>, contributions and solution travel as
    polled words).  Same arithmetic as the per-level launches: the solve and a whole device-resident Mehrotra run give
    the same bits; repeated solves too (the exchange arrays are back in their idle state after every solve)."""
    prog = problems.did_like_qp(K)
    st = problems.ip_state(prog, seed=3)
    A, da, ra = _solve(ipmatrix.IpRedSpBKP, prog, st)
    assert A.debug(31)[0] == 0 and A.debug(31)[4] == 1 and A.debug(31)[5] == 1
    for _ in range(4):
        d2 = [np.zeros(k) for k in (prog.n, prog.me, prog.m, prog.m)]
        assert A.solve(prog, *st, *d2) == ra
        for x, yv in zip(d2, da):
            assert np.array_equal(x, yv)
    xa = A.mehrotra(prog)
    monkeypatch.setenv("HQPKKT_NO_TREE_SWEEPS", "1")
    B, db, rb = _solve(ipmatrix.IpRedSpBKP, prog, st)
    assert B.debug(31)[4] == 0
    assert ra == rb and ra <= 1e-10
    for x, yv in zip(da, db):
        assert np.array_equal(x, yv)
    xb = B.mehrotra(prog)
    assert xa[-1]["iters"] == xb[-1]["iters"] and xa[-1]["result"] == 0
    assert np.array_equal(xa[0], xb[0])


@pytest.mark.parametrize("make,cls,what", [(lambda: problems.banded_qp(6000, 40, seed=7), ipmatrix.IpSpBKP, "top"),
                                           (lambda: problems.did_like_qp(400), ipmatrix.IpRedSpBKP, "tree")])
def test_a_poll_that_gives_up_falls_back_to_the_per_level_launches(make, cls, what, monkeypatch):
    """HQPKKT_POLL_LIMIT=0 (test hook): every poll of a launch that spans tree levels gives up as soon as it has to
    wait.  The call must not return a wrong result: the handle switches to the per-level launches, runs the operation
    again and counts the event (hqpkkt_stats.n_poll_fallbacks); the results are those of the per-level launches, and
    later calls on the handle stay there.  (the polled launches rest on index-ordered dispatch.)"""
    prog = make()
    st = problems.ip_state(prog, seed=3)
    monkeypatch.setenv("HQPKKT_NO_SOLVE_TOP", "1")
    monkeypatch.setenv("HQPKKT_NO_TREE_SWEEPS", "1")
    R, dr, rr = _solve(cls, prog, st)  # the per-level launches, chosen up front
    monkeypatch.delenv("HQPKKT_NO_SOLVE_TOP")
    monkeypatch.delenv("HQPKKT_NO_TREE_SWEEPS")
    monkeypatch.setenv("HQPKKT_POLL_LIMIT", "0")
    A, da, ra = _solve(cls, prog, st)
    monkeypatch.delenv("HQPKKT_POLL_LIMIT")
    assert A.stats()["n_poll_fallbacks"] >= 1
    assert A.debug(31)[0] == 0 and A.debug(31)[4] == 0  # neither k_solve_top nor the whole-tree sweeps any more
    assert ra == rr and ra <= 1e-10
    for x, yv in zip(da, dr):
        assert np.array_equal(x, yv)
    d2 = [np.zeros(k) for k in (prog.n, prog.me, prog.m, prog.m)]
    A.factor(prog, st[0], st[1])
    assert A.solve(prog, *st, *d2) == rr
    if what == "tree":  # the device-resident loop on the switched handle, and on one that has to switch inside the loop
        xa = A.mehrotra(prog)
        monkeypatch.setenv("HQPKKT_POLL_LIMIT", "0")
        B = cls()
        B.init(prog)
        xb = B.mehrotra(prog)
        monkeypatch.delenv("HQPKKT_POLL_LIMIT")
        assert B.stats()["n_poll_fallbacks"] >= 1
        assert xa[-1]["iters"] == xb[-1]["iters"] and xb[-1]["result"] == 0
        assert np.array_equal(xa[0], xb[0])
    # a handle made afterwards polls with the normal limit again
    C, dc, rc = _solve(cls, prog, st)
    assert C.stats()["n_poll_fallbacks"] == 0 and (C.debug(31)[0] >= 3 or C.debug(31)[4] == 1)
