"""GPU (-m gpu): hqpkkt_franke with one read-back per iteration (the first residual of the solve comes back with the
scalars of the step; an unfinished solve puts the iterate back, refines and takes the step again) against the same
loop waiting for the residual before the step (HQPKKT_FRANKE_TWO_READS): the same arithmetic, so the same iterates."""
import numpy as np
import pytest

from hqp_amd import ipmatrix, problems

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", ["banded", "did400", "did2000", "lq"])
def test_franke_one_read_back_gives_the_same_iterates(case, monkeypatch):
    prog = {"banded": lambda: problems.banded_qp(300, 8, 5), "did400": lambda: problems.did_like_qp(400),
            "did2000": lambda: problems.did_like_qp(2000), "lq": lambda: problems.lq_docp(40, 6, 2, final_eq=2)}[case]()
    A = ipmatrix.IpRedSpBKP()
    A.init(prog)
    xa, ya, za, wa, ia = A.franke(prog, max_iters=300)
    monkeypatch.setenv("HQPKKT_FRANKE_TWO_READS", "1")
    B = ipmatrix.IpRedSpBKP()
    B.init(prog)
    xb, yb, zb, wb, ib = B.franke(prog, max_iters=300)
    assert (ia["result"], ia["iters"]) == (ib["result"], ib["iters"]), (ia, ib)
    assert ia["result"] == 0
    for u, v in ((xa, xb), (ya, yb), (za, zb), (wa, wb)):
        assert np.array_equal(u, v)


@pytest.mark.parametrize("case", ["banded", "did400", "lq"])
def test_loops_on_the_callers_vectors_give_the_same_iterates(case, monkeypatch):
    """The device-resident loops capture the factorisation and the solve ON their own device vectors (no copies into
    and out of the handle's staging buffers, one graph per set of pointers) - against the staged form
    (HQPKKT_NO_DIRECT_VECTORS): the same kernels on the same values, so the same iterates, Mehrotra and Franke."""
    prog = {"banded": lambda: problems.banded_qp(300, 8, 5), "did400": lambda: problems.did_like_qp(400),
            "lq": lambda: problems.lq_docp(40, 6, 2, final_eq=2)}[case]()
    A = ipmatrix.IpRedSpBKP()
    A.init(prog)
    ma, fa = A.mehrotra(prog), A.franke(prog, max_iters=300)
    ma2 = A.mehrotra(prog)  # (again on the same handle: the graphs of the first run are replayed)
    monkeypatch.setenv("HQPKKT_NO_DIRECT_VECTORS", "1")
    B = ipmatrix.IpRedSpBKP()
    B.init(prog)
    mb, fb = B.mehrotra(prog), B.franke(prog, max_iters=300)
    mb2 = B.mehrotra(prog)
    # (the same sequence of calls on both handles: a handle's first solve may re-place zero diagonals once the values
    # are known - zd_policy -1 - so its first run and its later ones follow slightly different pivot orders)
    for a, b in ((ma, mb), (fa, fb), (ma2, mb2)):
        assert (a[-1]["result"], a[-1]["iters"]) == (b[-1]["result"], b[-1]["iters"]) and a[-1]["result"] == 0
        for u, v in zip(a[:4], b[:4]):
            assert np.array_equal(u, v)


@pytest.mark.parametrize("case", ["banded", "did400", "did2000", "lq", "banded_full"])
def test_small_ip_kernels_bit_identical(case, monkeypatch):
    """hqpkkt_mehrotra on small QPs runs an iteration's vector work between its solves in ONE workgroup each
    (k_ip_pred_small, k_ip_step_small: ipdriver.hip.h) instead of ten launches - minima and maxima in any order, the
    one sum in the order of the separate launches: the same iterates bit for bit (HQPKKT_NO_IP_SMALL keeps the launches)."""
    prog = {"banded": lambda: problems.banded_qp(300, 8, 5), "did400": lambda: problems.did_like_qp(400),
            "did2000": lambda: problems.did_like_qp(2000), "lq": lambda: problems.lq_docp(40, 6, 2, final_eq=2),
            "banded_full": lambda: problems.banded_qp(2000, 20, 3)}[case]()
    cls = ipmatrix.IpSpBKP if case == "banded_full" else ipmatrix.IpRedSpBKP
    A = cls()
    A.init(prog)
    xa, ya, za, wa, ia = A.mehrotra(prog)
    monkeypatch.setenv("HQPKKT_NO_IP_SMALL", "1")
    B = cls()
    B.init(prog)
    xb, yb, zb, wb, ib = B.mehrotra(prog)
    assert (ia["result"], ia["iters"], ia["n_solve"]) == (ib["result"], ib["iters"], ib["n_solve"]), (ia, ib)
    assert ia["result"] == 0 and ia["iters"] >= 3
    for u, v in ((xa, xb), (ya, yb), (za, zb), (wa, wb)):
        assert np.array_equal(u, v)
    assert ia["gap"] == ib["gap"] and ia["alpha"] == ib["alpha"]


@pytest.mark.parametrize("case", ["banded", "did400", "did2000", "lq", "banded_full"])
def test_segment_graphs_give_the_same_iterates(case, monkeypatch):
    """Round 6: everything between two read-backs of hqpkkt_mehrotra is one captured graph (factorisation + solve +
    residual + posting kernel; statistics + solve; step + the next right-hand sides), a whole step of hqpkkt_franke one
    graph - built from the same calls, no value of the host baked in (the posting kernel counts on the device, the
    residual word is cleared by the post, mu / zeta come through memory).  Against the loops launch by launch
    (HQPKKT_NO_IP_SEGMENTS): the same iterates bit for bit, also when a handle's graphs are replayed by a second run."""
    prog = {"banded": lambda: problems.banded_qp(300, 8, 5), "did400": lambda: problems.did_like_qp(400),
            "did2000": lambda: problems.did_like_qp(2000), "lq": lambda: problems.lq_docp(40, 6, 2, final_eq=2),
            "banded_full": lambda: problems.banded_qp(2000, 20, 3)}[case]()
    cls = ipmatrix.IpSpBKP if case == "banded_full" else ipmatrix.IpRedSpBKP
    A = cls()
    A.init(prog)
    ma, fa = A.mehrotra(prog), A.franke(prog, max_iters=300)
    ma2, fa2 = A.mehrotra(prog), A.franke(prog, max_iters=300)  # (replays)
    monkeypatch.setenv("HQPKKT_NO_IP_SEGMENTS", "1")
    B = cls()
    B.init(prog)
    mb, fb = B.mehrotra(prog), B.franke(prog, max_iters=300)
    mb2, fb2 = B.mehrotra(prog), B.franke(prog, max_iters=300)
    # (the same sequence of calls on both handles: a handle's first solve may re-place zero diagonals once the values are
    # known - zd_policy -1 - so its first run and its later ones follow slightly different pivot orders)
    for name, a, b in (("mehrotra", ma, mb), ("mehrotra, replayed", ma2, mb2), ("franke", fa, fb), ("franke, replayed", fa2, fb2)):
        assert (a[4]["result"], a[4]["iters"], a[4]["n_solve"]) == (b[4]["result"], b[4]["iters"], b[4]["n_solve"]), (name, a[4], b[4])
        assert a[4]["result"] == 0, name
        for vec, u, v in zip("xyzw", a[:4], b[:4]):
            assert np.array_equal(u, v), (name, vec, float(np.abs(u - v).max()), a[4], b[4])


def test_calls_with_host_vectors_as_graphs_give_the_same_results(monkeypatch):
    """A caller's factor / solve with HOST vectors (the reference's solvers through the shim): vectors in and out of the
    pinned buffer by kernels, status words posted, the whole call one graph - against the copy-engine chain
    (HQPKKT_NO_HOST_GRAPHS + HQPKKT_NO_HOST_KERNEL_COPIES): the same numbers, call after call."""
    prog = problems.did_like_qp(400)
    outs = []
    for env in ({}, {"HQPKKT_NO_HOST_GRAPHS": "1"}, {"HQPKKT_NO_HOST_GRAPHS": "1", "HQPKKT_NO_HOST_KERNEL_COPIES": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        M = ipmatrix.IpRedSpBKP()
        M.init(prog)
        got = []
        for rep in range(3):
            z, w, r1, r2, r3, r4 = problems.ip_state(prog, seed=3 + rep)
            M.factor(prog, z, w)
            d = [np.zeros(k) for k in (prog.n, prog.me, prog.m, prog.m)]
            res = M.solve(prog, z, w, r1, r2, r3, r4, *d)
            got.append((res, [x.copy() for x in d]))
        outs.append(got)
    for other in outs[1:]:
        for (ra, da), (rb, db) in zip(outs[0], other):
            assert ra == rb and ra < 1e-10
            for u, v in zip(da, db):
                assert np.array_equal(u, v)


@pytest.mark.parametrize("case", ["banded", "did400", "lq"])
def test_fused_vector_launches_give_the_same_bits(case, monkeypatch):
    """Round 6: the reduced plugin's vector work around the sweeps (tz + right-hand side; dx, dy + dz, dw) and its assembly
    (weights + entry values; scales + scatter) are one launch each, the dependent half evaluating what it needs in place
    with the same expressions - against the separate launches (HQPKKT_NO_FUSED_VECTORS): the same bits, solve and loop."""
    prog = {"banded": lambda: problems.banded_qp(300, 8, 5), "did400": lambda: problems.did_like_qp(400),
            "lq": lambda: problems.lq_docp(40, 6, 2, final_eq=2)}[case]()
    outs = []
    for env in ({}, {"HQPKKT_NO_FUSED_VECTORS": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        M = ipmatrix.IpRedSpBKP()
        M.init(prog)
        z, w, r1, r2, r3, r4 = problems.ip_state(prog, seed=5)
        M.factor(prog, z, w)
        d = [np.zeros(k) for k in (prog.n, prog.me, prog.m, prog.m)]
        res = M.solve(prog, z, w, r1, r2, r3, r4, *d)
        outs.append((res, d, M.mehrotra(prog)))
    (ra, da, ma), (rb, db, mb) = outs
    assert ra == rb
    for u, v in zip(da, db):
        assert np.array_equal(u, v)
    assert (ma[4]["result"], ma[4]["iters"]) == (mb[4]["result"], mb[4]["iters"]) and ma[4]["result"] == 0
    for u, v in zip(ma[:4], mb[:4]):
        assert np.array_equal(u, v)
