"""GPU (-m gpu): irregular sparsity (BASELINE.json configs[4]).
ordering=1 (nested dissection of the graph itself, hqp_amd/csrc/analysis.cpp) against the CPU oracle and
against the default tree on mesh-structured and random sparse QPs, and the reference's own Hqp_SqpPowell
(oracle/_ref, compiled unmodified) driving our plugin over a sparse NLP (Prg_GridNLP, oracle/ref_sqpdrive.cc)
next to the reference's RedSpBKP - what hqp_cute/hqp_cute.tcl:22-46 runs for the CUTE collection."""
import numpy as np
import pytest

from hqp_amd import ipmatrix, problems
from oracle import oracleapi, refapi

pytestmark = pytest.mark.gpu

CLS = {"SpBKP": ipmatrix.IpSpBKP, "RedSpBKP": ipmatrix.IpRedSpBKP}
CASES = {
    "grid16": lambda: problems.grid_sparse_qp(16, 16),
    "grid24x10": lambda: problems.grid_sparse_qp(24, 10, seed=3, eq_every=2),
    "grid20_far": lambda: problems.grid_sparse_qp(20, 20, seed=5, long_range=40),
    "random300": lambda: problems.random_sparse_qp(300, 90, 200, row_nnz=3, seed=9),
    "banded": lambda: problems.banded_qp(400, 10, 7),
}


@pytest.mark.parametrize("kind", ["SpBKP", "RedSpBKP"])
@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("spread", [0.0, 2.0])
@pytest.mark.parametrize("ordering", [1, 2])
def test_graph_dissection_against_the_oracle(kind, case, spread, ordering):
    prog = CASES[case]()
    st = problems.ip_state(prog, 3, spread)
    O = oracleapi.OracleIpMatrix(kind)
    O.init(prog)
    O.factor(st[0], st[1])
    osol, ores = O.solve(*st)
    M = CLS[kind](ordering=ordering)
    M.init(prog)
    if ordering == 1:  # reported as before; 2 numbers the graph by a plain reverse Cuthill-McKee pass
        assert M.mat_sbw == O.sbw and np.array_equal(M.perm(), O.perm())
    else:
        assert sorted(M.perm()) == list(range(prog.n + prog.me + (prog.m if kind == "SpBKP" else 0)))
    d = [np.zeros(k) for k in (prog.n, prog.me, prog.m, prog.m)]
    M.factor(prog, st[0], st[1])
    res = M.solve(prog, *st, *d)
    scale = max(1.0, max(np.abs(v).max() for v in osol))
    assert res <= ores + 1e-10 * scale
    assert O.residuum(*st, *d) <= ores + 1e-10 * scale
    for a, b in zip(d, osol):
        assert np.abs(a - b).max() <= 1e-8 * scale


def test_graph_dissection_shrinks_the_factor_of_a_mesh():
    """150 x 150 cells (KKT dimension 3.0e4 reduced): both trees solve the system, the graph's own
    dissection with a fifth of the entries in L and a twentieth of the operations"""
    prog = problems.grid_sparse_qp(150, 150)
    st = problems.ip_state(prog, 1, 1.0)
    sols, stats = [], []
    for o in (0, 1):
        M = ipmatrix.IpRedSpBKP(ordering=o)
        M.init(prog)
        d = [np.zeros(k) for k in (prog.n, prog.me, prog.m, prog.m)]
        M.factor(prog, st[0], st[1])
        assert M.solve(prog, *st, *d) < 1e-10
        sols.append(d), stats.append(M.stats())
    assert stats[1]["nnz_factor"] * 4 < stats[0]["nnz_factor"]
    assert stats[1]["flops_factor"] * 15 < stats[0]["flops_factor"]
    scale = max(1.0, max(np.abs(v).max() for v in sols[0]))
    for a, b in zip(*sols):
        assert np.abs(a - b).max() <= 1e-8 * scale


@pytest.mark.parametrize("g", [12, 30])
@pytest.mark.parametrize("pair", [("RedSpBKP", "RedSpBKPHip"), ("SpBKP", "SpBKPHip")])
@pytest.mark.parametrize("ordering", [0, 1])
def test_reference_sqp_solver_over_a_sparse_nlp(g, pair, ordering):
    if not refapi.host_available("hip"):
        pytest.skip("oracle/_ref/libhqphost_hip.so not present")
    ref = refapi.sqp_grid(g, g, "Mehrotra", pair[0], host="hip")
    got = refapi.sqp_grid(g, g, "Mehrotra", pair[1], host="hip", ordering=ordering)
    assert ref["rc"] == 0 and got["rc"] == 0
    assert got["sqp_iters"] == ref["sqp_iters"]
    assert abs(got["qp_iters"] - ref["qp_iters"]) <= 2
    assert abs(got["f"] - ref["f"]) <= 1e-6 * abs(ref["f"])
    assert got["norm_inf"] < 1e-6 and got["norm_grd_L"] < 1e-5


@pytest.mark.parametrize("kind", ["RedSpBKP", "SpBKP"])
def test_band_with_long_range_couplings_against_the_oracle(kind):
    """Irregular sparsity that is no mesh: a band of 21 entries per row of Q plus 1 % random far couplings
    (problems.banded_long_range_qp), through the dissection of the graph itself (ordering 1 keeps the reference's RCM
    numbers, 2 does without that pass): residual within 1e-10 of the CPU oracle's, same solution."""
    prog = problems.banded_long_range_qp(2000, 10, 20, seed=3, min_dist=300)
    st = problems.ip_state(prog, 4, 1.0)
    O = oracleapi.OracleIpMatrix(kind)
    O.init(prog)
    O.factor(st[0], st[1])
    osol, ores = O.solve(*st)
    cls = {"SpBKP": ipmatrix.IpSpBKP, "RedSpBKP": ipmatrix.IpRedSpBKP}[kind]
    for ordering in (1, 2):
        M = cls(ordering=ordering)
        M.init(prog)
        if ordering == 1:
            assert M.mat_sbw == O.sbw
        M.factor(prog, st[0], st[1])
        d = [np.zeros(k) for k in (prog.n, prog.me, prog.m, prog.m)]
        res = M.solve(prog, *st, *d)
        assert res <= ores + 1e-10, (ordering, res, ores)
        scale = max(1.0, max(np.abs(v).max() for v in osol))
        assert max(np.abs(a - b).max() for a, b in zip(d, osol)) <= 1e-8 * scale


def test_band_with_long_range_couplings_at_1e5_variables():
    """The same structure at 10^5 variables (KKT dimension 1.5e5 reduced), 21 entries per row of Q and 1000 far
    couplings: a band ordering sees a semi-bandwidth of ~1.9e4 (31 Tflop, 60 GB of fronts); the graph's own dissection
    (ordering 2) with separators from level structures alone had fronts of 1.17e4 rows and 1.19 Tflop (round 4); with
    the cuts by linear order among the candidates (analysis.cpp, round 5: one end of every crossing far coupling plus
    the band) fronts of 3.4e3 rows and 16 Gflop.  Size-independent properties: residual <= 1e-10, residuum() of the
    solution equal to what solve() returned, linear in the right-hand side, a second factorisation bit-identical."""
    prog = problems.banded_long_range_qp(100000, 10, 1000)
    st = problems.ip_state(prog, 2, 1.0)
    M = ipmatrix.IpRedSpBKP(ordering=2)
    M.init(prog)
    s = M.stats()
    assert s["max_front"] <= 4000 and s["flops_factor"] < 5e10 and s["bytes_panels"] + s["bytes_updates"] < 2e9, s
    new = lambda: [np.zeros(k) for k in (prog.n, prog.me, prog.m, prog.m)]
    M.factor(prog, st[0], st[1])
    d1 = new()
    res = M.solve(prog, *st, *d1)
    assert res <= 1e-10
    assert abs(M.residuum(prog, *st, *d1) - res) <= 1e-13
    st2 = (st[0], st[1]) + tuple(2.0 * v for v in st[2:])
    d2 = new()
    assert M.solve(prog, *st2, *d2) <= 1e-10
    scale = max(np.abs(v).max() for v in d1)
    assert max(np.abs(b - 2.0 * a).max() for a, b in zip(d1, d2)) <= 1e-9 * scale
    M.factor(prog, st[0], st[1])
    d3 = new()
    M.solve(prog, *st, *d3)
    assert all(np.array_equal(a, b) for a, b in zip(d1, d3))


def test_cute_style_rows_at_1e5_variables():
    """SURVEY.md 8(d) C5's row density at 10^5 variables (problems.cute_like_qp: 10 ... 100 entries per row of Q in a window
    of 400 columns, 0.1 % of them anywhere; 30 000 equality rows of 5 ... 50 entries; KKT dimension 1.3e5 reduced, 6.6e6
    entries) through the graph's own dissection (ordering 2) - the same generator the reference fixture of
    tests/golden_full_size/ uses at n = 6500.  Size-independent properties: residual <= 1e-10, residuum() of the solution
    equal to what solve() returned, linear in the right-hand side, a second factorisation bit-identical."""
    prog = problems.cute_like_qp(100000)
    st = problems.ip_state(prog, 2, 1.0)
    M = ipmatrix.IpRedSpBKP(ordering=2)
    M.init(prog)
    s = M.stats()
    assert s["bytes_panels"] + s["bytes_updates"] < 60e9, s
    new = lambda: [np.zeros(k) for k in (prog.n, prog.me, prog.m, prog.m)]
    M.factor(prog, st[0], st[1])
    d1 = new()
    res = M.solve(prog, *st, *d1)
    assert res <= 1e-10
    assert abs(M.residuum(prog, *st, *d1) - res) <= 1e-13
    st2 = (st[0], st[1]) + tuple(2.0 * v for v in st[2:])
    d2 = new()
    assert M.solve(prog, *st2, *d2) <= 1e-10
    scale = max(np.abs(v).max() for v in d1)
    assert max(np.abs(b - 2.0 * a).max() for a, b in zip(d1, d2)) <= 1e-9 * scale
    M.factor(prog, st[0], st[1])
    d3 = new()
    M.solve(prog, *st, *d3)
    assert all(np.array_equal(a, b) for a, b in zip(d1, d3))
    print("cute_like_qp(1e5):", {k: s[k] for k in ("dim", "max_front", "n_levels", "n_supernodes", "flops_factor", "bytes_panels", "bytes_updates")})


def test_sqp_loop_over_a_sparse_nlp_of_1e5_variables():
    """BASELINE configs[4]'s stand-in at 10^5 variables INSIDE the test suite: the reference's unmodified Hqp_SqpPowell
    over Prg_GridNLP on 320 x 320 cells (102 400 variables) with the reference's Hqp_IpsMehrotra driving RedSpBKPHip
    (dissection of the KKT graph, mat_ordering 2) and with the device-resident MehrotraHip: both converge (rc 0: optimal),
    to the same objective, in the same number of SQP iterations, with the KKT conditions of the NLP met."""
    if not refapi.host_available("hip"):
        pytest.skip("oracle/_ref/libhqphost_hip.so not present")
    a = refapi.sqp_grid(320, 320, "Mehrotra", "RedSpBKPHip", host="hip", ordering=2)
    b = refapi.sqp_grid(320, 320, "MehrotraHip", "RedSpBKPHip", host="hip", ordering=2)
    assert a["n"] == b["n"] == 102400
    assert a["rc"] == 0 and b["rc"] == 0, (a, b)
    assert a["sqp_iters"] == b["sqp_iters"] and abs(a["qp_iters"] - b["qp_iters"]) <= 3, (a, b)
    assert abs(a["f"] - b["f"]) <= 1e-6 * abs(a["f"])
    for r in (a, b):
        assert r["norm_inf"] < 1e-6 and r["norm_grd_L"] < 1e-5, r


@pytest.mark.parametrize("pair", [("RedSpBKP", "RedSpBKPHip"), ("SpBKP", "SpBKPHip")])
def test_reference_sqp_solver_over_a_sparse_nlp_with_far_couplings(pair):
    """Prg_GridNLP with couplings between distant cells (the irregular part: 1 % of the cells) through the reference's
    Hqp_SqpPowell: its own plugin against ours with the graph's dissection - same SQP iterations, same objective."""
    if not refapi.host_available("hip"):
        pytest.skip("oracle/_ref/libhqphost_hip.so not present")
    ref = refapi.sqp_grid(40, 40, "Mehrotra", pair[0], host="hip", far=16)
    got = refapi.sqp_grid(40, 40, "Mehrotra", pair[1], host="hip", ordering=2, far=16)
    assert ref["rc"] == 0 and got["rc"] == 0, (ref, got)
    assert got["sqp_iters"] == ref["sqp_iters"] and abs(got["qp_iters"] - ref["qp_iters"]) <= 2, (ref, got)
    assert abs(got["f"] - ref["f"]) <= 1e-6 * abs(ref["f"])
    assert got["norm_inf"] < 1e-6 and got["norm_grd_L"] < 1e-5


def test_sqp_loop_over_an_irregular_nlp_of_1e5_variables():
    """the SQP loop at 10^5 variables on the irregular generator - 316 x 316 cells (five entries per
    row) with 1000 couplings between distant cells, the reference's Hqp_SqpPowell + Hqp_IpsMehrotra driving RedSpBKPHip
    (mat_ordering 2), and the device-resident MehrotraHip: both optimal, same objective, same SQP iterations.  (The
    reference's own RedSpBKP is no partner here: the far couplings make its band the whole matrix.)"""
    if not refapi.host_available("hip"):
        pytest.skip("oracle/_ref/libhqphost_hip.so not present")
    a = refapi.sqp_grid(316, 316, "Mehrotra", "RedSpBKPHip", host="hip", ordering=2, far=1000)
    b = refapi.sqp_grid(316, 316, "MehrotraHip", "RedSpBKPHip", host="hip", ordering=2, far=1000)
    assert a["n"] == b["n"] == 99856
    assert a["rc"] == 0 and b["rc"] == 0, (a, b)
    assert a["sqp_iters"] == b["sqp_iters"] and abs(a["qp_iters"] - b["qp_iters"]) <= 3, (a, b)
    assert abs(a["f"] - b["f"]) <= 1e-6 * abs(a["f"])
    for r in (a, b):
        assert r["norm_inf"] < 1e-6 and r["norm_grd_L"] < 1e-5, r


def test_random_sparse_system_with_far_couplings_against_the_oracle():
    """a parity test on an irregular system of n >= 2000 - 4000 variables, 11 entries per row of Q,
    band-wide equality rows, 1 % couplings between variables at least 500 apart - both plugins, both graph orderings,
    w/z spread over four decades: residual within 1e-10 of the CPU oracle's, the oracle's residual of our solution too."""
    prog = problems.banded_long_range_qp(4000, 5, 40, seed=11, min_dist=500)
    st = problems.ip_state(prog, 6, 2.0)
    for kind in ("RedSpBKP", "SpBKP"):
        O = oracleapi.OracleIpMatrix(kind)
        O.init(prog)
        O.factor(st[0], st[1])
        osol, ores = O.solve(*st)
        scale = max(1.0, max(np.abs(v).max() for v in osol))
        for ordering in (1, 2):
            M = CLS[kind](ordering=ordering)
            M.init(prog)
            assert M.stats()["max_front"] <= 700, M.stats()
            M.factor(prog, st[0], st[1])
            d = [np.zeros(k) for k in (prog.n, prog.me, prog.m, prog.m)]
            res = M.solve(prog, *st, *d)
            assert res <= ores + 1e-10 * scale, (kind, ordering, res, ores)
            assert O.residuum(*st, *d) <= ores + 1e-10 * scale
            assert max(np.abs(a - b).max() for a, b in zip(d, osol)) <= 1e-8 * scale


def test_update_blocks_on_128_tiles_against_the_oracle(monkeypatch):
    """k_schur_update_big (fronts with >= 2048 border rows: the top of an irregular graph's tree) on EVERY front
    (HQPKKT_SCHUR_BIG_B=1, the test hook of Analysis::run): banded, double-integrator and irregular systems, borders
    from a handful to several hundred rows - residuals within 1e-10 of the CPU oracle's, solutions to 1e-8."""
    monkeypatch.setenv("HQPKKT_SCHUR_BIG_B", "1")
    cases = [("RedSpBKP", problems.banded_long_range_qp(4000, 5, 40, seed=11, min_dist=500), dict(ordering=2)),
             ("SpBKP", problems.banded_qp(3000, 30, 7), {}),
             ("RedSpBKP", problems.banded_qp(2000, 60, 3), dict(max_pivots=48)),
             ("SpBKP", problems.did_like_qp(300), {}),
             ("RedSpBKP", problems.grid_sparse_qp(60, 50, seed=2, long_range=40), dict(ordering=1))]
    for kind, prog, kw in cases:
        st = problems.ip_state(prog, 3, 1.0)
        O = oracleapi.OracleIpMatrix(kind)
        O.init(prog)
        O.factor(st[0], st[1])
        osol, ores = O.solve(*st)
        scale = max(1.0, max(np.abs(v).max() for v in osol))
        M = CLS[kind](**kw)
        M.init(prog)
        M.factor(prog, st[0], st[1])
        d = [np.zeros(k) for k in (prog.n, prog.me, prog.m, prog.m)]
        res = M.solve(prog, *st, *d)
        assert res <= ores + 1e-10 * scale, (kind, kw, res, ores)
        assert O.residuum(*st, *d) <= ores + 1e-10 * scale
        assert max(np.abs(a - b).max() for a, b in zip(d, osol)) <= 1e-8 * scale


def test_full_size_irregular_sqp_loop_and_properties():
    """BASELINE configs[4] at FULL size on the irregular generator: 10^6 variables (1000 x 1000 cells, five entries per
    row) with 1 % = 10 000 couplings between distant cells.  (i) The full SQP loop: the reference's unmodified
    Hqp_SqpPowell over Prg_GridNLP with the device-resident MehrotraHip driving RedSpBKPHip (mat_ordering 2) ends optimal
    with the KKT conditions of the NLP met (round 4's separators would have needed fronts of 1.4e5 rows and 660 GB here:
    tools/c5_irregular.py, profiles/r05_c5_irregular.jsonl).  (ii) The KKT system of the QP of the same structure through
    size-independent properties: residual <= 1e-10, residuum() of the solution equal to what solve() returned, linear
    in the right-hand side."""
    if refapi.host_available("hip"):
        r = refapi.sqp_grid(1000, 1000, "MehrotraHip", "RedSpBKPHip", host="hip", ordering=2, far=10000)
        assert r["n"] == 1000000 and r["rc"] == 0, r
        assert r["norm_inf"] < 1e-6 and r["norm_grd_L"] < 1e-5, r
    prog = problems.grid_sparse_qp(1000, 1000, seed=5, long_range=10000)
    st = problems.ip_state(prog, 2, 1.0)
    M = ipmatrix.IpRedSpBKP(ordering=2)
    M.init(prog)
    s = M.stats()
    assert s["max_front"] <= 20000 and s["bytes_panels"] + s["bytes_updates"] < 30e9, s
    new = lambda: [np.zeros(k) for k in (prog.n, prog.me, prog.m, prog.m)]
    M.factor(prog, st[0], st[1])
    d1 = new()
    res = M.solve(prog, *st, *d1)
    assert res <= 1e-10
    assert abs(M.residuum(prog, *st, *d1) - res) <= 1e-13
    st2 = (st[0], st[1]) + tuple(2.0 * v for v in st[2:])
    d2 = new()
    assert M.solve(prog, *st2, *d2) <= 1e-10
    scale = max(np.abs(v).max() for v in d1)
    assert max(np.abs(b - 2.0 * a).max() for a, b in zip(d1, d2)) <= 1e-9 * scale


def test_full_size_mesh_properties():
    """The stand-in for configs[4] at 10^6 variables (1000 x 1000 cells, reduced KKT dimension 1.33e6, ordering 2)
    through size-independent properties: residual <= 1e-10, residuum() of the solution equal to what solve()
    returned, linear in the right-hand side, a second factorisation bit-identical."""
    prog = problems.grid_sparse_qp(1000, 1000)
    assert prog.n == 1000000
    st = problems.ip_state(prog, 2, 1.0)
    M = ipmatrix.IpRedSpBKP(ordering=2)
    M.init(prog)
    new = lambda: [np.zeros(k) for k in (prog.n, prog.me, prog.m, prog.m)]
    M.factor(prog, st[0], st[1])
    d1 = new()
    res = M.solve(prog, *st, *d1)
    assert res <= 1e-10
    assert abs(M.residuum(prog, *st, *d1) - res) <= 1e-13
    st2 = (st[0], st[1]) + tuple(2.0 * v for v in st[2:])
    d2 = new()
    assert M.solve(prog, *st2, *d2) <= 1e-10
    scale = max(np.abs(v).max() for v in d1)
    assert max(np.abs(b - 2.0 * a).max() for a, b in zip(d1, d2)) <= 1e-9 * scale
    M.factor(prog, st[0], st[1])
    d3 = new()
    M.solve(prog, *st, *d3)
    assert all(np.array_equal(a, b) for a, b in zip(d1, d3))
