"""GPU (-m gpu): parity of the HIP path, called through the C ABI, against the
golden vectors of the reference and against the CPU oracle on seeded inputs.

Tolerances (fp64, stated by BASELINE.json's north_star and SURVEY.md 8(c)):
  * KKT residual after solve(): within 1e-10 (absolute) of the reference's;
  * solution: ||d_hip - d_ref||inf / ||d_ref||inf <= 1e-8 on well-conditioned
    inputs (the pivot sequences differ by design, so no bit parity);
  * residuum() of a given d: 1e-12 relative (same arithmetic, other sum order).
"""
import os

import numpy as np
import pytest

from common import GOLDEN, KINDS, load_golden, new_d, rel_err
from hqp_amd import _lib, ipmatrix, problems
from oracle import oracleapi

pytestmark = pytest.mark.gpu

CLS = {"SpBKP": ipmatrix.IpSpBKP, "RedSpBKP": ipmatrix.IpRedSpBKP}
RES_TOL = 1e-10
SOL_TOL = 1e-8


def test_mfma_f64_layout():
    assert ipmatrix.selftest_mfma(0) == 0.0


@pytest.mark.parametrize("name", GOLDEN)
@pytest.mark.parametrize("kind", KINDS)
def test_against_reference_golden(name, kind):
    prog, st, g = load_golden(name)
    M = CLS[kind]()
    M.init(prog)
    assert M.mat_sbw == int(g[f"{kind}_sbw"])
    assert np.array_equal(M.perm(), g[f"{kind}_perm"])
    M.factor(prog, st[0], st[1])
    d = new_d(prog)
    res = M.solve(prog, *st, *d)
    gold = [g[f"{kind}_solve_{k}"] for k in ("dx", "dy", "dz", "dw")]
    gres = float(g[f"{kind}_res"])
    assert res <= gres + RES_TOL, (res, gres, M.stats())
    # 1e-8 on every fixture, the w/z spreads over decades included (measured: 2e-16 ... 4e-12)
    assert rel_err(d, gold) <= SOL_TOL, (rel_err(d, gold), M.stats())
    # residuum() of the reference's own step result
    gstep = [g[f"{kind}_step_{k}"] for k in ("dx", "dy", "dz", "dw")]
    r = M.residuum(prog, *st, *gstep)
    rg = float(g[f"{kind}_res_of_step"])
    assert abs(r - rg) <= 1e-12 * max(1.0, max(np.abs(v).max() if len(v) else 0 for v in gstep)) + 1e-13


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("case", [
    ("banded", (1500, 16, 31), 0.0), ("banded", (1000, 30, 32), 2.0), ("did", (700,), 1.0),
    ("banded_small_blocks", (600, 10, 33), 0.0),
])
def test_against_oracle_seeded(kind, case):
    what, args, spread = case
    kw = {}
    if what == "did":
        prog = problems.did_like_qp(*args)
    else:
        prog = problems.banded_qp(*args)
    if what == "banded_small_blocks":
        kw = dict(leaf_size=40, max_pivots=16)  # deep tree, many levels
    st = problems.ip_state(prog, 77, spread)
    M = CLS[kind](**kw)
    M.init(prog)
    M.factor(prog, st[0], st[1])
    O = oracleapi.OracleIpMatrix(kind)
    O.init(prog)
    O.factor(st[0], st[1])
    assert M.mat_sbw == O.sbw and np.array_equal(M.perm(), O.perm())
    d = new_d(prog)
    M.step(prog, *st, *d)
    ostep = O.step(*st)
    # a single solve without refinement (pivoting is restricted to the supernode's pivot block; the contract is on
    # solve() below).  Measured on these cases in round 5: 2e-15 ... 7e-11 (the double-integrator QP with w / z spread
    # over two decades), no pivot perturbed - the bound was 1e-4 / 1e-6 until then
    loose = M.stats()["n_perturbed"] > 0
    assert rel_err(d, ostep) <= (1e-6 if loose else 1e-8), (rel_err(d, ostep), M.stats())
    d2 = new_d(prog)
    res = M.solve(prog, *st, *d2)
    osol, ores = O.solve(*st)
    assert res <= ores + RES_TOL
    assert rel_err(d2, osol) <= SOL_TOL, (rel_err(d2, osol), M.stats())
    assert abs(M.residuum(prog, *st, *osol) - O.residuum(*st, *osol)) <= 1e-12


@pytest.mark.parametrize("kind", KINDS)
def test_update_and_refactor(kind):
    prog, st, _ = load_golden("banded_n300_b10")
    M = CLS[kind]()
    M.init(prog)
    O = oracleapi.OracleIpMatrix(kind)
    O.init(prog)
    for scale in (1.0, 3.0, 0.25):
        prog.Q = (prog.Q[0], prog.Q[1], prog.Q[2] * scale)
        M.update(prog), O.update(prog)
        M.factor(prog, st[0], st[1]), O.factor(st[0], st[1])
        d = new_d(prog)
        res = M.solve(prog, *st, *d)
        osol, ores = O.solve(*st)
        assert res <= ores + RES_TOL and rel_err(d, osol) <= SOL_TOL


@pytest.mark.parametrize("kind", KINDS)
def test_zero_slack_is_singular(kind):
    """v_slash raises E_SING on a zero component (meschach/vecop.c:346-348)."""
    prog, st, _ = load_golden("banded_n60_b4")
    M = CLS[kind]()
    M.init(prog)
    z = st[0].copy()
    z[5] = 0.0
    with pytest.raises(ipmatrix.SingularError):
        M.factor(prog, z, st[1])
    M.factor(prog, st[0], st[1])  # handle stays usable
    d = new_d(prog)
    assert M.solve(prog, *st, *d) < 1e-9


@pytest.mark.parametrize("kind", KINDS)
def test_device_vectors_torch(kind):
    import torch
    prog, st, g = load_golden("did_K400_spread2")
    M = CLS[kind](device_vectors=True)
    M.init(prog)
    dev = [torch.as_tensor(a).cuda() for a in st]
    M.factor(prog, dev[0], dev[1])
    d = [torch.zeros(k, dtype=torch.float64, device="cuda") for k in (prog.n, prog.me, prog.m, prog.m)]
    res = M.solve(prog, *dev, *d)
    gold = [g[f"{kind}_solve_{k}"] for k in ("dx", "dy", "dz", "dw")]
    assert res <= float(g[f"{kind}_res"]) + RES_TOL
    assert rel_err([t.cpu().numpy() for t in d], gold) <= SOL_TOL


@pytest.mark.parametrize("kind,n,band", [("SpBKP", 3000, 20), ("RedSpBKP", 6000, 40)])
def test_repeated_calls_on_the_same_device_vectors_skip_the_staging_copies(kind, n, band):
    """hqpkkt_factor / hqpkkt_solve of a caller with device vectors: the first call stages the vectors into the handle's
    buffers, from the second call in a row with the same pointers on the sequences run on the caller's vectors themselves
    (captured on them; hqpkkt.hip DirectCall).  Same kernels on the same numbers: every call gives the bits of the first
    one; new right-hand sides in the same tensors are seen; a call whose result aliases an input (dx = r1) stays on the
    staging path and gives the same solution; new tensors go back to staging."""
    import torch
    prog = problems.banded_qp(n, band, seed=5)
    st = problems.ip_state(prog, seed=2)
    M = CLS[kind](device_vectors=True)
    M.init(prog)
    dev = [torch.as_tensor(a).cuda() for a in st]
    d = [torch.zeros(k, dtype=torch.float64, device="cuda") for k in (prog.n, prog.me, prog.m, prog.m)]
    out = []
    for _ in range(4):
        M.factor(prog, dev[0], dev[1])
        for t in d:
            t.zero_()
        torch.cuda.synchronize()  # (the handle has a stream of its own: the caller's writes must be complete, hqpkkt.h)
        res = M.solve(prog, *dev, *d)
        out.append((res, [t.cpu().numpy().copy() for t in d]))
    assert out[0][0] <= 1e-10
    for res, sol in out[1:]:
        assert res == out[0][0]
        for a, b in zip(sol, out[0][1]):
            assert np.array_equal(a, b)
    # twice the right-hand side, written into the same tensors: twice the solution
    for t in dev[2:]:
        t.mul_(2.0)
    torch.cuda.synchronize()
    M.solve(prog, *dev, *d)
    assert rel_err([t.cpu().numpy() for t in d], [2.0 * v for v in out[0][1]]) <= 1e-9
    for t in dev[2:]:
        t.mul_(0.5)
    torch.cuda.synchronize()
    # the result written over r1 (same length as dx): inputs are read before outputs are written, twice in a row
    r1 = torch.empty_like(dev[2])
    for _ in range(2):
        r1.copy_(dev[2])
        alias = [r1, d[1], d[2], d[3]]
        torch.cuda.synchronize()
        res = M.solve(prog, dev[0], dev[1], r1, dev[3], dev[4], dev[5], *alias)
        assert res <= 1e-10
        assert rel_err([t.cpu().numpy() for t in alias], out[0][1]) <= 1e-9
    # fresh tensors every call
    for _ in range(2):
        dev2 = [t.clone() for t in dev]
        d2 = [torch.zeros_like(t) for t in d]
        torch.cuda.synchronize()
        M.factor(prog, dev2[0], dev2[1])
        res = M.solve(prog, *dev2, *d2)
        assert res == out[0][0]
        for a, b in zip(d2, out[0][1]):
            assert np.array_equal(a.cpu().numpy(), b)


def test_full_size_c2_properties():
    """Config C2 of BASELINE.json (n=40000, b=80: KKT dim 1e5, mat_sbw 200) at full
    size through size-independent properties: the refined KKT residual reaches
    mat_eps, solve() is linear in the right-hand side and a second factor of the
    same data reproduces the same solution."""
    prog = problems.banded_qp(40000, 80, 12345)
    st = problems.ip_state(prog, 1)
    M = ipmatrix.IpSpBKP()
    M.init(prog)
    assert M.mat_sbw == 200 and M.stats()["dim"] == 100000
    M.factor(prog, st[0], st[1])
    d = new_d(prog)
    res = M.solve(prog, *st, *d)
    assert res <= 1e-10, (res, M.stats())
    assert M.residuum(prog, *st, *d) == pytest.approx(res, rel=1e-6, abs=1e-14)
    st2 = (st[0], st[1]) + tuple(2.0 * r for r in st[2:])
    d2 = new_d(prog)
    M.solve(prog, *st2, *d2)
    assert rel_err(d2, [2.0 * v for v in d]) <= 1e-9
    M.factor(prog, st[0], st[1])
    d3 = new_d(prog)
    M.solve(prog, *st, *d3)
    assert rel_err(d3, d) <= 1e-12


FULL_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_full_size")
FULL = sorted(os.path.splitext(os.path.basename(f))[0] for f in __import__("glob").glob(os.path.join(FULL_DIR, "*.npz")))


@pytest.mark.parametrize("name", FULL)
def test_full_size_c2_against_the_reference_golden(name):
    """Config C2 of BASELINE.json at FULL size (n = 40 000, band 80: KKT dimension 10^5) against the REFERENCE's own
    Hqp_IpSpBKP / Hqp_IpRedSpBKP (committed results, tests/golden_full_size/make_golden.py: every 37th component of the
    reference's solve() result, the norms, the sums and the residual; inputs regenerated from the seeds, guarded by a
    checksum), the second fixture with w / z spread over four decades; and the 316 x 316 mesh of 10^5 variables (the
    stand-in of configs[4]; the reference factorises its RCM band of 631 in 51 s, the tree engine the dissection of the
    KKT graph); and SURVEY.md 8(d) C5's row density - problems.cute_like_qp, 10 ... 100 entries per row, at n = 6500 (the
    largest the reference finishes in about a minute: 65 s): residual of solve() <= the reference's + 1e-10,
    the sampled components to 1e-8 of the vector's norm, norms and sums to 1e-8."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_full", os.path.join(FULL_DIR, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    g = dict(np.load(os.path.join(FULL_DIR, name + ".npz")))
    case = mg.CASES[name]
    prog, st = mg.inputs(case)
    np.testing.assert_allclose(mg.checksum(prog, st), g["checksum"], rtol=1e-13)
    irregular = case[0] in ("mesh", "cute")
    M = CLS[case[5]](**(dict(ordering=2) if irregular else {}))  # (the mesh, the CUTE-style rows: the dissection of the KKT graph itself)
    M.init(prog)
    if not irregular:
        assert M.stats()["dim"] == (100000 if case[5] == "SpBKP" else 60000)
    M.factor(prog, st[0], st[1])
    d = new_d(prog)
    res = M.solve(prog, *st, *d)
    scale = max(1.0, max(float(g[k + "_norm"]) for k in ("dx", "dy", "dz", "dw")))
    assert res <= float(g["res"]) + 1e-10 * scale, (res, float(g["res"]))
    for nm, v in zip(("dx", "dy", "dz", "dw"), d):
        nrm = max(float(g[nm + "_norm"]), 1e-300)
        assert np.abs(v[::mg.STRIDE] - g[nm + "_sample"]).max() <= 1e-8 * nrm, nm
        assert abs(np.abs(v).max() - nrm) <= 1e-8 * nrm, nm
        assert abs(v.sum() - float(g[nm + "_sum"])) <= 1e-8 * nrm * max(1.0, np.sqrt(len(v))), nm


def _empty_block():
    return (np.zeros(1, np.int32), np.zeros(0, np.int32), np.zeros(0))


@pytest.mark.parametrize("kind", KINDS)
def test_edge_shapes(kind):
    """Tiny and degenerate shapes the reference's callers can produce: no
    equalities, no inequalities (Hqp_IpsMehrotra passes zero-length z, w, r3, r4,
    hqp/Hqp_IpsMehrotra.C:364-415), a 3-variable program."""
    rng = np.random.default_rng(3)
    base = problems.banded_qp(40, 3, 2)
    cases = {
        "no_eq": problems.Program(base.n, 0, base.m, base.Q, _empty_block(), base.C),
        "no_ineq": problems.Program(base.n, base.me, 0, base.Q, base.A, _empty_block()),
        "tiny": problems.banded_qp(3, 1, 1),
    }
    for name, prog in cases.items():
        if prog.m == 0:
            # without inequalities the KKT matrix [-Q A'; A 0] must have full row rank A: ok here
            pass
        st = problems.ip_state(prog, 4)
        M = CLS[kind]()
        M.init(prog)
        M.factor(prog, st[0], st[1])
        d = new_d(prog)
        res = M.solve(prog, *st, *d)
        O = oracleapi.OracleIpMatrix(kind)
        O.init(prog)
        O.factor(st[0], st[1])
        osol, ores = O.solve(*st)
        assert M.mat_sbw == O.sbw, name
        assert res <= ores + RES_TOL, (name, res, ores)
        assert rel_err(d, osol) <= 1e-7, (name, rel_err(d, osol))


def test_reinit_and_two_handles():
    """init() again with another structure on the same object (the reference
    re-inits on sqp_init) and two plugin objects alive at once."""
    a, b = problems.banded_qp(200, 6, 1), problems.did_like_qp(60)
    M1, M2 = ipmatrix.IpSpBKP(), ipmatrix.IpRedSpBKP()
    for prog in (a, b, a):
        st = problems.ip_state(prog, 9)
        for M, kind in ((M1, "SpBKP"), (M2, "RedSpBKP")):
            M.init(prog)
            M.factor(prog, st[0], st[1])
            d = new_d(prog)
            res = M.solve(prog, *st, *d)
            O = oracleapi.OracleIpMatrix(kind)
            O.init(prog)
            O.factor(st[0], st[1])
            osol, ores = O.solve(*st)
            assert res <= ores + RES_TOL and rel_err(d, osol) <= 1e-7


@pytest.mark.parametrize("shape", [(10, 6, 2), (40, 16, 4)])
def test_multistage_docp_against_reference_lqdocp(shape):
    """Config 4 structure (multistage LQ DOCP) at test size: the full-system engine,
    under the plugin name LQDOCP, against the REFERENCE's Riccati-based Hqp_IpLQDOCP
    (from oracle/_ref when it travelled to this box) and against the oracle."""
    from oracle import refapi
    K, nx, nu = shape
    prog = problems.lq_docp(K, nx, nu)
    st = problems.ip_state(prog, 3, 1.0)
    M = ipmatrix.IpLQDOCP()
    assert M.name() == "LQDOCP"
    M.init(prog)
    M.factor(prog, st[0], st[1])
    d = new_d(prog)
    res = M.solve(prog, *st, *d)
    O = oracleapi.OracleIpMatrix("SpBKP")
    O.init(prog)
    O.factor(st[0], st[1])
    osol, ores = O.solve(*st)
    assert res <= ores + RES_TOL and rel_err(d, osol) <= SOL_TOL
    if refapi.available():
        L = refapi.RefIpMatrix("LQDOCP")
        L.init(prog)
        L.factor(st[0], st[1])
        lsol, lres = L.solve(*st)
        assert res <= lres + RES_TOL and rel_err(d, lsol) <= 1e-7


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["banded", "did", "docp"])
def test_explicit_inverse_of_pivot_blocks(case):
    """k_factor_diag / k_factor_diag_small leave M = L11^-1 next to L11: the panel
    solve and both tree sweeps are products with it.  Checked block by block
    (supernodes of 1..128 pivots, 1x1 and 2x2 pivots, partial 16-blocks)."""
    prog = {"banded": lambda: problems.banded_qp(6000, 60, 2), "did": lambda: problems.did_like_qp(300),
            "docp": lambda: problems.lq_docp(30, 20, 6)}[case]()
    st = problems.ip_state(prog, 5, 1.0)
    M = ipmatrix.IpSpBKP()
    M.init(prog)
    M.factor(prog, st[0], st[1])
    s = M.structure()
    assert s["npiv"].max() > (100 if case == "banded" else 16)
    worst = 0.0
    for node in range(len(s["npiv"])):
        p, b = int(s["npiv"][node]), int(s["nborder"][node])
        P = M.read_block(0, node).reshape(p, p + b).T
        L = np.tril(P[:p, :p], -1) + np.eye(p)
        W = M.read_block(1, node).reshape(p, p).T
        for kb in range(0, p, 16):  # the diagonal blocks are complete: zeros above the diagonal
            assert not np.triu(W[kb:kb + 16, kb:kb + 16], 1).any()
        worst = max(worst, float(np.abs(np.tril(W) @ L - np.eye(p)).max()))
    assert worst < 1e-12, worst


@pytest.mark.gpu
def test_slack_order_policies_agree():
    """The order of the slack rows inside a supernode (hqpkkt_opts.slack_policy) changes
    which pivots need a run-time interchange, not the solution."""
    prog = problems.banded_qp(3000, 24, 4)
    st = problems.ip_state(prog, 9, 1.0)
    sols, slow = [], []
    for sp in (0, 1, 2):
        M = ipmatrix.IpSpBKP(slack_policy=sp)
        M.init(prog)
        M.factor(prog, st[0], st[1])
        d = new_d(prog)
        res = M.solve(prog, *st, *d)
        assert res <= 1e-10
        sols.append(d)
        slow.append(M.stats()["n_slow_pivots"])
    assert rel_err(sols[1], sols[0]) < 1e-9 and rel_err(sols[2], sols[0]) < 1e-9
    assert slow[2] <= slow[0] and slow[1] <= slow[0]  # behind-its-x orders avoid the interchanges
    assert slow[0] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("kind", KINDS)
def test_fused_small_fronts_agree_with_the_general_kernels(kind):
    """Fronts with <= 32 pivots and <= 16 border rows go through one-wavefront kernels that
    fuse extend-add, pivot block, panel and update (and the two solve kernels of a sweep);
    hqpkkt_opts.no_small_fronts sends them through the general kernels instead.  Same
    factorisation: same pivots, solutions equal to rounding."""
    cls = ipmatrix.IpSpBKP if kind == "SpBKP" else ipmatrix.IpRedSpBKP
    for prog, spread in ((problems.did_like_qp(400), 0.0), (problems.did_like_qp(400), 4.0),
                         (problems.banded_qp(500, 3, 8), 1.0)):
        st = problems.ip_state(prog, 5, spread)
        sol, piv = [], []
        for sf in (False, True):
            M = cls(small_fronts=sf)
            M.init(prog)
            M.factor(prog, st[0], st[1])
            d = new_d(prog)
            M.step(prog, *st, *d)
            sol.append(d)
            piv.append((M.stats()["n_2x2"], M.stats()["n_perturbed"]))
        assert piv[0] == piv[1]
        assert rel_err(sol[1], sol[0]) < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("kind", KINDS)
def test_amalgamated_separators_agree(kind):
    """Narrow bands, hqpkkt_opts.amalgamation: a separator absorbs its child separators while
    the merged pivot set still fits a small front.  Fewer tree levels, every front still a
    small front, the same solution to rounding."""
    cls = ipmatrix.IpSpBKP if kind == "SpBKP" else ipmatrix.IpRedSpBKP
    for prog, spread in ((problems.did_like_qp(600), 0.0), (problems.did_like_qp(600), 4.0),
                         (problems.banded_qp(2000, 3, 8), 1.0)):
        st = problems.ip_state(prog, 5, spread)
        sol, lev = [], []
        for am in (False, True):
            M = cls(amalgamation=am)
            M.init(prog)
            M.factor(prog, st[0], st[1])
            d = new_d(prog)
            res = M.solve(prog, *st, *d)
            assert res <= (1e-10 if spread <= 1.0 else 1e-8)  # z/w spread over eight decades: five rounds may not do
            sol.append(d)
            lev.append(M.stats()["n_levels"])
            if am:
                p, b = np.array(M.debug(2)), np.array(M.debug(3))
                assert p.max() <= 32 and b.max() <= 16
        assert lev[1] < lev[0]
        assert rel_err(sol[1], sol[0]) < 1e-8


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["banded", "random", "did"])
def test_pingpong_update_arena_agrees(case):
    """Large systems do not keep every supernode's update block for the whole
    factorisation (hqpkkt_opts.upd_pingpong_mb): levels as late as possible, two
    alternating half-arenas.  Forced here on small systems: same solution, less memory."""
    prog = {"banded": lambda: problems.banded_qp(6000, 60, 2),
            "random": lambda: problems.random_sparse_qp(1500, 700, 1500, 4, 5),
            "did": lambda: problems.did_like_qp(500)}[case]()
    st = problems.ip_state(prog, 5, 1.0)
    sols, mem = [], []
    for mb in (-1, 1):
        M = ipmatrix.IpSpBKP(upd_pingpong_mb=mb)
        M.init(prog)
        M.factor(prog, st[0], st[1])
        d = new_d(prog)
        res = M.solve(prog, *st, *d)
        assert res <= 1e-10
        sols.append(d)
        mem.append(M.stats()["bytes_updates"])
    assert rel_err(sols[1], sols[0]) < 1e-9
    # a chain (RCM of a random sparse system) keeps two blocks instead of all of them; a
    # balanced tree still holds its two fullest levels
    if case == "random":
        assert mem[1] < 0.25 * mem[0]
    elif case == "banded":
        assert mem[1] < 0.8 * mem[0]


@pytest.mark.parametrize("case", [
    ("banded", (1067, 29, 465), "SpBKP", 2.0, dict(leaf_size=8, max_pivots=128)),
    ("docp", (41, 4, 2, 56, 1.0), "SpBKP", 0.0, dict(leaf_size=8, max_pivots=16)),
    ("sparse", (281, 36, 26, 2, 496), "RedSpBKP", 0.0, dict(leaf_size=8, max_pivots=128)),
])
def test_leaves_of_multipliers_only(case):
    """Leaves so small that all their rows are equality multipliers, which move up to the node
    of their last neighbour: the emptied node leaves the tree (found by tools/fuzz.py: the row
    used to stay behind as a 1x1 leaf with an exactly zero pivot)."""
    what, args, kind, spread, kw = case
    prog = {"banded": problems.banded_qp, "docp": problems.lq_docp, "sparse": problems.random_sparse_qp}[what](*args)
    st = problems.ip_state(prog, 5, spread)
    M = CLS[kind](**kw)
    M.init(prog)
    M.factor(prog, st[0], st[1])
    d = new_d(prog)
    res = M.solve(prog, *st, *d)
    O = oracleapi.OracleIpMatrix(kind)
    O.init(prog)
    O.factor(st[0], st[1])
    osol, ores = O.solve(*st)
    assert np.array_equal(M.perm(), O.perm())
    assert res <= ores + RES_TOL, (res, ores, M.stats())
    assert rel_err(d, osol) <= SOL_TOL, (rel_err(d, osol), M.stats())


@pytest.mark.gpu
def test_weak_hessian_test_runs_on_every_update():
    """zd_policy -1: whether some x has a Hessian diagonal that is weak against its coupling to an
    equality is decided from the VALUES, on every update() (an SQP run starts from an identity Hessian and
    may get weak ones later), so that a solve whose refinement fails can still switch the placement."""
    strong, weak = problems.did_like_qp(200, qx=1.0), problems.did_like_qp(200, qx=1e-4)
    M = ipmatrix.IpSpBKP()
    M.init(strong)
    assert list(M.debug(30)) == [2, 0]
    M.update(weak)
    assert list(M.debug(30)) == [2, 1]
    st = problems.ip_state(weak, 3, 1.0)
    M.factor(weak, st[0], st[1])
    d = new_d(weak)
    assert M.solve(weak, *st, *d) <= 1e-10
    M.update(strong)
    assert M.debug(30)[1] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["SpBKP", "RedSpBKP"])
def test_duplicated_equality_row_is_e_sing_like_the_reference(kind):
    """A rank-deficient equality block: the reference's search over the whole remaining column ends in an
    exactly zero pivot -> E_SING in solve (hqp/spBKP.C:699-700) for both plugins.  Here the full plugin is
    left with a pivot of ~1e-17 on the second multiplier (no exact zero inside the pivot block): marked
    soft-singular, and the solve whose refinement cannot reach mat_eps reports E_SING as well."""
    prog = problems.banded_qp(1500, 12)
    p, i, x = prog.A
    r = 1500 // 8
    i, x = i.copy(), x.copy()
    i[p[r + 1]:p[r + 2]], x[p[r + 1]:p[r + 2]] = i[p[r]:p[r + 1]], x[p[r]:p[r + 1]]
    bad = problems.Program(prog.n, prog.me, prog.m, prog.Q, (p, i, x), prog.C)
    st = problems.ip_state(bad, 7, 1.0)
    O = oracleapi.OracleIpMatrix(kind)
    O.init(bad)
    with pytest.raises(Exception):
        O.factor(st[0], st[1])
        O.solve(*st)
    M = {"SpBKP": ipmatrix.IpSpBKP, "RedSpBKP": ipmatrix.IpRedSpBKP}[kind]()
    M.init(bad)
    with pytest.raises(ipmatrix.SingularError):
        M.factor(bad, st[0], st[1])
        M.solve(bad, *st, *new_d(bad))
