"""CPU tests of the STAGED engine's host side and of its numpy model: stage detection
(semantics of Hqp_IpLQDOCP::Get_Dim / Get_Constr_Dim / Check_Structure), the model of the
recursion against the reference's own Hqp_IpLQDOCP (oracle/_ref, where present) and against a
dense solve of the KKT system, the dense hand-over's analysis, the column cuts of a sharded
system.  hqpkkt_analyze is host-only: no GPU needed."""
import ctypes as C

import numpy as np
import pytest

from hqp_amd import _lib, ipmatrix, problems
from model import dense_blocks
from model_staged import StagedModel, StageError, kkt_residual, stage_structure

CASES = {
    "plain": lambda: problems.lq_docp(10, 6, 2),
    "final5": lambda: problems.lq_docp(12, 5, 3, final_eq=5),
    "mix": lambda: problems.lq_docp(12, 5, 3, path_eq=2, final_eq=3, x_bounds=2),
    "free_x0": lambda: problems.lq_docp(8, 4, 2, x0_fixed=False, final_eq=2),
    "did": lambda: problems.did_like_qp(50),
    "nu0_last": lambda: problems.lq_docp(4, 3, 1, final_eq=1),
}


def _analyze(M, prog):
    arrs = []
    for (p, i, _x) in (prog.Q, prog.A, prog.C):
        arrs += [np.ascontiguousarray(p, dtype=np.int32), np.ascontiguousarray(i, dtype=np.int32)]
    sbw = C.c_int()
    ptrs = [C.c_void_p(a.ctypes.data) if a.size else None for a in arrs]
    return M._L.hqpkkt_analyze(M._h, prog.n, prog.me, prog.m, *ptrs, C.byref(sbw)), sbw.value


@pytest.mark.parametrize("case", sorted(CASES))
def test_stage_plan_equals_the_model(case):
    prog = CASES[case]()
    M = ipmatrix.IpLQDOCP()
    e, sbw = _analyze(M, prog)
    assert e == 0 and sbw == -1
    S, R = M.stage_structure(), stage_structure(prog.n, prog.me, prog.m, prog.A, prog.Q, prog.C)
    assert list(S["nk"]) == R["nk"] and list(S["mk"]) == R["mk"] and list(S["nmk"]) == R["nmk"]
    assert list(S["eq_rows"]) == [r for rows in R["eq_rows"] for r in rows]
    assert list(S["fix_rows"]) == R["fix_rows"]
    assert list(np.diff(S["eq_ptr"])) == [len(r) for r in R["eq_rows"]]
    # capacity of carried rows: own rows + what the next stage may hand back, at most 256
    e_k = [len(r) for r in R["eq_rows"]]
    cap = list(S["cap"])
    assert cap[-1] == e_k[-1] and all(cap[k] == min(e_k[k] + cap[k + 1], 256) for k in range(len(cap) - 1))


def test_not_a_staircase_is_e_format():
    M = ipmatrix.IpLQDOCP()
    e, _ = _analyze(M, problems.banded_qp(120, 5, 1))
    assert e == _lib.E_FORMAT
    with pytest.raises(StageError):
        p = problems.banded_qp(120, 5, 1)
        stage_structure(p.n, p.me, p.m, p.A, p.Q, p.C)
    # a Q entry that couples two stages (Check_Structure, hqp/Hqp_IpLQDOCP.C:332-339)
    prog = problems.lq_docp(4, 3, 2)
    qp, qi, qx = prog.Q
    rows = np.repeat(np.arange(prog.n), np.diff(qp))
    bad = problems._csr(np.append(rows, 0), np.append(qi, 7), np.append(qx, 0.5), prog.n)
    e, _ = _analyze(M, problems.Program(prog.n, prog.me, prog.m, bad, prog.A, prog.C))
    assert e == _lib.E_FORMAT


def test_size_limits_of_the_staged_engine():
    """Round 3: 150 controls per stage are accepted (the control-sized elimination then runs out of global memory,
    StagedPlan::big); beyond 512 controls, beyond 256 carried constraint rows, or with a free initial state of more than
    4096 components (round 4: the blocked inverse of [V_0 B_0'; B_0 0]; 1024 before): HQPKKT_E_SIZES (the shim's
    fall-back to the tree engine)."""
    M = ipmatrix.IpLQDOCP()
    e, _ = _analyze(M, problems.lq_docp(2, 4, 150))
    assert e == 0
    e, _ = _analyze(ipmatrix.IpLQDOCP(), problems.lq_docp(2, 4, 513))
    assert e == _lib.E_SIZES
    e, _ = _analyze(ipmatrix.IpLQDOCP(), problems.lq_docp(2, 300, 2, final_eq=280))
    assert e == _lib.E_SIZES
    e, _ = _analyze(ipmatrix.IpLQDOCP(), problems.lq_docp(2, 1100, 2, x0_fixed=False))
    assert e == 0
    e, _ = _analyze(ipmatrix.IpLQDOCP(), problems.lq_docp(2, 4100, 1, x0_fixed=False))
    assert e == _lib.E_SIZES
    e, _ = _analyze(ipmatrix.IpLQDOCP(), problems.lq_docp(2, 1100, 2))  # fixed x_0: no limit on the states
    assert e == 0


@pytest.mark.parametrize("case", sorted(CASES))
def test_model_against_dense_solve_and_reference(case):
    from oracle import refapi
    prog = CASES[case]()
    st = problems.ip_state(prog, 3, 1.0)
    R = StagedModel(prog)
    R.factor(st[0], st[1])
    d = R.step(*st[2:])
    assert kkt_residual(prog, st[0], st[1], st[2:], d) <= 1e-9
    # dense solve of the reduced system
    Q, A, Cm = dense_blocks(prog)
    n, me, m = prog.dims
    H = Q + Cm.T @ np.diag(st[0] / st[1]) @ Cm
    K = np.block([[-H, A.T], [A, np.zeros((me, me))]])
    z, w, r1, r2, r3, r4 = st
    g = r1 - Cm.T @ ((r4 + z * r3) / w)
    sol = np.linalg.solve(K, np.concatenate([g, r2]))
    assert np.abs(sol[:n] - d[0]).max() <= 1e-8 * max(1.0, np.abs(sol[:n]).max())
    if refapi.available():
        L = refapi.RefIpMatrix("LQDOCP")
        L.init(prog)
        L.factor(st[0], st[1])
        ls, _ = L.solve(*st)
        err = max(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300) for a, b in zip(d, ls) if len(b))
        assert err <= 1e-9, err


def test_dense_handover_analysis():
    K, nx, nu = 6, 5, 2
    prog = problems.lq_docp(K, nx, nu, final_eq=2, path_eq=1, path_eq_every=2)
    dq = problems.dense_docp_from_program(prog, [nx] * (K + 1), [nu] * K)
    assert dq.dims == prog.dims
    M = ipmatrix.IpLQDOCP()
    nxa, nua = np.asarray(dq.nx, np.int32), np.asarray(dq.nu, np.int32)
    arrs = []
    for (p, i, _x) in (dq.Q, dq.E, dq.C):
        arrs += [np.ascontiguousarray(p, dtype=np.int32), np.ascontiguousarray(i, dtype=np.int32)]
    e = M._L.hqpkkt_analyze_staged(M._h, K, C.c_void_p(nxa.ctypes.data), C.c_void_p(nua.ctypes.data), dq.n, dq.me_rest, dq.m,
                                   *[C.c_void_p(a.ctypes.data) if a.size else None for a in arrs])
    assert e == 0
    S, R = M.stage_structure(), stage_structure(prog.n, prog.me, prog.m, prog.A, prog.Q, prog.C)
    assert list(S["nk"]) == R["nk"] and list(S["mk"]) == R["mk"]
    assert list(S["eq_rows"]) == [r for rows in R["eq_rows"] for r in rows] and list(S["fix_rows"]) == R["fix_rows"]
    # wrong sizes
    bad = nxa.copy()
    bad[2] += 1
    e = M._L.hqpkkt_analyze_staged(M._h, K, C.c_void_p(bad.ctypes.data), C.c_void_p(nua.ctypes.data), dq.n, dq.me_rest, dq.m,
                                   *[C.c_void_p(a.ctypes.data) if a.size else None for a in arrs])
    assert e == _lib.E_SIZES


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_column_cuts_of_a_sharded_system(world):
    """One contiguous range of state columns per rank - the same width, a multiple of 128, for every rank but the last -
    and the memory goes with it: the rank's arenas hold its columns of F_k and its rows of V_k (hqpkkt_stats.bytes_panels
    <= 1 / P of the unsharded figure + 10 %).  The blocks of G_xx are dealt out in a ring, every rank the same number
    (flops_local within a few per cent of each other)."""
    K, nx, nu = 3, 5000, 50
    n = K * (nx + nu) + nx
    # pattern only: dense staircase rows would be 12 M entries; use the explicit sizes + dense hand-over analysis
    Q = (np.arange(n + 1, dtype=np.int32), np.arange(n, dtype=np.int32))
    E = (np.arange(nx + 1, dtype=np.int32), np.arange(nx, dtype=np.int32))
    nxa, nua = np.full(K + 1, nx, np.int32), np.full(K, nu, np.int32)

    def analyse(shard):
        M = ipmatrix.IpLQDOCP(**({"shard": shard} if shard else {}))
        e = M._L.hqpkkt_analyze_staged(M._h, K, C.c_void_p(nxa.ctypes.data), C.c_void_p(nua.ctypes.data), n, nx, 0,
                                       C.c_void_p(Q[0].ctypes.data), C.c_void_p(Q[1].ctypes.data),
                                       C.c_void_p(E[0].ctypes.data), C.c_void_p(E[1].ctypes.data), None, None)
        assert e == 0
        return M

    whole = analyse(None).stats()["bytes_panels"]
    cuts, flops = [], []
    for rank in range(world):
        M = analyse((rank, world, lambda *a: None))
        cuts.append(M.debug(27).reshape(K + 1, world + 1))
        st = M.stats()
        assert st["shard_count"] == world and st["bytes_exchange_factor"] > 0
        assert st["bytes_panels"] <= whole * (1.0 / world) * 1.10, (rank, st["bytes_panels"], whole)
        flops.append(st["flops_local"])
    for c in cuts[1:]:
        assert np.array_equal(cuts[0], c)  # every rank derives the same plan
    c = cuts[0][0]
    assert c[0] == 0 and c[-1] == nx and np.all(np.diff(c) >= 0) and np.all(c[:-1] % 128 == 0)
    wd = np.diff(c)
    assert np.all(wd[:-1] == wd[0]) and 0 < wd[-1] <= wd[0]
    assert max(flops) / (sum(flops) / world) < 1.10, flops  # (the last strip is the narrow one: 40 tiles over 3 ranks = 14 + 14 + 12)


def test_fractional_cut_partition_properties():
    """The index arithmetic of the fractional cut (k_dgemm_tn_sk with SplitPlan::frac, hqp_amd/csrc/staged.hip.h) restated in
    Python: the k-slabs of all tiles as one sequence, `per` units per workgroup.  Every unit is computed exactly once; a
    tile's sharers are consecutive workgroups w_first .. w_last and their k ranges follow each other in that order (the
    order in which the last arriver adds the parked pieces); a workgroup parks at most two partial tiles and no two
    pieces share a slot (2 w: the tile its range starts in, 2 w + 1: the tile it ends in)."""
    for tiles, nslab, grid in ((272, 125, 512), (300, 188, 512), (200, 313, 512), (160, 64, 512), (320, 400, 512), (7, 64, 512), (45, 63, 512)):
        U = tiles * nslab
        per = (U + grid - 1) // grid
        covered = np.zeros(U, dtype=np.int32)
        slots = {}
        for w in range(grid):
            lo = min(U, w * per)
            hi = min(U, lo + per)
            t_first = lo // nslab
            x, parked = lo, 0
            while x < hi:
                t = x // nslab
                s0 = x - t * nslab
                s1 = min(nslab, s0 + (hi - x))
                covered[t * nslab + s0:t * nslab + s1] += 1
                w_first, w_last = (t * nslab) // per, ((t + 1) * nslab - 1) // per
                assert w_first <= w <= w_last
                if w_last > w_first:  # shared tile: this piece is parked
                    slot = 2 * w + (0 if t == t_first else 1)
                    assert slot not in slots
                    slots[slot] = (t, s0, s1)
                    parked += 1
                x += s1 - s0
            assert parked <= 2
        assert (covered == 1).all()
        # what the last arriver of a tile reads: the slots of w_first .. w_last, k ranges in order, together the whole tile
        for t in range(tiles):
            w_first, w_last = (t * nslab) // per, ((t + 1) * nslab - 1) // per
            if w_last == w_first:
                continue
            at = 0
            for w in range(w_first, w_last + 1):
                slot = 2 * w + (0 if t == (w * per) // nslab else 1)
                tt, s0, s1 = slots[slot]
                assert tt == t and s0 == at and s1 > s0
                at = s1
            assert at == nslab
