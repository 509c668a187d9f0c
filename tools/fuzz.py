"""Randomised sweep of the HIP path against the CPU oracle (test infrastructure, like tests/):
random structures (banded, DID-like, multistage DOCP, unstructured sparse), sizes, w/z spreads,
plugin kinds and tree options; for each: perm / mat_sbw equal, residual of solve() within 1e-10 of
the oracle's, solutions agree through the residual.  Usage: python tools/fuzz.py [cases] [seed0]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hqp_amd import problems, ipmatrix
from oracle import oracleapi

CLS = {"SpBKP": ipmatrix.IpSpBKP, "RedSpBKP": ipmatrix.IpRedSpBKP}
SCALE = int(os.environ.get("FUZZ_SCALE", "1"))  # > 1: larger banded / DID systems (the oracle then takes seconds per case)


def make_case(case):
    """-> (prog, state, kind, opts, tag) of fuzz case number ``case``"""
    rng = np.random.default_rng(1000 + case)
    what = rng.choice(["banded", "did", "docp", "sparse"])
    if what == "banded":
        args = (int(rng.integers(20, 1500 * SCALE)), int(rng.integers(1, 40 * min(SCALE, 3))), int(rng.integers(1, 1000)))
        args = (max(args[0], 2 * args[1] + 2),) + args[1:]
        prog = problems.banded_qp(*args)
    elif what == "did":
        args = (int(rng.integers(2, 600 * SCALE)),)
        prog = problems.did_like_qp(*args)
    elif what == "docp":
        args = (int(rng.integers(2, 60)), int(rng.integers(1, 12)), int(rng.integers(1, 6)), int(rng.integers(1, 99)),
                float(rng.choice([1.0, 0.5, 0.25])))
        prog = problems.lq_docp(*args)
        if np.abs(prog.A[2]).max() > 1e6:  # a sparse fx with spectral radius ~0 was scaled to 1e11
            args = args[:4] + (1.0,)
            prog = problems.lq_docp(*args)
    else:
        n = int(rng.integers(5, 800))
        args = (n, int(rng.integers(0, max(1, n // 2))), int(rng.integers(0, n)), int(rng.integers(1, 6)),
                int(rng.integers(1, 999)))
        prog = problems.random_sparse_qp(*args)
    spread = float(rng.choice([0.0, 0.0, 1.0, 2.0, 4.0]))
    kind = str(rng.choice(["SpBKP", "RedSpBKP"]))
    kw = {}
    if rng.random() < 0.3:
        kw = dict(leaf_size=int(rng.choice([8, 24, 40, 100])), max_pivots=int(rng.choice([4, 16, 48, 128])))
    if rng.random() < 0.3:
        kw["amalgamation"] = True
    if os.environ.get("FUZZ_ORDERING"):  # every case through the tree of the graph's own dissection (opts.ordering 1 / 2)
        kw["ordering"] = int(os.environ["FUZZ_ORDERING"])
    st = problems.ip_state(prog, case, spread)
    return prog, st, kind, kw, f"case {case}: {what}{args} {kind} spread {spread} {kw}"


def check(case):
    """-> (status, detail) of fuzz case number ``case``: 'ok', 'unsolved' (the reference does not solve the system
    either; both sides agree on that) or 'BAD'"""
    prog, st, kind, kw, tag = make_case(case)
    status = "ok"
    try:
        M = CLS[kind](**kw)
        M.init(prog)
        M.factor(prog, st[0], st[1])
        d = [np.zeros(k) for k in (prog.n, prog.me, prog.m, prog.m)]
        res = M.solve(prog, *st, *d)
        O = oracleapi.OracleIpMatrix(kind)
        O.init(prog)
        O.factor(st[0], st[1])
        osol, ores = O.solve(*st)
        # (ordering 2 skips the reference's RCM pass: no mat_sbw / permutation to compare)
        ok = kw.get("ordering") == 2 or (M.mat_sbw == O.sbw and np.array_equal(M.perm(), O.perm()))
        # the oracle's residual of OUR solution (independent arithmetic); a solve whose last damped
        # refinement step is rejected returns the residual of that trial (hqp/Hqp_IpMatrix.C:104-121),
        # on both sides, so judge the solution through rchk
        rchk = O.residuum(*st, *d)
        scale = max(1.0, max((np.abs(v).max() if len(v) else 0.0) for v in d))
        if ores > 1e-8 * scale:  # the reference does not solve this system either
            status = "unsolved"
            ok = ok and (rchk <= 10 * ores or res <= 10 * ores)
        else:
            ok = ok and rchk <= ores + 1e-10 * scale
        if not ok:
            return "BAD", f"MISMATCH {tag} res {res} oracle {ores} check {rchk} {M.stats()}"
        # update(): new values on the same pattern (scaled blocks, as an SQP iteration changes them),
        # then factor + solve again on the same handle
        f = 1.0 + 0.5 * np.sin(np.arange(7) + case)
        prog2 = problems.Program(prog.n, prog.me, prog.m, (prog.Q[0], prog.Q[1], prog.Q[2] * f[0]),
                                 (prog.A[0], prog.A[1], prog.A[2] * (1.0 + 0.1 * np.cos(np.arange(len(prog.A[2])) + case))),
                                 (prog.C[0], prog.C[1], prog.C[2] * f[2]), c=prog.c, b=prog.b, d=prog.d)
        M.update(prog2)
        M.factor(prog2, st[0], st[1])
        d2 = [np.zeros(k) for k in (prog.n, prog.me, prog.m, prog.m)]
        res2 = M.solve(prog2, *st, *d2)
        O.update(prog2)
        O.factor(st[0], st[1])
        osol2, ores2 = O.solve(*st)
        scale2 = max(1.0, max((np.abs(v).max() if len(v) else 0.0) for v in d2))
        if ores2 <= 1e-8 * scale2 and not O.residuum(*st, *d2) <= ores2 + 1e-10 * scale2:
            return "BAD", f"MISMATCH after update() {tag} res {res2} oracle {ores2} check {O.residuum(*st, *d2)}"
        return status, tag
    except Exception as e:  # both sides must agree on singular systems
        try:
            O = oracleapi.OracleIpMatrix(kind)
            O.init(prog)
            O.factor(st[0], st[1])
            _, ores = O.solve(*st)
            if ores > 1e-8:
                return "unsolved", f"singular here, unsolved there {tag} oracle res {ores}"
            return "BAD", f"ONLY-HIP-FAILED {tag} {e!r} oracle res {ores}"
        except Exception as e2:
            return "unsolved", f"both failed {tag} {repr(e)[:60]} {repr(e2)[:60]}"


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = nsing = 0
    t0 = time.time()
    for case in range(seed0, seed0 + ncases):
        status, detail = check(case)
        bad += status == "BAD"
        nsing += status == "unsolved"
        if status == "BAD" or detail.startswith(("singular", "both failed")):
            print(detail, flush=True)
    print(f"{ncases} cases from {seed0}: {bad} bad, {nsing} not solved by the reference either, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
