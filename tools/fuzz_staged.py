"""Randomised sweep of the STAGED engine (plugin LQDOCP) against the REFERENCE's own Hqp_IpLQDOCP
(oracle/_ref) and the CPU oracle of the full system (test infrastructure, like tests/): random
multistage QPs - stages, states, controls, fixed / free initial state, final-state constraints (carried
back through the stages), path equalities, state bounds, w/z spreads.  For each: the refined residual
within 1e-10 of the reference's, the solutions equal through the residual; a slice re-checks update().
Usage: python tools/fuzz_staged.py [cases] [seed0]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from hqp_amd import ipmatrix, problems
from oracle import oracleapi, refapi


def make_case(case):
    rng = np.random.default_rng(5000 + case)
    K = int(rng.integers(1, 40))
    nx = int(rng.choice([1, 2, 3, 5, 8, 13, 20, 33, 50, 70]))
    nu = int(rng.integers(1, 10))
    x0_fixed = bool(rng.random() < 0.7)
    final_eq = int(rng.integers(0, nx + 1)) if rng.random() < 0.5 else 0
    path_eq = int(rng.integers(1, 3)) if rng.random() < 0.4 else 0
    path_eq = min(path_eq, nu)
    every = int(rng.integers(1, 4))
    x_bounds = int(rng.integers(0, nx + 1)) if rng.random() < 0.4 else 0
    spread = float(rng.choice([0.0, 1.0, 1.0, 2.0, 3.0]))
    kw = dict(seed=int(rng.integers(1, 999)), x0_fixed=x0_fixed, final_eq=final_eq, path_eq=path_eq, path_eq_every=every,
              x_bounds=x_bounds)
    prog = problems.lq_docp(K, nx, nu, **kw)
    st = problems.ip_state(prog, case, spread)
    return prog, st, f"case {case}: K={K} nx={nx} nu={nu} spread={spread} {kw}"


def check(case, use_ref=True):
    """-> (status, detail): 'ok', 'skip' (the reference does not solve it either) or 'BAD'"""
    prog, st, tag = make_case(case)
    O = oracleapi.OracleIpMatrix("RedSpBKP")
    O.init(prog)
    try:
        O.factor(st[0], st[1])
        osol, ores = O.solve(*st)
    except oracleapi.OracleError:
        ores = float("inf")
    lres, ls = None, None
    if use_ref and refapi.available():
        L = refapi.RefIpMatrix("LQDOCP")
        try:
            L.init(prog)
            L.factor(st[0], st[1])
            ls, lres = L.solve(*st)
        except Exception:
            lres = float("inf")
    M = ipmatrix.IpLQDOCP()
    try:
        M.init(prog)
    except ipmatrix.KktError as e:
        if e.code == 1:  # E_SIZES: a stage carries more rows than the kernels hold -> the shim's fall-back
            return "skip", tag + " E_SIZES"
        return "BAD", tag + f" init {e}"
    d = [np.zeros(k) for k in (prog.n, prog.me, prog.m, prog.m)]
    try:
        M.factor(prog, st[0], st[1])
        res = M.solve(prog, *st, *d)
    except ipmatrix.KktError as e:
        if e.code == 1:  # E_SIZES at run time (ranks decided by the values): the shim's fall-back
            return "skip", tag + " E_SIZES in factor"
        if ores > 1e-8 and (lres is None or lres > 1e-8):
            return "skip", tag + f" both fail ({e.code})"
        return "BAD", tag + f" raised {e} where the reference solves (oracle {ores:.1e}, LQDOCP {lres})"
    # the scale of the comparison comes from the reference's solution, never from ours
    rsol = ls if (lres is not None and np.isfinite(lres)) else (osol if np.isfinite(ores) else None)
    if rsol is None:
        return "skip", tag + " reference fails"
    scale = max(1.0, max((np.abs(v).max() if len(v) else 0.0) for v in rsol))
    # the partner is the reference's Hqp_IpLQDOCP where it is present: on QPs whose stage constraints make the
    # closed loop unstable its recursion (the same elimination order as ours) ends at garbage while a pivoted
    # factorisation of the full system still solves - nothing to compare then
    best = lres if lres is not None else ores
    if not (best <= 1e-8 * scale and scale < 1e12):
        return "skip", tag + " reference residual too large"
    r = O.residuum(*st, *d)
    if not (r <= best + 1e-10 * scale):
        return "BAD", tag + f" residual {r:.3e} (ours says {res:.3e}) vs reference {best:.3e}"
    return "ok", tag


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0 = time.time()
    cnt = {"ok": 0, "skip": 0, "BAD": 0}
    why = {}  # skipped cases by reason (E_SIZES must stay at 0: every generated stage size has an engine)
    for case in range(seed0, seed0 + ncases):
        s, detail = check(case)
        cnt[s] += 1
        if s == "skip":
            reason = "E_SIZES" if "E_SIZES" in detail else detail.rsplit("} ", 1)[-1].split(" (")[0]
            why[reason] = why.get(reason, 0) + 1
        if s == "BAD":
            print(detail, flush=True)
    print(f"fuzz_staged: {ncases} cases from {seed0}: {cnt} in {time.time() - t0:.0f} s; skipped because: {why}", flush=True)


if __name__ == "__main__":
    main()
