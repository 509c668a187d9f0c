"""Randomised sweep of the STAGED engine on stage shapes beyond tools/fuzz_staged.py (which stops at 9 controls):
10 ... 300 controls, path equalities that consume a part of them, final-state rows carried back through the stages,
fixed / free initial state, w/z spreads - K of order <= 64 in registers, 65 ... 136 in LDS (diagonal-first or with the
search), beyond that the blocked elimination - against the tree engine (Hqp_IpLQDOCPFull) on the same QP: residual of
the refined solve <= 1e-10, same solution to 1e-8; the blocked elimination must not have fallen back.
FUZZ_X0=1: free initial states of 140 ... 1600 components with few controls (the blocked inverse of [V_0 B_0'; B_0 0],
k_x0_*; counted separately: ran / fell back to the LU factors of one workgroup).
Usage: [FUZZ_LARGE=1 | FUZZ_X0=1] python tools/fuzz_bigstage.py [cases] [seed0]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

from hqp_amd import ipmatrix, problems
from common import new_d, rel_err


def make_case(case):
    rng = np.random.default_rng(77000 + case)
    large = bool(os.environ.get("FUZZ_LARGE"))  # up to the engine's limits: 512 controls, x_0 systems of ~1000
    if os.environ.get("FUZZ_X0"):
        nx, nu, K = int(rng.integers(140, 1601)), int(rng.integers(1, 9)), int(rng.integers(2, 4))
        kw = dict(seed=int(rng.integers(1, 999)), x0_fixed=False, final_eq=int(rng.integers(0, 8)) if rng.random() < 0.6 else 0,
                  path_eq=int(rng.integers(1, nu + 1)) if rng.random() < 0.3 else 0, path_eq_every=int(rng.integers(1, 3)),
                  x_bounds=int(rng.integers(0, nx + 1)) if rng.random() < 0.5 else 0)
        prog = problems.lq_docp(K, nx, nu, **kw)
        st = problems.ip_state(prog, case, float(rng.choice([0.0, 1.0, 2.0, 3.0])))
        return prog, st, f"case {case}: K={K} nx={nx} nu={nu} {kw}"
    nu = int(rng.choice([rng.integers(10, 65), rng.integers(65, 137), rng.integers(137, 513 if large else 301)]))
    nx = int(rng.integers(20, 900 if large else 260))
    K = int(rng.integers(2, 4 if large else 5))
    path_eq = int(rng.integers(1, max(2, min(nu, 70)))) if rng.random() < 0.6 else 0
    final_eq = int(rng.integers(1, max(2, min(nx, 60)))) if rng.random() < 0.5 else 0
    x0_fixed = bool(rng.random() < 0.7)
    x_bounds = int(rng.integers(0, nx + 1)) if rng.random() < 0.3 else 0
    kw = dict(seed=int(rng.integers(1, 999)), x0_fixed=x0_fixed, path_eq=path_eq, final_eq=final_eq, x_bounds=x_bounds,
              path_eq_every=int(rng.integers(1, 3)))
    prog = problems.lq_docp(K, nx, nu, **kw)
    st = problems.ip_state(prog, case, float(rng.choice([0.0, 1.0, 2.0, 3.0])))
    return prog, st, f"case {case}: K={K} nx={nx} nu={nu} {kw}"


def solve(M, prog, st):
    M.init(prog)
    M.factor(prog, st[0], st[1])
    d = new_d(prog)
    return d, M.solve(prog, *st, *d)


def check(case, tally=None):
    """-> (status, detail) of case number ``case``: 'ok', 'skip' (nothing to compare with: the tree engine or the
    reference does not solve it either) or 'BAD'; tally: dict that collects which engines the stages went through"""
    prog, st, tag = make_case(case)
    try:
        S, F = ipmatrix.IpLQDOCP(), ipmatrix.IpLQDOCPFull()
        df, rf = solve(F, prog, st)
    except ipmatrix.KktError as e:
        return "skip", tag + " the tree engine does not solve it"
    try:
        ds, rs = solve(S, prog, st)
    except ipmatrix.KktError as e:
        if e.code == 1:
            return "skip", tag + " E_SIZES"
        # E_SING: e.g. more final-state rows than the controls of all stages can absorb and a FIXED x_0 - the
        # recursion cannot solve that, the reference's Hqp_IpLQDOCP neither (hqp/Hqp_IpLQDOCP.C:2097-2108); the
        # tree engine factorises the whole system and may.  Not a difference if the reference fails as well.
        ref_fails = False
        try:
            from oracle import refapi
            if refapi.available():
                L = refapi.RefIpMatrix("LQDOCP")
                L.init(prog)
                L.factor(st[0], st[1])
                _ls, lres = L.solve(*st)
                ref_fails = not (lres <= 1e-8)
        except Exception:
            ref_fails = True
        if ref_fails:
            return "skip", tag + " the reference fails as well"
        return "BAD", f"{tag} raised {e} where the reference solves"
    if tally is not None:
        u, f = S.debug(28)
        u0, f0 = S.debug(32)[:2]
        for key, val in (("blocked", u), ("fell", f), ("x0_ran", u0), ("x0_fell", f0)):
            tally[key] = tally.get(key, 0) + int(val)
    if not (rf <= 1e-10):
        return "skip", tag + " tree engine above 1e-10"
    if rs <= 1e-10 and rel_err(ds, df) <= 1e-8:
        return "ok", tag
    return "BAD", f"{tag} residual {rs:.2e} (tree engine {rf:.2e}), relative difference {rel_err(ds, df):.2e}"


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0 = time.time()
    cnt = {"ok": 0, "skip": 0, "BAD": 0}
    t = {"blocked": 0, "fell": 0, "x0_ran": 0, "x0_fell": 0}
    for case in range(seed0, seed0 + ncases):
        status, detail = check(case, t)
        cnt[status] += 1
        if status == "BAD":
            print(detail, flush=True)
    print(f"fuzz_bigstage: {ncases} cases from {seed0}: {cnt}; stages through the blocked elimination {t['blocked']}, fallen back {t['fell']}; "
          f"free initial states through the blocked inverse {t['x0_ran']}, fallen back {t['x0_fell']}; {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
