"""Generates tests/golden_lqdocp/*.npz from the REFERENCE's own Hqp_IpLQDOCP (oracle/_ref/libhqpref.so, compiled from
/root/reference by oracle/Makefile).  Run in the build container only:

    python tests/golden_lqdocp/make_golden.py

Each fixture holds the inputs (multistage QP in CSR, z, w, r1..r4) and what the reference's extended Riccati recursion
returned: step() result, solve() result and residual, residuum() of the step.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from hqp_amd import problems  # noqa: E402
from oracle import refapi  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def _sweep(case):
    import fuzz_staged
    prog, st, _tag = fuzz_staged.make_case(case)
    return prog, st


CASES = {
    "plain_K8_nx12_nu3": lambda: (problems.lq_docp(8, 12, 3, seed=4), 1, 0.0),
    "final_eq_K10_nx9_nu2": lambda: (problems.lq_docp(10, 9, 2, seed=5, final_eq=5), 2, 1.0),
    "path_eq_bounds_K12_nx10_nu4": lambda: (problems.lq_docp(12, 10, 4, seed=6, path_eq=2, path_eq_every=2, x_bounds=6), 3, 2.0),
    "free_x0_K9_nx8_nu2": lambda: (problems.lq_docp(9, 8, 2, seed=7, x0_fixed=False, final_eq=3), 4, 1.0),
    "did_K50": lambda: (problems.did_like_qp(50), 5, 0.0),
    "wide_K3_nx130_nu4": lambda: (problems.lq_docp(3, 130, 4, seed=8), 6, 1.0),
    "stiff_sweep672": lambda: _sweep(672),
}


def main():
    assert refapi.available(), refapi.load_error()
    for name, mk in CASES.items():
        got = mk()
        if len(got) == 3:
            prog, seed, spread = got
            st = problems.ip_state(prog, seed, spread)
        else:
            prog, st = got
        z, w, r1, r2, r3, r4 = st
        out = dict(n=prog.n, me=prog.me, m=prog.m, z=z, w=w, r1=r1, r2=r2, r3=r3, r4=r4)
        for blk, (p, i, x) in zip("QAC", (prog.Q, prog.A, prog.C)):
            out[f"{blk}p"], out[f"{blk}i"], out[f"{blk}x"] = p, i, x
        R = refapi.RefIpMatrix("LQDOCP")
        R.init(prog)
        R.factor(z, w)
        stp = R.step(z, w, r1, r2, r3, r4)
        so, res = R.solve(z, w, r1, r2, r3, r4)
        for nm, a, b in zip(("dx", "dy", "dz", "dw"), stp, so):
            out[f"LQDOCP_step_{nm}"] = a
            out[f"LQDOCP_solve_{nm}"] = b
        out["LQDOCP_res"] = res
        out["LQDOCP_res_of_step"] = R.residuum(z, w, r1, r2, r3, r4, *stp)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, prog.dims, "res", res, "res of step", out["LQDOCP_res_of_step"])


if __name__ == "__main__":
    main()
