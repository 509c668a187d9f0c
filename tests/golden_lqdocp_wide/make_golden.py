"""Generates tests/golden_lqdocp_wide/*.npz from the REFERENCE's own Hqp_IpLQDOCP (oracle/_ref/libhqpref.so, compiled from
/root/reference by oracle/Makefile) on slices of the headline workload at WIDE stages: K = 2 stages of 1000 states and
K = 3 stages of 2100 states with 50 controls each (problems.c4_docp_csr, the QP family of BASELINE configs[3]; the
reference needs 3 s and 35 s per factor + solve there - the full width of 5000 states takes minutes per stage).  Run in
the build container only:

    python tests/golden_lqdocp_wide/make_golden.py

The dense dynamics blocks are megabytes of random numbers, so a fixture holds the generator's arguments (the inputs are
regenerated from the seeds; a checksum of the values of A, z and r1 guards the generator) and what the reference
returned: step() result, solve() result and residual, residuum() of the step.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hqp_amd import problems  # noqa: E402
from oracle import refapi  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

# name: (K, nx, nu, seed of the QP, seed of the interior-point state, w / z spread in decades[, options of problems.lq_docp])
# (five numbers: problems.c4_docp_csr; with options: problems.lq_docp - a free initial state of 600 components with
#  final-state equalities carried back through the stages, path equalities with state bounds at 800 states, stages of
#  300 controls - the blocked elimination of the control-sized system - with carried final-state rows)
CASES = {
    "c4_K2_nx1000_nu50": (2, 1000, 50, 1, 1, 0.0),
    "c4_K3_nx2100_nu50": (3, 2100, 50, 2, 2, 1.0),
    "c4_K2_nx5000_nu50": (2, 5000, 50, 3, 3, 1.0),  # the headline's FULL stage width: the reference needs 11 minutes
    "free_x0_final_eq_K4_nx600_nu20": (4, 600, 20, 11, 3, 1.0, dict(x0_fixed=False, final_eq=40)),
    "path_eq_bounds_K3_nx800_nu40": (3, 800, 40, 12, 4, 2.0, dict(path_eq=6, path_eq_every=1, x_bounds=100)),
    "many_controls_K3_nx400_nu300": (3, 400, 300, 13, 5, 1.0, dict(final_eq=30)),
}


def inputs(case):
    K, nx, nu, seed, sseed, spread = case[:6]
    if len(case) > 6:
        prog = problems.lq_docp(int(K), int(nx), int(nu), seed=int(seed), **case[6])
    else:
        prog = problems.c4_docp_csr(int(K), int(nx), int(nu), int(seed))
    return prog, problems.ip_state(prog, int(sseed), spread)


def checksum(prog, st):
    return np.array([np.abs(prog.A[2]).sum(), np.abs(st[0]).sum(), np.abs(st[2]).sum()])


def main():
    assert refapi.available(), refapi.load_error()
    for name, case in CASES.items():
        if os.path.exists(os.path.join(HERE, name + ".npz")) and "--all" not in sys.argv:
            continue  # (python make_golden.py --all regenerates everything)
        prog, st = inputs(case)
        z, w, r1, r2, r3, r4 = st
        out = dict(case=np.array(case[:6], dtype=np.float64), checksum=checksum(prog, st))
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
        print(name, prog.dims, "res", res, "res of step", out["LQDOCP_res_of_step"], flush=True)


if __name__ == "__main__":
    main()
