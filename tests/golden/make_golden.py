"""Generates tests/golden/*.npz from the REFERENCE itself (oracle/_ref/libhqpref.so =
the reference's Hqp_IpSpBKP / Hqp_IpRedSpBKP compiled from /root/reference by
oracle/Makefile).  Run in the build container only:

    python tests/golden/make_golden.py

Each fixture holds the inputs (QP blocks in CSR, z, w, r1..r4) and what the
reference returned: mat_sbw, _QP2J, step() result, solve() result and residual.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hqp_amd import problems  # noqa: E402
from oracle import refapi  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    "banded_n60_b4": (lambda: problems.banded_qp(60, 4, 1), 1, 0.0),
    "banded_n300_b10": (lambda: problems.banded_qp(300, 10, 2), 2, 0.0),
    "banded_n800_b10_spread3": (lambda: problems.banded_qp(800, 10, 3), 3, 3.0),
    "did_K50": (lambda: problems.did_like_qp(50), 5, 0.0),
    "did_K50_spread4": (lambda: problems.did_like_qp(50), 6, 4.0),
    "did_K400_spread2": (lambda: problems.did_like_qp(400), 7, 2.0),
    "random_n200": (lambda: problems.random_sparse_qp(200, 60, 150), 3, 2.0),
    "noineq_n120": (lambda: _noineq(), 4, 0.0),
}


def _noineq():
    p = problems.banded_qp(120, 6, 9)
    e = (np.zeros(1, np.int32), np.zeros(0, np.int32), np.zeros(0))
    return problems.Program(p.n, p.me, 0, p.Q, p.A, e)


def main():
    assert refapi.available(), refapi.load_error()
    for name, (mk, seed, spread) in CASES.items():
        prog = mk()
        z, w, r1, r2, r3, r4 = problems.ip_state(prog, seed, spread)
        out = dict(n=prog.n, me=prog.me, m=prog.m, z=z, w=w, r1=r1, r2=r2, r3=r3, r4=r4)
        for blk, (p, i, x) in zip("QAC", (prog.Q, prog.A, prog.C)):
            out[f"{blk}p"], out[f"{blk}i"], out[f"{blk}x"] = p, i, x
        for kind in ("SpBKP", "RedSpBKP"):
            R = refapi.RefIpMatrix(kind)
            R.init(prog)
            R.factor(z, w)
            st = R.step(z, w, r1, r2, r3, r4)
            so, res = R.solve(z, w, r1, r2, r3, r4)
            out[f"{kind}_sbw"] = R.sbw
            out[f"{kind}_perm"] = R.perm()
            out[f"{kind}_pivot"] = R.pivot()
            for nm, a, b in zip(("dx", "dy", "dz", "dw"), st, so):
                out[f"{kind}_step_{nm}"] = a
                out[f"{kind}_solve_{nm}"] = b
            out[f"{kind}_res"] = res
            out[f"{kind}_res_of_step"] = R.residuum(z, w, r1, r2, r3, r4, *st)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, prog.dims, "SpBKP sbw", out["SpBKP_sbw"], "res", out["SpBKP_res"],
              "| RedSpBKP sbw", out["RedSpBKP_sbw"], "res", out["RedSpBKP_res"])


if __name__ == "__main__":
    main()
