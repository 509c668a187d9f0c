"""Generates tests/golden_full_size/*.npz from the REFERENCE's own Hqp_IpSpBKP / Hqp_IpRedSpBKP (oracle/_ref/libhqpref.so,
compiled from /root/reference by oracle/Makefile) on BASELINE.json's configs[1] at FULL size: the banded QP with n = 40 000,
band 80 - KKT dimension 10^5, mat_sbw 200 (1.3 s per factor + solve on one core).  Run in the build container only:

    python tests/golden_full_size/make_golden.py

The inputs are regenerated from the seeds (problems.banded_qp / ip_state; a checksum guards the generator); of the
reference's solve() result a fixture keeps every 37th component of dx, dy, dz, dw, the four infinity norms, the four
sums and the residual - 90 KB instead of 2.4 MB.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hqp_amd import problems  # noqa: E402
from oracle import refapi  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
STRIDE = 37
# name: (n, band, seed of the QP, seed of the interior-point state, w / z spread in decades, plugin)
#   or: ("mesh", cells per side, seed of the QP, seed of the state, spread, plugin) - problems.grid_sparse_qp, the stand-in of
#   BASELINE.json's configs[4] at 10^5 variables (the reference: RCM band 631, 51 s; ours: the dissection of the KKT graph)
#   or: ("cute", n, seed of the QP, seed of the state, spread, plugin) - problems.cute_like_qp
CASES = {
    "c2_banded_n40000_b80_SpBKP": (40000, 80, 12345, 1, 0.0, "SpBKP"),
    "c2_banded_n40000_b80_RedSpBKP_spread": (40000, 80, 12345, 2, 2.0, "RedSpBKP"),
    "mesh_316x316_RedSpBKP": ("mesh", 316, 5, 1, 1.0, "RedSpBKP"),
    # SURVEY.md 8(d) C5's row density (problems.cute_like_qp: 10 ... 100 entries per row in a window of 400 columns, 0.1 % of
    # them anywhere) at the largest size the reference finishes in about a minute (65 s): its RCM band is the whole matrix here
    "cute_n6500_RedSpBKP": ("cute", 6500, 17, 1, 1.0, "RedSpBKP"),
}


def inputs(case):
    n, band, seed, sseed, spread, _kind = case
    prog = problems.grid_sparse_qp(band, band, seed=seed) if n == "mesh" else \
        problems.cute_like_qp(band, seed=seed) if n == "cute" else problems.banded_qp(n, band, seed)
    return prog, problems.ip_state(prog, sseed, spread)


def checksum(prog, st):
    return np.array([np.abs(prog.A[2]).sum(), np.abs(prog.Q[2]).sum(), np.abs(st[0]).sum(), np.abs(st[2]).sum()])


def main():
    assert refapi.available(), refapi.load_error()
    for name, case in CASES.items():
        if os.path.exists(os.path.join(HERE, name + ".npz")) and "--all" not in sys.argv:
            continue  # (python make_golden.py --all regenerates everything)
        prog, st = inputs(case)
        R = refapi.RefIpMatrix(case[5])
        R.init(prog)
        R.factor(st[0], st[1])
        sol, res = R.solve(*st)
        out = dict(checksum=checksum(prog, st), res=res, mat_sbw=R.sbw if hasattr(R, "sbw") else -1)
        for nm, v in zip(("dx", "dy", "dz", "dw"), sol):
            out[nm + "_sample"] = v[::STRIDE].copy()
            out[nm + "_norm"] = np.abs(v).max() if len(v) else 0.0
            out[nm + "_sum"] = v.sum()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, prog.dims, "res", res, "seconds", getattr(R, "t_factor", 0) + getattr(R, "t_solve", 0), flush=True)


if __name__ == "__main__":
    main()
