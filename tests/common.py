"""Shared helpers for the test-suite (test infrastructure)."""
import glob
import os

import numpy as np

from hqp_amd import problems

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLDEN = sorted(os.path.splitext(os.path.basename(f))[0] for f in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
KINDS = ("SpBKP", "RedSpBKP")
# multistage QPs with the results of the reference's own Hqp_IpLQDOCP (tests/golden_lqdocp/make_golden.py)
GOLDEN_LQDOCP_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_lqdocp")
GOLDEN_LQDOCP = sorted(os.path.splitext(os.path.basename(f))[0] for f in glob.glob(os.path.join(GOLDEN_LQDOCP_DIR, "*.npz")))


def load_golden(name, directory=None):
    g = dict(np.load(os.path.join(directory or GOLDEN_DIR, name + ".npz")))
    n, me, m = int(g["n"]), int(g["me"]), int(g["m"])
    prog = problems.Program(n, me, m, (g["Qp"], g["Qi"], g["Qx"]), (g["Ap"], g["Ai"], g["Ax"]),
                            (g["Cp"], g["Ci"], g["Cx"]))
    state = tuple(g[k] for k in ("z", "w", "r1", "r2", "r3", "r4"))
    return prog, state, g


def rel_err(a, b):
    """max over the four blocks of ||a-b||inf / ||b||inf"""
    worst = 0.0
    for x, y in zip(a, b):
        if len(y) == 0:
            continue
        worst = max(worst, float(np.abs(np.asarray(x) - np.asarray(y)).max() / max(np.abs(y).max(), 1e-300)))
    return worst


def new_d(prog):
    return [np.zeros(k) for k in (prog.n, prog.me, prog.m, prog.m)]
