"""Checker of the pivot-block kernels on single dense blocks (hqpkkt_debug_factor_block):
reassembles P A P' from L, D, the pivot order, and compares; checks M L = I.
Used by tests/test_gpu_block.py and tools/block_time.py."""
from __future__ import annotations

import ctypes as C

import numpy as np

from hqp_amd import _lib


def make_block(kind, p, seed):
    """Symmetric p x p test blocks.
    'qd'   quasi-definite KKT-like block in the order the symbolic phase produces (negative definite part
           first, then the multipliers): no interchange needed;
    'spd'  diagonally dominant, negative definite;
    'indef' random indefinite with small diagonals: interchanges and 2x2 pivots;
    'kkt0' [[-Q, A'], [A, 0]] with the multipliers FIRST: 2x2 pivots throughout;
    'tiny' like qd with a few exactly zero and tiny diagonals (perturbation path)."""
    rng = np.random.default_rng(seed)
    if kind == "spd":
        B = rng.uniform(-0.5, 0.5, (p, p))
        A = -(B + B.T) / 2
        A -= np.diag(np.abs(A).sum(1) + 1.0)
        return A
    if kind in ("qd", "tiny"):
        n1 = max(1, (2 * p) // 3)
        B = rng.uniform(-0.5, 0.5, (n1, n1))
        Q = (B + B.T) / 2 + np.diag(np.abs(B).sum(1) + 1.0)
        Aeq = rng.uniform(-1, 1, (p - n1, n1))
        A = np.zeros((p, p))
        A[:n1, :n1] = -Q
        A[n1:, :n1] = Aeq
        A[:n1, n1:] = Aeq.T
        if kind == "tiny" and p > 4:
            A[1, 1] = 0.0
            A[3, 3] = 1e-30
        return A
    if kind == "indef":
        B = rng.uniform(-1, 1, (p, p))
        A = (B + B.T) / 2
        A[np.diag_indices(p)] *= 0.05
        return A
    if kind == "kkt0":
        me = p // 3
        n1 = p - me
        B = rng.uniform(-0.5, 0.5, (n1, n1))
        Q = (B + B.T) / 2 + np.diag(np.abs(B).sum(1) + 1.0)
        Aeq = rng.uniform(-1, 1, (me, n1))
        A = np.zeros((p, p))
        A[:me, me:] = Aeq
        A[me:, :me] = Aeq.T
        A[me:, me:] = -Q
        return A
    raise ValueError(kind)


def factor_block(A, variant=0, tol=1.0, pivot_eps=1e-20, reps=1, device=0):
    L_ = _lib.lib()
    p = A.shape[0]
    A = np.ascontiguousarray(A, dtype=np.float64)
    Lo = np.zeros(p * p)
    W = np.zeros(p * p)
    dinv = np.zeros(2 * p)
    pt = np.zeros(p, dtype=np.int32)
    lp = np.zeros(p, dtype=np.int32)
    cnt = np.zeros(128, dtype=np.int32)
    ms = C.c_double(0.0)
    rc = L_.hqpkkt_debug_factor_block(device, p, A.ctypes.data, tol, pivot_eps, variant, reps, Lo.ctypes.data,
                                      dinv.ctypes.data, pt.ctypes.data, lp.ctypes.data, W.ctypes.data,
                                      cnt.ctypes.data, C.cast(C.byref(ms), C.c_void_p))
    if rc:
        raise RuntimeError(f"hqpkkt_debug_factor_block: {rc} {_lib.strerror(rc)}")
    return dict(L=Lo.reshape(p, p).T.copy(), W=W.reshape(p, p).T.copy(), dinv=dinv, ptype=pt, lperm=lp,
                counters=cnt, ms=ms.value)


def check_block(A, out):
    """Returns (relative error of P A P' = L D L', max |tril(W) L - I|, is-permutation, counters)."""
    p = A.shape[0]
    L = np.tril(out["L"], -1) + np.eye(p)
    D = np.zeros((p, p))
    k = 0
    pt, dv = out["ptype"], out["dinv"]
    ok = True
    while k < p:
        if pt[k] == 0:
            D[k, k] = 1.0 / dv[2 * k] if dv[2 * k] != 0 else np.inf
            k += 1
        elif pt[k] == 1 and k + 1 < p and pt[k + 1] == 2:
            Di = np.array([[dv[2 * k], dv[2 * k + 1]], [dv[2 * k + 1], dv[2 * k + 2]]])
            D[k:k + 2, k:k + 2] = np.linalg.inv(Di)
            ok = ok and L[k + 1, k] == 0.0
            k += 2
        else:
            ok = False
            k += 1
    lp = out["lperm"]
    perm_ok = sorted(lp.tolist()) == list(range(p))
    if not perm_ok:
        return np.inf, np.inf, False, out["counters"]
    PAP = A[np.ix_(lp, lp)]
    err = np.abs(L @ D @ L.T - PAP).max() / max(np.abs(A).max(), 1e-300)
    W = out["W"]
    blockdiag_zero = all(not np.triu(W[kb:kb + 16, kb:kb + 16], 1).any() for kb in range(0, p, 16))
    inv = np.abs(np.tril(W) @ L - np.eye(p)).max()
    growth = np.abs(L).max()
    return err, inv, ok and perm_ok and blockdiag_zero, growth
