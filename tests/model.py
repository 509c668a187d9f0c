"""TEST HELPER: numpy model of the supernodal LDL' the HIP kernels implement.

It consumes the symbolic structure exported by ``hqpkkt_debug_get`` (so the
host-side analysis is what gets checked) and mirrors the numeric rules of the
kernels in hqp_amd/csrc/kernels.hip.h: Bunch-Kaufman pivoting restricted to each
supernode's pivot block, static perturbation of tiny pivots, right-looking Schur
updates restricted to the symbolic border.  Dense O(dim^2) storage: test sizes
only.  Never imported by the product.
"""
from __future__ import annotations

import numpy as np

ALPHA0 = 0.6403882032022076


def dense_blocks(prog):
    n, me, m = prog.dims
    Q = np.zeros((n, n))
    p, i, x = prog.Q
    for r in range(n):
        for k in range(p[r], p[r + 1]):
            if i[k] >= r:
                Q[r, i[k]] = x[k]
                Q[i[k], r] = x[k]
    A = np.zeros((me, n))
    p, i, x = prog.A
    for r in range(me):
        A[r, i[p[r]:p[r + 1]]] = x[p[r]:p[r + 1]]
    Cm = np.zeros((m, n))
    p, i, x = prog.C
    for r in range(m):
        Cm[r, i[p[r]:p[r + 1]]] = x[p[r]:p[r + 1]]
    return Q, A, Cm


def scaled_kkt(prog, z, w, mode):
    """Dense scaled matrix in QP numbering + scale vector (reference semantics:
    hqp/Hqp_IpSpBKP.C:139-176, hqp/Hqp_IpRedSpBKP.C:281-316)."""
    n, me, m = prog.dims
    Q, A, Cm = dense_blocks(prog)
    if mode == 0:
        dim = n + me + m
        K = np.zeros((dim, dim))
        K[:n, :n] = -Q
        K[n:n + me, :n] = A
        K[:n, n:n + me] = A.T
        K[n + me:, :n] = Cm
        K[:n, n + me:] = Cm.T
        wz = w / z
        K[n + me:, n + me:] = np.diag(wz)
        sc = np.ones(dim)
        sc[n + me:] = np.minimum(1.0, np.sqrt(1.0 / wz))
    else:
        dim = n + me
        K = np.zeros((dim, dim))
        K[:n, :n] = -(Q + Cm.T @ np.diag(z / w) @ Cm)
        K[n:, :n] = A
        K[:n, n:] = A.T
        sc = np.ones(dim)
        sc[:n] = np.minimum(1.0, np.sqrt(-1.0 / np.diag(K)[:n]))
    return K * sc[:, None] * sc[None, :], sc


def bk_block(Ain, alpha, pert, signs):
    """Bunch-Kaufman LDL' of a dense symmetric block, mirroring k_factor_diag.
    Returns (L unit lower, list of pivot blocks (start, size, Dinv), perm) with
    A[perm][:, perm] = L D L'."""
    a = np.array(Ain, dtype=float)
    p = a.shape[0]
    a = np.tril(a)
    lp = np.arange(p)
    sg = np.array(signs, dtype=float)
    blocks = []
    n2, npert = 0, 0

    def sym(i, j):
        return a[i, j] if i >= j else a[j, i]

    k = 0
    while k < p:
        col = np.abs(a[k + 1:, k])
        if col.size:
            r = k + 1 + int(np.argmax(col))  # first max
            lam = col[r - k - 1]
        else:
            r, lam = p, 0.0
        akk = abs(a[k, k])
        kind = 0
        if not (akk >= alpha * lam):
            sigma = max(abs(sym(r, t)) for t in range(k, p) if t != r)
            if sigma * akk >= alpha * lam * lam:
                kind = 0
            elif abs(a[r, r]) >= alpha * sigma:
                kind = 1
            else:
                kind = 2
        p1 = k + 1 if kind == 2 else k
        if kind != 0 and r != p1:
            # symmetric interchange p1 <-> r on the lower triangle
            full = a + np.tril(a, -1).T
            idx = np.arange(p)
            idx[p1], idx[r] = r, p1
            full = full[np.ix_(idx, idx)]
            a = np.tril(full)
            lp[p1], lp[r] = lp[r], lp[p1]
        if kind != 2:
            d = a[k, k]
            if not (abs(d) >= pert) or d == 0.0:
                d = sg[lp[k]] * max(pert, 1e-300)
                npert += 1
            a[k, k] = d
            c = a[k + 1:, k].copy()
            l = c / d
            a[k + 1:, k] = l
            a[k + 1:, k + 1:] -= np.tril(np.outer(l, c))
            blocks.append((k, 1, np.array([[1.0 / d]])))
            k += 1
        else:
            D = np.array([[a[k, k], a[k + 1, k]], [a[k + 1, k], a[k + 1, k + 1]]])
            det = D[0, 0] * D[1, 1] - D[0, 1] ** 2
            if not (abs(det) >= pert * pert) or det == 0.0:
                D = np.diag([sg[lp[k]] * max(pert, 1e-300), sg[lp[k + 1]] * max(pert, 1e-300)])
                det = D[0, 0] * D[1, 1]
                a[k, k], a[k + 1, k], a[k + 1, k + 1] = D[0, 0], 0.0, D[1, 1]
                npert += 2
            Dinv = np.array([[D[1, 1], -D[0, 1]], [-D[0, 1], D[0, 0]]]) / det
            c = a[k + 2:, k:k + 2].copy()
            l = c @ Dinv
            a[k + 2:, k:k + 2] = l
            a[k + 2:, k + 2:] -= np.tril(l @ c.T)
            blocks.append((k, 2, Dinv))
            n2 += 1
            k += 2
    L = np.tril(a, -1) + np.eye(p)
    for (s, sz, _) in blocks:
        if sz == 2:
            L[s + 1, s] = 0.0
    return L, blocks, lp, n2, npert


def apply_dinv(blocks, Y):
    out = np.array(Y, dtype=float)
    for (s, sz, Dinv) in blocks:
        out[s:s + sz] = Dinv @ Y[s:s + sz]
    return out


class Model:
    def __init__(self, struct, tol=1.0, pivot_eps=1e-10):
        self.s = struct
        self.alpha = tol * ALPHA0
        self.pivot_eps = pivot_eps

    def nodes(self):
        s = self.s
        for k in range(len(s["npiv"])):
            P = np.arange(s["piv_start"][k], s["piv_start"][k] + s["npiv"][k])
            B = s["border_idx"][s["border_ptr"][k]:s["border_ptr"][k + 1]]
            yield k, P, np.asarray(B, dtype=int)

    def begin(self, Kq, n):
        """Kq: dense scaled matrix in QP numbering; n: number of x variables."""
        s = self.s
        e = np.asarray(s["elim"], dtype=int)
        dim = Kq.shape[0]
        self.S = np.zeros((dim, dim))
        self.S[np.ix_(e, e)] = Kq
        self.signs = np.ones(dim)
        self.signs[e[:n]] = -1.0
        self.pert = self.pivot_eps * np.abs(Kq).max()
        self.fac = {}
        self.n2 = self.npert = 0
        self.struct_violation = 0.0
        self.done = np.zeros(dim, dtype=bool)

    def eliminate(self, ids=None):
        """Eliminate the supernodes ``ids`` (ascending = children first; default all)."""
        S, dim = self.S, self.S.shape[0]
        want = None if ids is None else set(int(i) for i in ids)
        for k, P, B in self.nodes():
            if want is not None and k not in want:
                continue
            rest = np.ones(dim, dtype=bool)
            rest[P] = False
            rest[B] = False
            rest[self.done] = False
            # everything coupled to the pivots must be inside the symbolic front
            if rest.any():
                self.struct_violation = max(self.struct_violation, np.abs(S[np.ix_(rest, P)]).max())
            A11 = S[np.ix_(P, P)]
            L, blocks, lp, n2, npert = bk_block(A11, self.alpha, self.pert, self.signs[P])
            self.n2 += n2
            self.npert += npert
            A21 = S[np.ix_(B, P)][:, lp]
            X = np.linalg.solve(L, A21.T).T if B.size else np.zeros((0, P.size))
            L21 = apply_dinv(blocks, X.T).T if B.size else X
            if B.size:
                S[np.ix_(B, B)] -= L21 @ X.T
            self.fac[k] = (P, B, L, blocks, lp, L21)
            self.done[P] = True

    def factor(self, Kq, n):
        self.begin(Kq, n)
        self.eliminate()

    def forward(self, x, ids=None):
        for k in sorted(self.fac if ids is None else ids):
            P, B, L, blocks, lp, L21 = self.fac[int(k)]
            y = np.linalg.solve(L, x[P][lp])
            if B.size:
                x[B] -= L21 @ y
            x[P] = apply_dinv(blocks, y)
        return x

    def backward(self, x, ids=None):
        for k in sorted(self.fac if ids is None else ids, reverse=True):
            P, B, L, blocks, lp, L21 = self.fac[int(k)]
            v = x[P] - (L21.T @ x[B] if B.size else 0.0)
            xp = np.linalg.solve(L.T, v)
            out = np.zeros(P.size)
            out[lp] = xp
            x[P] = out
        return x

    def solve(self, rhs_e):
        x = np.array(rhs_e, dtype=float)
        return self.backward(self.forward(x))
