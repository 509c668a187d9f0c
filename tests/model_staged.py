"""TEST HELPER: numpy model of the STAGED engine (HQPKKT_MODE_STAGED).

The stage-structured solution of the interior-point Newton system that the
reference's Hqp_IpLQDOCP computes by its extended Riccati recursion
(hqp/Hqp_IpLQDOCP.C:796-976, ExRiccatiFactorSc :1794-1999, ExRiccatiSolveSc
:2007-2182), restated the way the HIP kernels of hqp_amd/csrc/staged.hip.h do it:

  reduced system   H s + A' l = g,  A s = -r2          (s = -dx, l = dy)
                   H = Q + C' (Z/W) C,  g = r1 - C' ((r4 + z r3) / w)
  stage k          s_k = (x_k, u_k),  x_{k+1} = F_k s_k + f_k,  E_k s_k + a_k = 0
  cost-to-go       J_k(x) = 1/2 x' V_k x + v_k' x   with carried constraints B_k x + beta_k = 0

  backward  G = H_k + F' V_{k+1} F,  N = [E_k ; B_{k+1} F]  (rows: own, then carried)
            rank-revealing elimination of N_u (complete pivoting): consumed rows R, leftover
            rows L with  N_L - t N_R  free of u  ->  B_k = (N_L - t N_R)_x
            K = [G_uu N_uR' ; N_uR 0],  Y = [G_ux ; N_xR],  Rm = K^-1 Y,  V_k = G_xx - Y' Rm
  forward   [u ; yhat] = -(Rm x + K^-1 y0),  multipliers of N's rows: consumed yhat - t' eta,
            leftover eta;  x_{k+1} = F s + f;  p_k = V x + v + B' eta

Where the reference uses a QR factorisation with column pivoting for the range /
null space split (GE_QP, meschach/addon_hqp.c:399-475) and a dense Bunch-Kaufman
factorisation of Z' G_uu Z, this model (like the kernels) uses Gaussian elimination with
complete pivoting for both; the results agree to rounding (tests/test_staged_model.py).
Dense storage, test sizes only.  Never imported by the product.
"""
from __future__ import annotations

import numpy as np

GE_TOL = 1e-6  # _ge_tol of the reference (hqp/Hqp_IpLQDOCP.C:113)


class StageError(ValueError):
    pass


def stage_structure(n, me, m, A, Q=None, C=None):
    """Stage dimensions from the -1.0 staircase of A, as Hqp_IpLQDOCP::Get_Dim
    (hqp/Hqp_IpLQDOCP.C:201-287), Get_Constr_Dim (:368-407) and Check_Structure
    (:298-354) find them.  A, Q, C are (indptr, indices, data) triples."""
    Ap, Ai, Ax = A
    if me == 0:
        raise StageError("no equalities")
    nk, nmk = [0], [0]  # states per stage, first column of the stage
    cur, last = 0, -1
    ndyn = None
    for i in range(me):
        ln = Ap[i + 1] - Ap[i]
        if ln <= 1 or Ax[Ap[i + 1] - 1] != -1.0:
            raise StageError(f"row {i}: no -1.0 at the end")
        icl, icl1 = int(Ai[Ap[i + 1] - 1]), int(Ai[Ap[i + 1] - 2])
        if icl <= last:
            raise StageError("staircase not increasing")
        if icl - last > 1 or icl - icl1 < cur:  # a new stage starts with this row
            if len(nk) > 1:
                nk[-1] = cur
            nk.append(0)
            nmk.append(icl)
            cur = 1
        else:
            cur += 1
        last = icl
        if icl == n - 1:
            nk[-1] = cur
            ndyn = i + 1
            break
    if ndyn is None or len(nk) < 2:
        raise StageError("staircase does not reach the last column")
    K = len(nk) - 1
    nk[0] = min(nk[1], nmk[1])
    mk = [nmk[k + 1] - nmk[k] - nk[k] for k in range(K)]
    if min(mk) < 0:
        raise StageError("negative number of controls")
    stage_of = np.zeros(n, dtype=np.int64)
    for k in range(K):
        stage_of[nmk[k]:nmk[k + 1]] = k
    stage_of[nmk[K]:] = K
    nks = np.concatenate([[0], np.cumsum(nk[1:])]).astype(np.int64)  # first dynamics row of stage k
    # dynamics rows: first and second last entry in stage k, last one in stage k+1
    for k in range(K):
        for i in range(nks[k], nks[k + 1]):
            c = Ai[Ap[i]:Ap[i + 1]]
            if stage_of[c[0]] != k or stage_of[c[-2]] != k or stage_of[c[-1]] != k + 1:
                raise StageError(f"dynamics row {i} leaves stage {k}")
    eq_rows = [[] for _ in range(K + 1)]
    for i in range(ndyn, me):
        c = Ai[Ap[i]:Ap[i + 1]]
        if len(c) == 0:
            raise StageError("empty equality row")
        k = int(stage_of[c[0]])
        if stage_of[c[-1]] != k:
            raise StageError(f"equality row {i} spans stages")
        eq_rows[k].append(i)
    in_rows = [[] for _ in range(K + 1)]
    if C is not None:
        Cp, Ci, _ = C
        for i in range(m):
            c = Ci[Cp[i]:Cp[i + 1]]
            if len(c) == 0:
                raise StageError("empty inequality row")
            k = int(stage_of[c[0]])
            if stage_of[c[-1]] != k:
                raise StageError(f"inequality row {i} spans stages")
            in_rows[k].append(i)
    if Q is not None:
        Qp, Qi, _ = Q
        for i in range(n):
            c = Qi[Qp[i]:Qp[i + 1]]
            if len(c) and (stage_of[c[0]] != stage_of[i] or stage_of[c[-1]] != stage_of[i]):
                raise StageError(f"Q row {i} spans stages")
    # fixed initial state: every x_0 component has a singleton row among the stage-0 equalities
    fix = {}
    for i in eq_rows[0]:
        if Ap[i + 1] - Ap[i] == 1 and Ai[Ap[i]] < nk[0] and Ax[Ap[i]] != 0.0 and int(Ai[Ap[i]]) not in fix:
            fix[int(Ai[Ap[i]])] = i
    fixed_x0 = len(fix) == nk[0] and nk[0] > 0
    fix_rows = [fix[j] for j in range(nk[0])] if fixed_x0 else []
    if fixed_x0:
        eq_rows[0] = [i for i in eq_rows[0] if i not in set(fix_rows)]
    return dict(K=K, nk=list(map(int, nk)), mk=list(map(int, mk)), nmk=list(map(int, nmk)),
                nks=nks, ndyn=int(ndyn), eq_rows=eq_rows, in_rows=in_rows, fixed_x0=fixed_x0,
                fix_rows=fix_rows)


def ge_complete(M, ncols, tol, jordan=False):
    """Gaussian elimination with complete pivoting over the first ncols columns of M
    (rows x (ncols + extra)); returns (pivot rows, pivot cols, M reduced).  Stops when the
    largest remaining entry is <= tol.  With jordan the pivot column is cleared in all
    other rows."""
    M = M.copy()
    rows, _ = M.shape
    free = np.ones(rows, dtype=bool)
    cfree = np.ones(ncols, dtype=bool)
    pr, pc = [], []
    for _ in range(min(rows, ncols)):
        sub = np.abs(M[:, :ncols]) * free[:, None] * cfree[None, :]
        i, j = np.unravel_index(np.argmax(sub), sub.shape)
        if not sub[i, j] > tol:
            break
        free[i], cfree[j] = False, False
        pr.append(int(i)), pc.append(int(j))
        piv = M[i, j]
        for r in range(rows):
            if r != i and (jordan or free[r]):
                f = M[r, j] / piv
                if f != 0.0:
                    M[r, :] -= f * M[i, :]
                    M[r, j] = 0.0
    return pr, pc, M


def inv_complete(Kmat):
    """Inverse by Gauss-Jordan with complete pivoting and the symmetric scaling of the
    kernels (u part: 1/sqrt(K_ii) where K_ii > 1, hqp/Hqp_IpLQDOCP.C:1851-1858; constraint
    rows: their largest entry).  Raises on a singular matrix."""
    q = Kmat.shape[0]
    if q == 0:
        return np.zeros((0, 0))
    return np.linalg.inv(Kmat)


class ScaledSolve:
    """[V_0 B_0'; B_0 0] of a free initial state: factors of the diagonally scaled matrix applied by substitution (the
    kernels: LU with complete pivoting, k_st_init_factor / k_st_x0_free; the reference: BKPfactor of the same scaled
    matrix, hqp/Hqp_IpLQDOCP.C:1984-1996).  A product with an explicit inverse is not backward stable and loses the
    solution when V_0 is ill-conditioned (unstable closed loops)."""

    def __init__(self, K0, n0):
        import scipy.linalg as sla
        d = np.ones(K0.shape[0])
        dg = np.diag(K0)[:n0]
        d[:n0] = np.where(dg > 1.0, 1.0 / np.sqrt(np.where(dg > 1.0, dg, 1.0)), 1.0)
        self.d, self.lu, self._solve = d, sla.lu_factor(K0 * d[:, None] * d[None, :]), sla.lu_solve

    def __matmul__(self, b):
        return self.d * self._solve(self.lu, self.d * b)


class StagedModel:
    def __init__(self, prog):
        self.prog = prog
        n, me, m = prog.dims
        self.S = stage_structure(n, me, m, prog.A, prog.Q, prog.C)
        from model import dense_blocks
        self.Q, self.A, self.C = dense_blocks(prog)

    # ------------------------------------------------------------------ factor
    def factor(self, z, w):
        S, Q, A, C = self.S, self.Q, self.A, self.C
        K, nk, mk, nmk, nks = S["K"], S["nk"], S["mk"], S["nmk"], S["nks"]
        self.z, self.w = z.copy(), w.copy()
        H = Q + C.T @ np.diag(z / w) @ C if len(z) else Q.copy()
        self.H = H
        self.F, self.E = [], []
        for k in range(K):
            c0, c1 = nmk[k], nmk[k + 1]
            self.F.append(A[nks[k]:nks[k + 1], c0:c1].copy())
        for k in range(K + 1):
            c0 = nmk[k]
            c1 = nmk[k + 1] if k < K else len(Q)
            self.E.append(A[S["eq_rows"][k], c0:c1].copy() if S["eq_rows"][k] else np.zeros((0, c1 - c0)))
        V = [None] * (K + 1)
        B = [None] * (K + 1)
        c0 = nmk[K]
        V[K] = H[c0:, c0:].copy()
        B[K] = self.E[K].copy()
        st = [None] * K
        for k in range(K - 1, -1, -1):
            n, mu = nk[k], mk[k]
            c0, c1 = nmk[k], nmk[k + 1]
            F = self.F[k]
            G = H[c0:c1, c0:c1] + F.T @ V[k + 1] @ F
            G = 0.5 * (G + G.T)
            N = np.vstack([self.E[k], B[k + 1] @ F])
            c = N.shape[0]
            Nx, Nu = N[:, :n], N[:, n:]
            aug = np.hstack([Nu, np.eye(c)])
            pr, pc, red = ge_complete(aug, mu, GE_TOL)
            R = pr
            L = [i for i in range(c) if i not in set(R)]
            coef = red[:, mu:]  # reduced row i = sum_j coef[i, j] * original row j
            # leftover rows in terms of the originals: N_L - t N_R
            t = -coef[np.ix_(L, R)] if L and R else np.zeros((len(L), len(R)))
            r = len(R)
            Kmat = np.zeros((mu + r, mu + r))
            Kmat[:mu, :mu] = G[n:, n:]
            Kmat[mu:, :mu] = Nu[R, :]
            Kmat[:mu, mu:] = Nu[R, :].T
            Y = np.vstack([G[n:, :n], Nx[R, :]])
            Kinv = inv_complete(Kmat)
            Rm = Kinv @ Y
            Rm = Rm + Kinv @ (Y - Kmat @ Rm)  # one round of refinement, as the kernels do
            Vk = G[:n, :n] - Y.T @ Rm
            V[k] = 0.5 * (Vk + Vk.T)
            B[k] = Nx[L, :] - t @ Nx[R, :]
            st[k] = dict(R=R, L=L, t=t, Kinv=Kinv, Kmat=Kmat, Y=Y, Rm=Rm, c=c)
        self.V, self.B, self.st = V, B, st
        if S["fixed_x0"]:
            if B[0].shape[0] and np.abs(B[0]).max() > GE_TOL:
                raise np.linalg.LinAlgError("constraints left on a fixed initial state")
            self.K0inv = None
        else:
            n0, c0n = nk[0], B[0].shape[0]
            K0 = np.zeros((n0 + c0n, n0 + c0n))
            K0[:n0, :n0] = V[0]
            K0[n0:, :n0] = B[0]
            K0[:n0, n0:] = B[0].T
            self.K0 = K0
            self.K0inv = ScaledSolve(K0, n0)

    # -------------------------------------------------------------------- step
    def step(self, r1, r2, r3, r4):
        S, A, C = self.S, self.A, self.C
        K, nk, mk, nmk, nks = S["K"], S["nk"], S["mk"], S["nmk"], S["nks"]
        z, w = self.z, self.w
        nvar = len(r1)
        g = r1 - (C.T @ ((r4 + z * r3) / w) if len(z) else 0.0)
        q = -g
        V, B, st = self.V, self.B, self.st
        v = [None] * (K + 1)
        beta = [None] * (K + 1)
        rho = [None] * K
        c0 = nmk[K]
        v[K] = q[c0:].copy()
        beta[K] = r2[S["eq_rows"][K]].copy() if S["eq_rows"][K] else np.zeros(0)
        for k in range(K - 1, -1, -1):
            n, mu = nk[k], mk[k]
            c0, c1 = nmk[k], nmk[k + 1]
            F = self.F[k]
            f = r2[nks[k]:nks[k + 1]]
            a = r2[S["eq_rows"][k]] if S["eq_rows"][k] else np.zeros(0)
            tt = v[k + 1] + V[k + 1] @ f
            gam = q[c0:c1] + F.T @ tt
            nu = np.concatenate([a, beta[k + 1] + B[k + 1] @ f])
            s = st[k]
            y0 = np.concatenate([gam[n:], nu[s["R"]]])
            rho[k] = s["Kinv"] @ y0
            rho[k] = rho[k] + s["Kinv"] @ (y0 - s["Kmat"] @ rho[k])
            v[k] = gam[:n] - s["Y"].T @ rho[k]
            beta[k] = nu[s["L"]] - s["t"] @ nu[s["R"]]
        sx = np.zeros(nvar)
        lam = np.zeros(len(r2))
        n0 = nk[0]
        if S["fixed_x0"]:
            fr = np.array(S["fix_rows"])
            val = np.array([A[i, j] for j, i in enumerate(fr)])
            x = -r2[fr] / val
            eta = np.zeros(B[0].shape[0])
            lam[fr] = -(V[0] @ x + v[0] + B[0].T @ eta) / val
        else:
            b0 = np.concatenate([v[0], beta[0]])
            sol = self.K0inv @ b0
            sol = -(sol + self.K0inv @ (b0 - self.K0 @ sol))
            x, eta = sol[:n0], sol[n0:]
        for k in range(K):
            n, mu = nk[k], mk[k]
            c0, c1 = nmk[k], nmk[k + 1]
            s = st[k]
            uy = -(s["Rm"] @ x + rho[k])
            u, yhat = uy[:mu], uy[mu:]
            yN = np.zeros(s["c"])
            yN[s["R"]] = yhat - s["t"].T @ eta
            yN[s["L"]] = eta
            e = len(S["eq_rows"][k])
            if e:
                lam[S["eq_rows"][k]] = yN[:e]
            eta = yN[e:]
            sk = np.concatenate([x, u])
            sx[c0:c1] = sk
            x = self.F[k] @ sk + r2[nks[k]:nks[k + 1]]
            lam[nks[k]:nks[k + 1]] = V[k + 1] @ x + v[k + 1] + B[k + 1].T @ eta
        sx[nmk[K]:] = x
        if S["eq_rows"][K]:
            lam[S["eq_rows"][K]] = eta
        dx = -sx
        dy = lam
        dw = C @ dx - r3 if len(z) else np.zeros(0)
        dz = (r4 - z * dw) / w if len(z) else np.zeros(0)
        return dx, dy, dz, dw


def kkt_residual(prog, z, w, r, d):
    """max inf-norm of the four block residuals (hqp/Hqp_IpMatrix.C:131-178)"""
    from model import dense_blocks
    Q, A, C = dense_blocks(prog)
    r1, r2, r3, r4 = r
    dx, dy, dz, dw = d
    res = [r1 + Q @ dx - A.T @ dy - (C.T @ dz if len(z) else 0.0), r2 - A @ dx]
    if len(z):
        res += [r3 - C @ dx + dw, r4 - (z * dw + w * dz)]
    return max(float(np.abs(x).max()) if len(x) else 0.0 for x in res)
