"""Synthetic Hqp_Program generators for parity tests and bench.py.

These are OUR workload generators; they only produce the
QP blocks an ``Hqp_Program`` carries (hqp/Hqp_Program.h:33-65):

    min 1/2 x'Qx + c'x   s.t.  Ax + b = 0,  Cx + d >= 0

``Q`` holds the upper triangle only (the reference reads ``col >= row`` only,
meschach/addon2_hqp.c:1078-1086), everything is 0-based CSR with int32 indices
and float64 values.
"""
from __future__ import annotations

import numpy as np


class Program:
    """QP data container mirroring ``Hqp_Program`` (hqp/Hqp_Program.h:33-65)."""

    def __init__(self, n, me, m, Q, A, C, c=None, b=None, d=None):
        self.n, self.me, self.m = int(n), int(me), int(m)
        self.Q, self.A, self.C = Q, A, C  # (indptr, indices, data) triples
        self.c = np.zeros(n) if c is None else c
        self.b = np.zeros(me) if b is None else b
        self.d = np.zeros(m) if d is None else d

    @property
    def dims(self):
        return self.n, self.me, self.m


def _csr(rows, cols, vals, nrows):
    rows = np.asarray(rows, dtype=np.int64)
    cols = np.asarray(cols, dtype=np.int64)
    vals = np.asarray(vals, dtype=np.float64)
    order = np.lexsort((cols, rows))
    rows, cols, vals = rows[order], cols[order], vals[order]
    indptr = np.zeros(nrows + 1, dtype=np.int32)
    np.add.at(indptr, rows + 1, 1)
    indptr = np.cumsum(indptr).astype(np.int32)
    return indptr, cols.astype(np.int32), vals


def banded_qp(n, b, seed=12345):
    """Config C2 of SURVEY.md section 8(d): banded SPD ``Q`` (semi-bandwidth ``b``,
    upper stored), ``A`` (n/2 x n) with ``b``-wide rows at column offset 2i,
    ``C = I`` (simple bounds).  n=40000, b=80 gives KKT dim 1e5, mat_sbw 200."""
    rng = np.random.default_rng(seed)
    me, m = n // 2, n
    # Q upper band
    i = np.repeat(np.arange(n), b)
    j = i + np.tile(np.arange(1, b + 1), n)
    keep = j < n
    i, j = i[keep], j[keep]
    v = rng.uniform(-0.5, 0.5, size=i.size)
    rowsum = np.zeros(n)
    np.add.at(rowsum, i, np.abs(v))
    np.add.at(rowsum, j, np.abs(v))
    diag = 2.0 * b + 1.0 + rowsum
    Q = _csr(np.concatenate([np.arange(n), i]), np.concatenate([np.arange(n), j]),
             np.concatenate([diag, v]), n)
    # A
    ai = np.repeat(np.arange(me), b)
    aj = 2 * ai + np.tile(np.arange(b), me)
    keep = aj < n
    ai, aj = ai[keep], aj[keep]
    A = _csr(ai, aj, rng.uniform(-0.5, 0.5, size=ai.size), me)
    C = _csr(np.arange(m), np.arange(m), np.ones(m), m)
    return Program(n, me, m, Q, A, C, c=rng.uniform(-0.5, 0.5, n),
                   b=np.zeros(me), d=np.ones(m))


def did_like_qp(K, qx=1e-4):
    """QP with the block structure of the reference's double-integrator demo
    (hqp_docp/Prg_DID.C:32-157 through Hqp_Docp::setup_qp, hqp/Hqp_Docp.C:585-755):
    x = [x_0(2), u_0, x_1(2), u_1, ..., x_K(2)], n = 3K+2;
    A = 2K dynamics rows  fx x_k + fu u_k - x_{k+1}, then 2 initial-state and 2
    final-state equalities (me = 2K+4); C = K-1 state upper bounds then K path
    constraints (m = 2K-1).  Q = diag(2 dt on u, qx on x) (upper stored)."""
    dt = 1.0 / K
    n, me, m = 3 * K + 2, 2 * K + 4, 2 * K - 1
    xi = lambda k, c: 3 * k + c
    ui = lambda k: 3 * k + 2
    qd = np.full(n, qx)
    qd[[ui(k) for k in range(K)]] = 2.0 * dt
    Q = _csr(np.arange(n), np.arange(n), qd, n)
    r, c, v = [], [], []
    for k in range(K):
        # f0 = x0 + u dt ; f1 = x0 dt + x1 + u dt^2/2   (Prg_DID.C:80-82, 115-121)
        r += [2 * k] * 3
        c += [xi(k, 0), ui(k), xi(k + 1, 0)]
        v += [1.0, dt, -1.0]
        r += [2 * k + 1] * 4
        c += [xi(k, 0), xi(k, 1), ui(k), xi(k + 1, 1)]
        v += [dt, 1.0, 0.5 * dt * dt, -1.0]
    r += [2 * K, 2 * K + 1, 2 * K + 2, 2 * K + 3]
    c += [xi(0, 0), xi(0, 1), xi(K, 0), xi(K, 1)]
    v += [1.0, 1.0, 1.0, 1.0]
    A = _csr(r, c, v, me)
    b = np.zeros(me)
    b[2 * K], b[2 * K + 1], b[2 * K + 2], b[2 * K + 3] = -1.0, 0.0, 1.0, 0.0
    r, c, v = [], [], []
    for k in range(1, K):  # x_k[1] <= 0.01
        r.append(k - 1)
        c.append(xi(k, 1))
        v.append(-1.0)
    for k in range(K):  # 0.5 dt x0 + x1 <= 0.01  (Prg_DID.C:86-88, 126-130)
        r += [K - 1 + k] * 2
        c += [xi(k, 0), xi(k, 1)]
        v += [-0.5 * dt, -1.0]
    C = _csr(r, c, v, m)
    return Program(n, me, m, Q, A, C, c=np.zeros(n), b=b, d=np.full(m, 0.01))


def lq_docp(K, nx, nu, seed=3, density=1.0, x0_fixed=True, final_eq=0, path_eq=0, path_eq_every=1,
            x_bounds=0):
    """Multistage LQ optimal-control QP (config 4 of BASELINE.json at a chosen size) in
    the layout Hqp_Docp::setup_qp produces (hqp/Hqp_Docp.C:585-755) and
    Hqp_IpLQDOCP::Get_Dim expects (hqp/Hqp_IpLQDOCP.C:201-287):
    x = [x_0, u_0, x_1, u_1, ..., x_K]; A = K*nx dynamics rows
    fx x_k + fu u_k - x_{k+1} (the -1.0 is the last entry of each row), then nx
    initial-state equalities; C = box bounds on every u (2 K nu rows);
    Q = block diagonal (I + low rank on x_k, 0.1 I on u_k), upper stored.
    fx is a dense random matrix scaled to spectral radius ~0.9, fu dense random.
    Options (the equality rows follow the dynamics rows, as Hqp_Docp::setup_qp puts its
    variable / constraint equalities, hqp/Hqp_Docp.C:680-743): ``x0_fixed`` the nx rows
    x_0 = const; ``final_eq`` rows fixing the first components of x_K (they cannot be
    eliminated by a control of their own stage: Hqp_IpLQDOCP carries them back through the
    stages, hqp/Hqp_IpLQDOCP.C:1829-1846); ``path_eq`` random equality rows on (x_k, u_k)
    of every ``path_eq_every``-th stage; ``x_bounds`` upper bounds on the first components
    of every x_k, k >= 1."""
    rng = np.random.default_rng(seed)
    nz = nx + nu
    n = K * nz + nx
    xi = lambda k: k * nz
    ui = lambda k: k * nz + nx
    fx = rng.uniform(-1, 1, (nx, nx))
    if density < 1.0:
        fx *= rng.uniform(0, 1, (nx, nx)) < density
    fx *= 0.9 / max(np.abs(np.linalg.eigvals(fx)).max(), 1e-12)
    fu = rng.uniform(-1, 1, (nx, nu))
    # Q
    qr, qc, qv = [], [], []
    low = rng.uniform(-0.3, 0.3, (nx, 2))
    Lxx = np.eye(nx) + low @ low.T
    iu, ju = np.triu_indices(nx)
    for k in range(K + 1):
        qr.append(xi(k) + iu), qc.append(xi(k) + ju), qv.append(Lxx[iu, ju])
        if k < K:
            qr.append(ui(k) + np.arange(nu)), qc.append(ui(k) + np.arange(nu)), qv.append(np.full(nu, 0.1))
    Q = _csr(np.concatenate(qr), np.concatenate(qc), np.concatenate(qv), n)
    # A: dynamics (columns ascending: x_k, u_k, then the single -1 on x_{k+1})
    ar, ac, av = [], [], []
    rows = np.arange(nx)
    for k in range(K):
        r0 = k * nx
        rr = np.repeat(r0 + rows, nx)
        ar.append(rr), ac.append(xi(k) + np.tile(np.arange(nx), nx)), av.append(fx.ravel())
        rr = np.repeat(r0 + rows, nu)
        ar.append(rr), ac.append(ui(k) + np.tile(np.arange(nu), nx)), av.append(fu.ravel())
        ar.append(r0 + rows), ac.append(xi(k + 1) + rows), av.append(np.full(nx, -1.0))
    me = K * nx
    bvals = [np.zeros(K * nx)]
    if x0_fixed:
        ar.append(me + rows), ac.append(xi(0) + rows), av.append(np.ones(nx))
        bvals.append(-rng.uniform(-1, 1, nx))  # x_0 fixed
        me += nx
    if path_eq:
        for k in range(0, K, path_eq_every):
            for _ in range(path_eq):
                cols = np.arange(xi(k), xi(k) + nz)
                ar.append(np.full(nz, me)), ac.append(cols), av.append(rng.uniform(-1, 1, nz))
                bvals.append(rng.uniform(-0.1, 0.1, 1))
                me += 1
    if final_eq:
        fr = np.arange(final_eq)
        ar.append(me + fr), ac.append(xi(K) + fr), av.append(np.ones(final_eq))
        bvals.append(rng.uniform(-0.1, 0.1, final_eq))
        me += final_eq
    ar, ac, av = np.concatenate(ar), np.concatenate(ac), np.concatenate(av)
    keep = av != 0.0
    A = _csr(ar[keep], ac[keep], av[keep], me)
    b = np.concatenate(bvals)
    # C: -1 <= u <= 1
    m = 2 * K * nu
    cr = np.arange(m)
    cc = np.concatenate([[ui(k) + j for j in range(nu)] * 2 for k in range(K)]) if K else np.zeros(0, int)
    cv = np.concatenate([np.concatenate([np.ones(nu), -np.ones(nu)]) for _ in range(K)]) if K else np.zeros(0)
    if x_bounds:
        xb = np.concatenate([xi(k) + np.arange(x_bounds) for k in range(1, K + 1)])
        cr = np.concatenate([cr, m + np.arange(xb.size)])
        cc = np.concatenate([cc, xb])
        cv = np.concatenate([cv, -np.ones(xb.size)])
        m += xb.size
    C = _csr(cr, cc, cv, m)
    return Program(n, me, m, Q, A, C, c=rng.uniform(-0.1, 0.1, n), b=b, d=np.ones(m))


class DenseDocp:
    """A multistage QP whose dynamics rows are handed over as dense blocks
    (hqpkkt_analyze_staged / hqpkkt_set_values_staged): ``F[k]`` = [fx_k fu_k], row-major
    nx[k+1] x (nx[k] + nu[k]) (numpy arrays, or torch CUDA tensors for device hand-over);
    ``E`` = the other equality rows (me_rest x n CSR).  Vectors of length ``me`` keep the
    reference's order: dynamics rows first."""

    def __init__(self, nx, nu, Q, E, C, F, me_rest, m, c=None, b=None, d=None):
        self.nx, self.nu = [int(v) for v in nx], [int(v) for v in nu]
        self.K = len(self.nu)
        self.n = sum(self.nx) + sum(self.nu)
        self.ndyn = sum(self.nx[1:])
        self.me_rest, self.me, self.m = int(me_rest), self.ndyn + int(me_rest), int(m)
        self.Q, self.E, self.C, self.F = Q, E, C, F
        self.c = np.zeros(self.n) if c is None else c
        self.b = np.zeros(self.me) if b is None else b
        self.d = np.zeros(self.m) if d is None else d

    @property
    def dims(self):
        return self.n, self.me, self.m


def dense_docp_from_program(prog, nx, nu):
    """The dense-dynamics form of a Program in Hqp_Docp's layout (tests: both hand-overs must
    give the same results)."""
    K = len(nu)
    p, i, x = prog.A
    nmk = np.concatenate([[0], np.cumsum([nx[k] + nu[k] for k in range(K)])])
    F, row = [], 0
    for k in range(K):
        nz = nx[k] + nu[k]
        blk = np.zeros((nx[k + 1], nz))
        for li in range(nx[k + 1]):
            r = row + li
            cols, vals = i[p[r]:p[r + 1] - 1], x[p[r]:p[r + 1] - 1]
            blk[li, cols - nmk[k]] = vals
        F.append(blk)
        row += nx[k + 1]
    ndyn = row
    Ep = (p[ndyn:] - p[ndyn]).astype(np.int32)
    E = (Ep, i[p[ndyn]:].astype(np.int32), x[p[ndyn]:].copy())
    return DenseDocp(nx, nu, prog.Q, E, prog.C, F, prog.me - ndyn, prog.m, c=prog.c, b=prog.b, d=prog.d)


def random_sparse_qp(n, me, m, row_nnz=4, seed=7):
    """Irregular (non-banded) QP: random sparse A, C rows, Q = diag + random
    symmetric sparse part made diagonally dominant.  Exercises the general path."""
    rng = np.random.default_rng(seed)
    i = rng.integers(0, n, size=2 * n)
    j = rng.integers(0, n, size=2 * n)
    lo, hi = np.minimum(i, j), np.maximum(i, j)
    keep = lo != hi
    key = np.unique(lo[keep].astype(np.int64) * n + hi[keep])
    lo, hi = (key // n).astype(np.int64), (key % n).astype(np.int64)
    v = rng.uniform(-0.5, 0.5, size=lo.size)
    rowsum = np.zeros(n)
    np.add.at(rowsum, lo, np.abs(v))
    np.add.at(rowsum, hi, np.abs(v))
    Q = _csr(np.concatenate([np.arange(n), lo]), np.concatenate([np.arange(n), hi]),
             np.concatenate([1.0 + rowsum, v]), n)

    def rows(nr):
        rr, cc = [], []
        for r in range(nr):
            cols = np.unique(rng.integers(0, n, size=row_nnz))
            rr += [r] * cols.size
            cc += list(cols)
        return _csr(rr, cc, rng.uniform(-1.0, 1.0, size=len(rr)), nr)

    return Program(n, me, m, Q, rows(me), rows(m), c=rng.uniform(-1, 1, n),
                   b=rng.uniform(-1, 1, me), d=rng.uniform(0.5, 1.5, m))


def cute_like_qp(n, nnz_lo=10, nnz_hi=100, window=400, far=0.001, eq_frac=0.3, bound_frac=0.6, seed=17):
    """SURVEY.md section 8(d) C5's row density - "CUTE-style" sparse QP with 10 ... 100 (nnz_lo ... nnz_hi) entries per
    row - with the locality such programs have (hqp_cute/hqp_cute.tcl:22-46 runs the CUTE collection through
    RedSpBKP: discretised control and structural problems, not random graphs): a row of Q couples variable i with
    nnz_lo ... nnz_hi variables drawn from a window of `window` columns behind it, a fraction `far` of the entries
    anywhere (irregular part: one entry in a thousand - 2600 far couplings at n = 10^5; with one in a hundred every second
    variable has one and the graph has no small separators left: fronts of 19 000 rows at n = 5 10^4); diagonally dominant.  n eq_frac equality rows over nnz_lo / 2 ... nnz_hi / 2 columns of a
    window around a random centre; bounds (one-entry inequality rows) on a fraction bound_frac of the variables."""
    rng = np.random.default_rng(seed)
    cnt = rng.integers(max(1, nnz_lo // 2), max(1, nnz_hi // 2) + 1, size=n)  # (entries behind the diagonal: a full row has about twice as many)
    i = np.repeat(np.arange(n, dtype=np.int64), cnt)
    off = rng.integers(1, window + 1, size=i.size)
    j = i + off
    is_far = rng.random(i.size) < far
    j[is_far] = rng.integers(0, n, size=int(is_far.sum()))
    keep = (j < n) & (j != i)
    lo, hi = np.minimum(i[keep], j[keep]), np.maximum(i[keep], j[keep])
    key = np.unique(lo * n + hi)
    lo, hi = key // n, key % n
    v = rng.uniform(-0.5, 0.5, size=lo.size)
    rowsum = np.zeros(n)
    np.add.at(rowsum, lo, np.abs(v))
    np.add.at(rowsum, hi, np.abs(v))
    Q = _csr(np.concatenate([np.arange(n), lo]), np.concatenate([np.arange(n), hi]), np.concatenate([1.0 + rowsum, v]), n)
    me = int(eq_frac * n)
    centre = np.sort(rng.integers(0, n, size=me))
    ecnt = rng.integers(max(1, nnz_lo // 2), max(2, nnz_hi // 2) + 1, size=me)
    er = np.repeat(np.arange(me, dtype=np.int64), ecnt)
    ec = np.repeat(centre, ecnt) + rng.integers(-window // 2, window // 2 + 1, size=er.size)
    ec = np.clip(ec, 0, n - 1)
    ekey = np.unique(er * n + ec)
    er, ec = ekey // n, ekey % n
    A = _csr(er, ec, rng.uniform(-1.0, 1.0, size=er.size), me)
    bounded = np.flatnonzero(rng.random(n) < bound_frac)
    m = bounded.size
    C = _csr(np.arange(m), bounded, np.ones(m), m)
    return Program(n, me, m, Q, A, C, c=rng.uniform(-1, 1, n), b=0.1 * rng.uniform(-1, 1, me), d=rng.uniform(0.5, 1.5, m))


def grid_sparse_qp(gx, gy, seed=11, eq_every=3, bound_frac=0.5, long_range=0):
    """Mesh-structured QP, the sparsity of a discretised control problem (what the CUTE collection's
    large programs look like, BASELINE configs[4]): one variable per cell of a gx x gy grid, Q couples
    a cell with its right and lower neighbour (diagonally dominant), every eq_every-th cell has an
    equality row over the cell and those two neighbours, a fraction of the cells is bounded (one-entry
    inequality rows).  long_range > 0 adds that many random far couplings to Q (irregular part)."""
    rng = np.random.default_rng(seed)
    n = gx * gy
    idx = np.arange(n).reshape(gy, gx)
    right = np.stack([idx[:, :-1].ravel(), idx[:, 1:].ravel()])
    down = np.stack([idx[:-1, :].ravel(), idx[1:, :].ravel()])
    lo = np.concatenate([right[0], down[0]])
    hi = np.concatenate([right[1], down[1]])
    if long_range:
        a, b = rng.integers(0, n, long_range), rng.integers(0, n, long_range)
        keep = a != b
        key = np.unique(np.minimum(a, b)[keep].astype(np.int64) * n + np.maximum(a, b)[keep])
        near = set((lo.astype(np.int64) * n + hi).tolist())
        key = np.array([k for k in key.tolist() if k not in near], dtype=np.int64)
        lo, hi = np.concatenate([lo, key // n]), np.concatenate([hi, key % n])
    v = rng.uniform(-0.5, 0.5, size=lo.size)
    rowsum = np.zeros(n)
    np.add.at(rowsum, lo, np.abs(v))
    np.add.at(rowsum, hi, np.abs(v))
    Q = _csr(np.concatenate([np.arange(n), lo]), np.concatenate([np.arange(n), hi]),
             np.concatenate([1.0 + rowsum, v]), n)
    cells = idx[:-1, :-1].ravel()
    cells = cells[(cells // gx + cells % gx) % eq_every == 0]
    me = cells.size
    rr = np.repeat(np.arange(me), 3)
    cc = np.stack([cells, cells + 1, cells + gx], axis=1).ravel()
    A = _csr(rr, cc, rng.uniform(0.5, 1.5, size=rr.size) * rng.choice([-1.0, 1.0], size=rr.size), me)
    bounded = np.flatnonzero(rng.random(n) < bound_frac)
    m = bounded.size
    Cm = _csr(np.arange(m), bounded, np.ones(m), m)
    return Program(n, me, m, Q, A, Cm, c=rng.uniform(-1, 1, n), b=rng.uniform(-1, 1, me), d=rng.uniform(0.5, 1.5, m))


def banded_long_range_qp(n, band, far, seed=7, min_dist=1000):
    """Irregular sparsity without a mesh behind it (BASELINE configs[4], SURVEY 8(d) C5: "row density" instead of
    0.1 % fill): the banded QP of `banded_qp` (2 band + 1 entries per row of Q, band-wide rows of A) plus `far` random
    couplings Q_ij between variables at least `min_dist` apart - every one of them ties two distant parts of the band
    together, so a band ordering sees a bandwidth of ~n while a dissection of the graph itself only has to put one end of
    each crossing coupling into a separator."""
    rng = np.random.default_rng(seed)
    prog = banded_qp(n, band, seed)
    p, i, x = prog.Q
    rows = np.repeat(np.arange(n), np.diff(p))
    a, b = rng.integers(0, n, far), rng.integers(0, n, far)
    keep = np.abs(a - b) >= min(min_dist, n // 4)
    lo, hi = np.minimum(a, b)[keep], np.maximum(a, b)[keep]
    key = np.unique(lo.astype(np.int64) * n + hi)
    lo, hi = key // n, key % n
    v = rng.uniform(-0.05, 0.05, lo.size)  # (small against the diagonal: Q stays positive definite)
    Q = _csr(np.concatenate([rows, lo]), np.concatenate([i, hi]), np.concatenate([x, v]), n)
    return Program(n, prog.me, prog.m, Q, prog.A, prog.C, c=prog.c, b=prog.b, d=prog.d)


def ip_state(prog, seed=1, spread=0.0):
    """Strictly positive (z, w) and right-hand sides r1..r4 as an interior-point
    iteration would pass them (hqp/Hqp_IpsMehrotra.C:425-445, 527-530).
    ``spread`` > 0 draws z/w log-uniformly over 10**(+-spread) to mimic late
    iterations where w/z spans many decades."""
    rng = np.random.default_rng(seed)
    n, me, m = prog.dims
    if spread > 0:
        z = 10.0 ** rng.uniform(-spread, spread, m)
        w = 10.0 ** rng.uniform(-spread, spread, m)
    else:
        z = 0.1 + rng.uniform(0, 1, m)
        w = 0.1 + rng.uniform(0, 1, m)
    r = [rng.uniform(-0.5, 0.5, k) for k in (n, me, m, m)]
    return z, w, r[0], r[1], r[2], r[3]


def c4_docp_csr(K, nx, nu, seed=0):
    """The QP family of the headline workload (BASELINE configs[3]: multistage LQ optimal control, K stages of nx states and
    nu controls, dense dynamics, x_0 fixed, box bounds on u) in CSR form (Hqp_Docp's layout): what the CPU reference's
    Hqp_IpLQDOCP is timed on (bench.py) and checked against at wide stages (tests/golden_lqdocp_wide)."""
    rng = np.random.default_rng(seed)
    nz = nx + nu
    n = K * nz + nx
    ar, ac, av = [], [], []
    rows = np.arange(nx)
    for k in range(K):
        blk = rng.uniform(-1.0, 1.0, (nx, nz))
        blk[:, :nx] *= 0.9 / np.sqrt(nx / 3.0)
        ar.append(np.repeat(k * nx + rows, nz)), ac.append(np.tile(k * nz + np.arange(nz), nx)), av.append(blk.ravel())
        ar.append(k * nx + rows), ac.append((k + 1) * nz + rows), av.append(np.full(nx, -1.0))
    ar.append(K * nx + rows), ac.append(rows), av.append(np.ones(nx))
    A = _csr(np.concatenate(ar), np.concatenate(ac), np.concatenate(av), K * nx + nx)
    qd = np.ones(n)
    qd[:K * nz].reshape(K, nz)[:, nx:] = 0.1
    Q = (np.arange(n + 1, dtype=np.int32), np.arange(n, dtype=np.int32), qd)
    ucols = (np.arange(K)[:, None] * nz + nx + np.arange(nu)[None, :]).ravel()
    cols = np.concatenate([ucols, ucols]).astype(np.int32)
    C = (np.arange(cols.size + 1, dtype=np.int32), cols, np.concatenate([np.ones(ucols.size), -np.ones(ucols.size)]))
    return Program(n, K * nx + nx, cols.size, Q, A, C)
