"""Host-side mirror of the reference's ``Hqp_IpMatrix`` plugin interface
(hqp/Hqp_IpMatrix.h:42-89) over the C ABI of include/hqpkkt.h.

Same method names, argument meaning and error behaviour as the reference:

    m = IpSpBKP()            # Hqp_IpSpBKP    (hqp/Hqp_IpSpBKP.C)
    m = IpRedSpBKP()         # Hqp_IpRedSpBKP (hqp/Hqp_IpRedSpBKP.C)
    m.init(qp)               # structure: RCM, mat_sbw, symbolic factorisation
    m.update(qp)             # new values, same pattern
    m.factor(qp, z, w)
    res = m.solve(qp, z, w, r1, r2, r3, r4, dx, dy, dz, dw)   # fills dx..dw
    m.step(...), m.residuum(...)

``qp`` is an :class:`hqp_amd.problems.Program` (the Hqp_Program data contract).
Vectors are numpy float64 arrays (host pointers, like Meschach ``VEC::ve``) or
torch CUDA tensors (device pointers, used in place) -- one kind per object,
chosen at construction (``device_vectors=True``).  A numerically singular system
raises :class:`SingularError`, the counterpart of ``m_error(E_SING, ...)``
(meschach/err.h:63,88) which the interior-point solvers catch
(hqp/Hqp_IpsMehrotra.C:525-536).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


class KktError(RuntimeError):
    def __init__(self, code, where):
        super().__init__(f"hqpkkt status {code} in {where}: {_lib.strerror(code)}")
        self.code = code


class SingularError(KktError):
    """E_SING of the reference (meschach/err.h:88)."""


def _check(code, where):
    if code == _lib.OK:
        return
    if code == _lib.E_SING:
        raise SingularError(code, where)
    raise KktError(code, where)


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a


class Hqp_IpMatrix:
    """Abstract base (hqp/Hqp_IpMatrix.h:42-89)."""

    _mode = None
    _name = None

    def __init__(self, device=0, device_vectors=False, mat_tol=1.0, mat_eps=1e-10,
                 pivot_eps=None, leaf_size=0, max_pivots=0, zd_policy=None, shard=None, slack_policy=None, small_fronts=True, upd_pingpong_mb=0,
                 amalgamation=False, ordering=0):
        L = _lib.lib()
        o = _lib.Opts()
        L.hqpkkt_default_opts(C.byref(o))
        o.mode = self._mode
        o.device = device
        o.loc = _lib.LOC_DEVICE if device_vectors else _lib.LOC_HOST
        o.tol, o.eps = mat_tol, mat_eps
        if pivot_eps is not None:
            o.pivot_eps = pivot_eps
        o.leaf_size, o.max_pivots = leaf_size, max_pivots
        if zd_policy is not None:
            o.zd_policy = zd_policy
        if slack_policy is not None:
            o.slack_policy = slack_policy
        o.no_small_fronts = 0 if small_fronts else 1
        o.upd_pingpong_mb = upd_pingpong_mb
        o.amalgamation = 1 if amalgamation else 0
        o.ordering = int(ordering)
        self._L = L
        self._h = C.c_void_p()
        self._device_vectors = bool(device_vectors)
        self._dev = int(device)
        _check(L.hqpkkt_create(C.byref(o), C.byref(self._h)), "create")
        self._keep = None
        self._xchg = None
        self.n = self.me = self.m = 0
        if shard is not None:
            if hasattr(shard, "fn"):  # hqp_amd.dist.RcclShard: stream-ordered RCCL collectives
                _check(L.hqpkkt_set_shard_stream(self._h, shard.rank, shard.world, shard.fn, shard._ctx), "set_shard_stream")
                self._xchg = shard
            else:
                self.set_shard(*shard)

    def set_shard(self, rank, count, exchange=None):
        """One system over ``count`` ranks (call before init()).  ``exchange(op,
        device_pointer, slot_elems, nslots)`` performs the collectives, see
        hqp_amd.dist.make_exchange and hqpkkt_set_shard in include/hqpkkt.h."""
        import sys
        import traceback

        def tramp(_ctx, op, buf, slot, nslots):
            try:
                exchange(op, buf, slot, nslots)
                return 0
            except Exception:  # never unwind through the C frames
                traceback.print_exc(file=sys.stderr)
                return 1

        cb = _lib.EXCHANGE_FN(tramp) if exchange is not None else C.cast(None, _lib.EXCHANGE_FN)
        _check(self._L.hqpkkt_set_shard(self._h, rank, count, cb, None), "set_shard")
        self._xchg = cb  # keep the thunk alive as long as the handle

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._L.hqpkkt_destroy(h)
            self._h = None

    # -- Tcl-visible members of the reference -------------------------------
    @property
    def mat_sbw(self):
        v = C.c_int()
        _check(self._L.hqpkkt_get_sbw(self._h, C.byref(v)), "get_sbw")
        return v.value

    def set_mat_tol(self, tol):
        _check(self._L.hqpkkt_set_tol(self._h, tol), "set_tol")

    def set_mat_eps(self, eps):
        _check(self._L.hqpkkt_set_eps(self._h, eps), "set_eps")

    def name(self):
        return self._name

    # -- pointer plumbing ------------------------------------------------------
    def _ptr(self, a, size, what, out=False):
        if a is None:
            if size:
                raise KktError(_lib.E_NULL, what)
            return None
        if self._device_vectors:
            if not (hasattr(a, "data_ptr") and a.is_cuda):
                raise TypeError(f"{what}: device_vectors=True needs torch CUDA float64 tensors")
            if a.numel() != size or str(a.dtype) != "torch.float64" or not a.is_contiguous():
                raise KktError(_lib.E_SIZES, what)
            return C.c_void_p(a.data_ptr())
        if not isinstance(a, np.ndarray) or a.dtype != np.float64 or not a.flags.c_contiguous:
            if out:
                raise TypeError(f"{what}: output must be a C-contiguous float64 numpy array")
            a = np.ascontiguousarray(a, dtype=np.float64)
            self._tmp.append(a)
        if a.size != size:
            raise KktError(_lib.E_SIZES, what)  # the reference asserts (Hqp_IpSpBKP.C:189-192)
        return C.c_void_p(a.ctypes.data)

    def _vecs(self, z, w, r1, r2, r3, r4, dx, dy, dz, dw, out=True):
        self._tmp = []
        n, me, m = self.n, self.me, self.m
        sizes = (m, m, n, me, m, m)
        ins = [self._ptr(a, s, nm) for a, s, nm in zip((z, w, r1, r2, r3, r4), sizes,
                                                      ("z", "w", "r1", "r2", "r3", "r4"))]
        outs = [self._ptr(a, s, nm, out=out) for a, s, nm in zip((dx, dy, dz, dw), (n, me, m, m),
                                                                 ("dx", "dy", "dz", "dw"))]
        return ins + outs

    # -- the plugin interface ----------------------------------------------------
    def init(self, qp):
        """Hqp_IpSpBKP::init (hqp/Hqp_IpSpBKP.C:76-114): analyse, then update."""
        self.n, self.me, self.m = qp.dims
        arrs = []
        for (p, i, _x) in (qp.Q, qp.A, qp.C):
            arrs += [_i32(p), _i32(i)]
        self._keep = arrs
        sbw = C.c_int()
        ptrs = [C.c_void_p(a.ctypes.data) if a.size else None for a in arrs]
        _check(self._L.hqpkkt_analyze(self._h, self.n, self.me, self.m, *ptrs, C.byref(sbw)), "init")
        self.update(qp)

    def update(self, qp):
        """Hqp_IpSpBKP::update (hqp/Hqp_IpSpBKP.C:117-136)."""
        vals = []
        for (_p, _i, x) in (qp.Q, qp.A, qp.C):
            if self._device_vectors and hasattr(x, "data_ptr"):
                vals.append(C.c_void_p(x.data_ptr()))
            elif self._device_vectors:
                import torch
                t = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float64)).cuda()
                self._keepvals = getattr(self, "_keepvals", []) + [t]
                vals.append(C.c_void_p(t.data_ptr()))
            else:
                a = np.ascontiguousarray(x, dtype=np.float64)
                self._keep.append(a)
                vals.append(C.c_void_p(a.ctypes.data) if a.size else None)
        _check(self._L.hqpkkt_set_values(self._h, *vals), "update")
        self._keepvals = []

    def factor(self, qp, z, w):
        self._tmp = []
        m = self.m
        _check(self._L.hqpkkt_factor(self._h, self._ptr(z, m, "z"), self._ptr(w, m, "w")), "factor")

    def step(self, qp, z, w, r1, r2, r3, r4, dx, dy, dz, dw):
        _check(self._L.hqpkkt_step(self._h, *self._vecs(z, w, r1, r2, r3, r4, dx, dy, dz, dw)), "step")

    def solve(self, qp, z, w, r1, r2, r3, r4, dx, dy, dz, dw):
        """Hqp_IpMatrix::solve (hqp/Hqp_IpMatrix.C:65-128); returns the residual."""
        res = C.c_double()
        _check(self._L.hqpkkt_solve(self._h, *self._vecs(z, w, r1, r2, r3, r4, dx, dy, dz, dw),
                                    C.byref(res)), "solve")
        return res.value

    def residuum(self, qp, z, w, r1, r2, r3, r4, dx, dy, dz, dw):
        """Hqp_IpMatrix::residuum (hqp/Hqp_IpMatrix.C:131-178)."""
        res = C.c_double()
        _check(self._L.hqpkkt_residual(self._h, *self._vecs(z, w, r1, r2, r3, r4, dx, dy, dz, dw, out=False),
                                       C.byref(res)), "residuum")
        return res.value

    # -- introspection ---------------------------------------------------------------
    def stats(self):
        s = _lib.Stats()
        _check(self._L.hqpkkt_get_stats(self._h, C.byref(s)), "get_stats")
        return s.asdict()

    def perm(self):
        """_QP2J of the reference: band position of each QP index."""
        p = np.zeros(self.stats()["dim"], dtype=np.int32)
        _check(self._L.hqpkkt_get_perm(self._h, C.c_void_p(p.ctypes.data)), "get_perm")
        return p

    def set_profile(self, on=True):
        _check(self._L.hqpkkt_set_profile(self._h, 1 if on else 0), "set_profile")

    def profile(self):
        """{kernel class: (summed device ms, launches)} since set_profile()."""
        ms = (C.c_double * 16)()
        ln = (C.c_longlong * 16)()
        k = self._L.hqpkkt_get_profile(self._h, 16, ms, ln)
        return {self._L.hqpkkt_profile_class_name(c).decode(): (ms[c], ln[c]) for c in range(k)}

    def set_stream(self, hip_stream):
        _check(self._L.hqpkkt_set_stream(self._h, C.c_void_p(hip_stream)), "set_stream")

    def debug(self, what):
        k = C.c_longlong()
        _check(self._L.hqpkkt_debug_get(self._h, what, None, C.byref(k)), "debug_get")
        out = np.zeros(max(k.value, 1), dtype=np.int32)
        _check(self._L.hqpkkt_debug_get(self._h, what, C.c_void_p(out.ctypes.data), C.byref(k)), "debug_get")
        return out[: k.value]

    def franke(self, qp, eps=1e-10, max_iters=200, hot_start=0, qp_mu0=0.0):
        """Device-resident run of the reference's other interior-point solver, Hqp_IpsFranke
        (``hqpkkt_franke``): returns (x, y, z, w, info).  ``hot_start`` 1: Hqp_IpsFranke::hot_start
        from the iterate this handle's previous franke() call ended with.  ``qp_mu0``: the reference's
        interface variable of that name (cold start, hqp/Hqp_IpsFranke.C:167-173)."""
        return self.mehrotra(qp, eps, max_iters, hot_start, _entry="hqpkkt_franke", qp_mu0=qp_mu0)

    def mehrotra(self, qp, eps=1e-10, max_iters=200, hot_start=0, init_method=0, _entry="hqpkkt_mehrotra", qp_mu0=0.0):
        """Device-resident Mehrotra predictor-corrector solve of the QP, the restatement of
        hqp/Hqp_IpsMehrotra.C behind ``hqpkkt_mehrotra``: returns (x, y, z, w, info).
        init()/update() must have been called with ``qp``.  ``hot_start``: 0 cold start,
        1 Hqp_IpsMehrotra::hot_start from this handle's previous solve (which must have run
        with hot_start != 0), 2 cold start that keeps what the next hot start needs."""
        o = _lib.IpOpts()
        self._L.hqpkkt_default_ip_opts(C.byref(o))
        o.eps, o.max_iters, o.hot_start, o.init_method = eps, max_iters, int(hot_start), int(init_method)
        o.qp_mu0 = float(qp_mu0)

        def rowsum(csr, rows):  # sp_norm_inf (meschach/addon2_hqp.c:723-743)
            p, _i, x = csr
            if rows == 0 or len(x) == 0:
                return 0.0
            return float(np.add.reduceat(np.abs(np.r_[np.asarray(x, dtype=float), 0.0]),
                                         np.asarray(p[:-1], dtype=np.int64))[np.diff(p) > 0].max(initial=0.0))

        def ninf(v):
            if hasattr(v, "data_ptr"):  # torch tensor (device-resident problem data)
                return float(v.abs().max()) if v.numel() else 0.0
            return float(np.abs(v).max()) if len(v) else 0.0

        o.norm_Q, o.norm_C, o.norm_d = rowsum(qp.Q, qp.n), rowsum(qp.C, qp.m), ninf(qp.d)
        if getattr(qp, "norm_A", None) is not None:  # (the caller has released the blocks)
            norm_A = float(qp.norm_A)
        elif hasattr(qp, "F"):  # problems.DenseDocp: the dynamics rows are [fx fu | -1], the others are in E
            norm_A = max([float(abs(blk).sum(1).max()) + 1.0 for blk in qp.F] + [rowsum(qp.E, qp.me_rest)])
        else:
            norm_A = rowsum(qp.A, qp.me)
        o.norm_data = max(o.norm_Q, norm_A, o.norm_C, ninf(qp.c), ninf(qp.b), o.norm_d)
        res = _lib.IpResult()
        self._tmp = []
        if self._device_vectors:
            import torch
            dev = torch.device("cuda", self._dev)
            mk = lambda k: torch.zeros(k, dtype=torch.float64, device=dev)
            cin = [v.to(dev) if hasattr(v, "data_ptr") else torch.as_tensor(np.ascontiguousarray(v, dtype=np.float64)).to(dev)
                   for v in (qp.c, qp.b, qp.d)]
        else:
            mk = lambda k: np.zeros(k)
            cin = [np.ascontiguousarray(v, dtype=np.float64) for v in (qp.c, qp.b, qp.d)]
        x, y, z, w = mk(qp.n), mk(qp.me), mk(qp.m), mk(qp.m)
        ptrs = [self._ptr(a, k, nm) for a, k, nm in zip(cin + [x, y, z, w], (qp.n, qp.me, qp.m, qp.n, qp.me, qp.m, qp.m),
                                                      ("c", "b", "d", "x", "y", "z", "w"))]
        _check(getattr(self._L, _entry)(self._h, C.byref(o), *ptrs, C.byref(res)), _entry)
        return x, y, z, w, res.asdict()

    def read_block(self, what, node):
        """Numeric block of a supernode after factor() (tests): 0 panel, 1 inverse of
        L11, 2 X, 3 update block; flat float64 array, column-major."""
        k = C.c_longlong()
        _check(self._L.hqpkkt_debug_read(self._h, what, node, None, 0, C.byref(k)), "debug_read")
        out = np.zeros(max(k.value, 1))
        _check(self._L.hqpkkt_debug_read(self._h, what, node, C.c_void_p(out.ctypes.data), k.value,
                                         C.byref(k)), "debug_read")
        return out[: k.value]

    def structure(self):
        names = ["elim", "piv_start", "npiv", "nborder", "parent", "level", "border_ptr",
                 "border_idx", "entry_row", "entry_col", "node_owner", "exchange_roots"]
        return {nm: self.debug(i) for i, nm in enumerate(names)}


class Hqp_IpSpBKP(Hqp_IpMatrix):
    """Full (n+me+m) KKT system; semantics of hqp/Hqp_IpSpBKP.C."""
    _mode = _lib.MODE_FULL
    _name = "SpBKP"


class Hqp_IpRedSpBKP(Hqp_IpMatrix):
    """Reduced (n+me) system with C'ZW^-1C folded in; semantics of hqp/Hqp_IpRedSpBKP.C."""
    _mode = _lib.MODE_REDUCED
    _name = "RedSpBKP"


class Hqp_IpLQDOCP(Hqp_IpMatrix):
    """The multistage plugin hqp/Hqp_IpLQDOCP.C: stage structure found from the staircase of
    A (Get_Dim, :201-287), dense per-stage blocks, the extended Riccati recursion as fp64
    MFMA products (HQPKKT_MODE_STAGED).  ``set_stages(nx, nu)`` before init() gives the stage
    sizes explicitly."""
    _mode = _lib.MODE_STAGED
    _name = "LQDOCP"

    def set_stages(self, nx, nu):
        nx, nu = _i32(nx), _i32(nu)
        _check(self._L.hqpkkt_set_stages(self._h, len(nu), C.c_void_p(nx.ctypes.data),
                                         C.c_void_p(nu.ctypes.data) if nu.size else None), "set_stages")

    def init_dense(self, dq):
        """init() for a :class:`hqp_amd.problems.DenseDocp`: the dynamics as dense blocks
        (hqpkkt_analyze_staged + hqpkkt_set_values_staged)."""
        self.n, self.me, self.m = dq.dims
        nx, nu = _i32(dq.nx), _i32(dq.nu)
        arrs = []
        for (p, i, _x) in (dq.Q, dq.E, dq.C):
            arrs += [_i32(p), _i32(i)]
        self._keep = arrs + [nx, nu]
        ptrs = [C.c_void_p(a.ctypes.data) if a.size else None for a in arrs]
        _check(self._L.hqpkkt_analyze_staged(self._h, dq.K, C.c_void_p(nx.ctypes.data), C.c_void_p(nu.ctypes.data),
                                             dq.n, dq.me_rest, dq.m, *ptrs), "init_dense")
        self.update_dense(dq)

    def update_dense(self, dq):
        vals, keep = [], []
        for (_p, _i, x) in (dq.Q, dq.E, dq.C):
            if self._device_vectors and not hasattr(x, "data_ptr"):
                import torch
                x = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float64)).cuda()
            if hasattr(x, "data_ptr"):
                keep.append(x)
                vals.append(C.c_void_p(x.data_ptr()) if x.numel() else None)
            else:
                a = np.ascontiguousarray(x, dtype=np.float64)
                keep.append(a)
                vals.append(C.c_void_p(a.ctypes.data) if a.size else None)
        fp = (C.c_void_p * dq.K)()
        ld = (C.c_longlong * dq.K)()
        for k, blk in enumerate(dq.F):
            if hasattr(blk, "data_ptr"):
                if not self._device_vectors or blk.stride(1) != 1:
                    raise TypeError("F blocks: row-major torch CUDA tensors need device_vectors=True")
                fp[k], ld[k] = blk.data_ptr(), blk.stride(0)
                keep.append(blk)
            else:
                if self._device_vectors:
                    raise TypeError("device_vectors=True needs torch CUDA F blocks")
                a = np.ascontiguousarray(blk, dtype=np.float64)
                fp[k], ld[k] = a.ctypes.data, a.shape[1]
                keep.append(a)
        _check(self._L.hqpkkt_set_values_staged(self._h, vals[0], fp, ld, vals[1], vals[2]), "update_dense")

    def stage_structure(self):
        names = {"nk": 20, "mk": 21, "nmk": 22, "eq_ptr": 23, "eq_rows": 24, "fix_rows": 25, "cap": 26}
        return {nm: self.debug(i) for nm, i in names.items()}

    def stage_ranks(self):
        """(rank, carried rows) per stage of the last factorisation (tests)."""
        K1 = len(self.debug(20))
        out = np.zeros(2 * K1, dtype=np.int32)
        _check(self._L.hqpkkt_debug_stage_ranks(self._h, C.c_void_p(out.ctypes.data), out.size), "stage_ranks")
        return out.reshape(K1, 2)


class Hqp_IpLQDOCPFull(Hqp_IpMatrix):
    """The same KKT system under the plugin name LQDOCP solved by the full-system engine
    (the band ordering carries the stage structure): the comparison partner of the STAGED
    engine, and what the reference-side shim falls back to for QPs whose stages the STAGED
    kernels do not hold."""
    _mode = _lib.MODE_FULL
    _name = "LQDOCP"


IpSpBKP = Hqp_IpSpBKP
IpRedSpBKP = Hqp_IpRedSpBKP
IpLQDOCP = Hqp_IpLQDOCP
IpLQDOCPFull = Hqp_IpLQDOCPFull


def bench_dgemm(M, N, K, lower=False, mirror=False, reps=5, device=0):
    """(ms per launch, TFLOP/s, max relative error) of the STAGED engine's fp64 MFMA product."""
    ms, err = C.c_double(), C.c_double()
    _check(_lib.lib().hqpkkt_debug_dgemm(device, M, N, K, int(lower), int(mirror), reps, C.byref(ms), C.byref(err)), "debug_dgemm")
    flops = (1.0 if lower else 2.0) * M * N * K
    return ms.value, flops / (ms.value * 1e-3) / 1e12, err.value


def sk_table(tiles, nslab, grid=512):
    """The work lists of the cut form of the fp64 product (host only): (units[grid, stride, 6], pieces, whole_a, whole_b),
    a unit = (tile or -1, first k-slab, one past the last, first parking slot of the tile, pieces of the tile, piece)."""
    import numpy as np
    pieces, wa, wb = C.c_longlong(), C.c_int(), C.c_int()
    stride = _lib.lib().hqpkkt_debug_sk_table(tiles, nslab, grid, None, 0, C.byref(pieces), C.byref(wa), C.byref(wb))
    if stride <= 0:
        return None
    u = np.zeros((grid, stride, 6), dtype=np.int32)
    got = _lib.lib().hqpkkt_debug_sk_table(tiles, nslab, grid, u.ctypes.data_as(C.POINTER(C.c_int)), u.size, C.byref(pieces), C.byref(wa), C.byref(wb))
    assert got == stride
    return u, pieces.value, wa.value, wb.value


def selftest_mfma(device=0):
    err = C.c_double()
    _check(_lib.lib().hqpkkt_selftest_mfma(device, C.byref(err)), "selftest_mfma")
    return err.value
