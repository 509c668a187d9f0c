"""TEST INFRASTRUCTURE ONLY -- ctypes binding of oracle/_ref/libhqpref.so.

The library holds the reference's own Hqp_IpSpBKP / Hqp_IpRedSpBKP
(hqp/Hqp_IpSpBKP.C, hqp/Hqp_IpRedSpBKP.C, hqp/Hqp_IpMatrix.C, hqp/spBKP.C,
hqp/sprcm.C + Meschach) compiled by oracle/Makefile from /root/reference.
It exists only where that build ran (this container; the prebuilt .so travels
to the GPU box).  ``available()`` tells whether it can be loaded.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libhqpref.so")
_lib = None
_err = None

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def _load():
    global _lib, _err
    if _lib is not None or _err is not None:
        return _lib
    try:
        os.environ.setdefault("TCL_LIBRARY", "/opt/conda/lib/tcl8.6")
        lib = C.CDLL(_PATH)
    except OSError as e:  # not built / libtcl missing
        _err = e
        return None
    lib.hqpref_create.restype = C.c_void_p
    lib.hqpref_create.argtypes = [C.c_int]
    lib.hqpref_set_params.argtypes = [C.c_void_p, C.c_double, C.c_double]
    lib.hqpref_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int] + [_ip, _ip, _dp] * 3 + [C.POINTER(C.c_double)]
    lib.hqpref_update.argtypes = [C.c_void_p] + [_ip, _ip, _dp] * 3
    lib.hqpref_factor.argtypes = [C.c_void_p, _dp, _dp, C.POINTER(C.c_double)]
    lib.hqpref_step.argtypes = [C.c_void_p] + [_dp] * 10
    lib.hqpref_solve.argtypes = [C.c_void_p] + [_dp] * 10 + [C.POINTER(C.c_double)] * 2
    lib.hqpref_residuum.argtypes = [C.c_void_p] + [_dp] * 10 + [C.POINTER(C.c_double)]
    lib.hqpref_sbw.argtypes = [C.c_void_p]
    lib.hqpref_dim.argtypes = [C.c_void_p]
    lib.hqpref_get_perm.argtypes = [C.c_void_p, _ip]
    lib.hqpref_get_pivot.argtypes = [C.c_void_p, _ip]
    lib.hqpref_matrix_nnz.restype = C.c_long
    lib.hqpref_matrix_nnz.argtypes = [C.c_void_p, C.c_int]
    lib.hqpref_get_matrix.argtypes = [C.c_void_p, C.c_int, _ip, _ip, _dp]
    lib.hqpref_destroy.argtypes = [C.c_void_p]
    _lib = lib
    return lib


def available():
    return _load() is not None


def load_error():
    _load()
    return _err


class RefError(RuntimeError):
    def __init__(self, code, where):
        super().__init__(f"reference raised Meschach error {code} in {where}")
        self.code = code


def _pad(a):
    # ndpointer rejects 0-length views of None; keep at least one element
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a if a.size else np.zeros(1)


class RefIpMatrix:
    """The reference plugin (kind 'SpBKP' or 'RedSpBKP') driven through its own
    virtual interface hqp/Hqp_IpMatrix.h:63-88."""

    def __init__(self, kind="SpBKP", tol=1.0, eps=1e-10):
        lib = _load()
        if lib is None:
            raise RuntimeError(f"oracle/_ref/libhqpref.so not loadable: {_err}")
        self._lib = lib
        self.kind = kind
        self._h = lib.hqpref_create({"SpBKP": 0, "RedSpBKP": 1, "LQDOCP": 2}[kind])
        if not self._h:
            raise RuntimeError("hqpref_create failed (Tcl interpreter?)")
        lib.hqpref_set_params(self._h, tol, eps)
        self.t_init = self.t_factor = self.t_solve = 0.0

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.hqpref_destroy(self._h)
            self._h = None

    def _blocks(self, prog):
        out = []
        for (p, i, x) in (prog.Q, prog.A, prog.C):
            out += [np.ascontiguousarray(p, dtype=np.int32),
                    np.ascontiguousarray(i, dtype=np.int32) if len(i) else np.zeros(1, np.int32),
                    _pad(x)]
        return out

    def init(self, prog):
        self.n, self.me, self.m = prog.dims
        t = C.c_double()
        e = self._lib.hqpref_init(self._h, self.n, self.me, self.m, *self._blocks(prog), C.byref(t))
        self.t_init = t.value
        if e:
            raise RefError(e, "init")

    def update(self, prog):
        e = self._lib.hqpref_update(self._h, *self._blocks(prog))
        if e:
            raise RefError(e, "update")

    def factor(self, z, w):
        t = C.c_double()
        e = self._lib.hqpref_factor(self._h, _pad(z), _pad(w), C.byref(t))
        self.t_factor = t.value
        if e:
            raise RefError(e, "factor")

    def _out(self):
        return [np.zeros(max(k, 1)) for k in (self.n, self.me, self.m, self.m)]

    def _trim(self, d):
        return [d[0][: self.n], d[1][: self.me], d[2][: self.m], d[3][: self.m]]

    def step(self, z, w, r1, r2, r3, r4):
        d = self._out()
        e = self._lib.hqpref_step(self._h, *map(_pad, (z, w, r1, r2, r3, r4)), *d)
        if e:
            raise RefError(e, "step")
        return self._trim(d)

    def solve(self, z, w, r1, r2, r3, r4):
        d = self._out()
        res, t = C.c_double(), C.c_double()
        e = self._lib.hqpref_solve(self._h, *map(_pad, (z, w, r1, r2, r3, r4)), *d, C.byref(res), C.byref(t))
        self.t_solve = t.value
        if e:
            raise RefError(e, "solve")
        return self._trim(d), res.value

    def residuum(self, z, w, r1, r2, r3, r4, dx, dy, dz, dw):
        res = C.c_double()
        e = self._lib.hqpref_residuum(self._h, *map(_pad, (z, w, r1, r2, r3, r4, dx, dy, dz, dw)), C.byref(res))
        if e:
            raise RefError(e, "residuum")
        return res.value

    @property
    def sbw(self):
        return self._lib.hqpref_sbw(self._h)

    @property
    def dim(self):
        return self._lib.hqpref_dim(self._h)

    def perm(self):
        p = np.zeros(self.dim, dtype=np.int32)
        self._lib.hqpref_get_perm(self._h, p)
        return p

    def pivot(self):
        p = np.zeros(self.dim, dtype=np.int32)
        self._lib.hqpref_get_pivot(self._h, p)
        return p

    def matrix(self, which="raw"):
        w = 0 if which == "raw" else 1
        nnz = self._lib.hqpref_matrix_nnz(self._h, w)
        rp = np.zeros(self.dim + 1, dtype=np.int32)
        ci = np.zeros(max(nnz, 1), dtype=np.int32)
        va = np.zeros(max(nnz, 1))
        self._lib.hqpref_get_matrix(self._h, w, rp, ci, va)
        return rp, ci[:nnz], va[:nnz]


# ---------------------------------------------------------------------------
# The reference's own interior-point solvers (oracle/ref_ipdrive.cc).
# host = "ref"  -> oracle/_ref/libhqpref.so       (reference only)
# host = "hip"  -> oracle/_ref/libhqphost_hip.so  (reference + shim/Hqp_IpSpBKPHip.C
#                  + the product's libhqpkkt.so): adds the plugins SpBKPHip / RedSpBKPHip
_HOST_LIBS = {}


def _host(host):
    if host in _HOST_LIBS:
        return _HOST_LIBS[host]
    os.environ.setdefault("TCL_LIBRARY", "/opt/conda/lib/tcl8.6")
    if host == "hip":
        try:
            import torch  # noqa: F401  (same HIP runtime as the product binding, see hqp_amd/_lib.py)
        except ImportError:
            pass
    path = os.path.join(_HERE, "_ref", "libhqpref.so" if host == "ref" else "libhqphost_hip.so")
    lib = C.CDLL(path)
    lib.hqpip_solve.argtypes = ([C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int] + [_ip, _ip, _dp, _dp] * 3
                                + [C.c_double, C.c_int, _dp, _dp, _dp, _dp])
    if hasattr(lib, "hqpsqp_did"):
        lib.hqpsqp_did.restype = C.c_int
        lib.hqpsqp_did.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_double, C.c_int, _dp]
    if hasattr(lib, "hqpip_set_init_method"):
        lib.hqpip_set_init_method.restype = None
        lib.hqpip_set_init_method.argtypes = [C.c_int]
    if hasattr(lib, "hqpip_solve_hot"):
        lib.hqpip_solve_hot.restype = C.c_int
        lib.hqpip_solve_hot.argtypes = ([C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int] + [_ip, _ip, _dp, _dp] * 3
                                        + [_dp, _dp, _dp, C.c_double, C.c_int, _dp, _dp, _dp, _dp])
    _HOST_LIBS[host] = lib
    return lib


def host_available(host="ref"):
    try:
        _host(host)
        return True
    except OSError:
        return False


def ip_solve(prog, solver="Mehrotra", mat_solver="SpBKP", host="ref", qp_eps=1e-10, max_iters=250, init_method=0, qp_mu0=0.0):
    """Run Hqp_IpsMehrotra / Hqp_IpsFranke of the reference on ``prog`` with the KKT
    plugin ``mat_solver``.  Returns dict(x, y, z, iters, result, seconds)."""
    lib = _host(host)
    lib.hqpip_set_init_method(int(init_method))
    if hasattr(lib, "hqpip_set_mu0"):
        lib.hqpip_set_mu0.argtypes = [C.c_double]
        lib.hqpip_set_mu0(float(qp_mu0))
    elif qp_mu0:
        raise RefError(-1, "this oracle/_ref build has no hqpip_set_mu0")
    n, me, m = prog.dims
    args = []
    for (p, i, x), vec in zip((prog.Q, prog.A, prog.C), (prog.c, prog.b, prog.d)):
        args += [np.ascontiguousarray(p, dtype=np.int32),
                 np.ascontiguousarray(i, dtype=np.int32) if len(i) else np.zeros(1, np.int32),
                 _pad(x), _pad(vec)]
    x, y, z = np.zeros(max(n, 1)), np.zeros(max(me, 1)), np.zeros(max(m, 1))
    out = np.zeros(5)
    e = lib.hqpip_solve({"Mehrotra": 0, "Franke": 1, "MehrotraHip": 2, "FrankeHip": 3}[solver], mat_solver.encode(), n, me, m, *args,
                        qp_eps, max_iters, x, y, z, out)
    if e:
        raise RefError(e, f"ip_solve[{solver},{mat_solver}]")
    return dict(x=x[:n], y=y[:me], z=z[:m], iters=int(out[0]), result=int(out[1]), seconds=out[2],
                setup_seconds=out[3], mat_sbw=int(out[4]))


def trace_franke(prog, mat_solver="SpBKP", host="ref", qp_eps=1e-10, max_iters=250):
    """Diagnosis: the reference's Hqp_IpsFranke from a cold start, step by step (oracle/ref_ipdrive.cc, hqpip_trace_franke):
    array (steps, 6) of gap, alpha, alphabar, zeta, rhomin, Hqp_Result after every step."""
    lib = _host(host)
    n, me, m = prog.dims
    args = []
    for (p, i, x), vec in zip((prog.Q, prog.A, prog.C), (prog.c, prog.b, prog.d)):
        args += [np.ascontiguousarray(p, dtype=np.int32),
                 np.ascontiguousarray(i, dtype=np.int32) if len(i) else np.zeros(1, np.int32),
                 _pad(x), _pad(vec)]
    trace = np.zeros((max_iters, 6))
    niter = C.c_int(0)
    f = lib.hqpip_trace_franke
    f.restype = C.c_int
    f.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int] + [np.ctypeslib.ndpointer(dtype=a.dtype, flags="C") for a in args] + \
                 [C.c_double, C.c_int, np.ctypeslib.ndpointer(dtype=np.float64, flags="C"), C.POINTER(C.c_int)]
    e = f(mat_solver.encode(), n, me, m, *args, qp_eps, max_iters, trace, C.byref(niter))
    if e:
        raise RefError(e, f"trace_franke[{mat_solver}]")
    return trace[:niter.value]


def trace_franke_hot(prog, c2, b2, d2, mat_solver="SpBKP", host="ref", qp_eps=1e-10, max_iters=400):
    """Diagnosis: the reference's Hqp_IpsFranke on ``prog`` from a cold start, then hot-started on (c2, b2, d2), step by step
    (oracle/ref_ipdrive.cc, hqpip_trace_franke_hot): (array (steps, 8) of gap, alpha, alphabar, zeta, rhomin, Hqp_Result,
    hot, iter after every step of the SECOND solve; dict(iters, result, first_iters))."""
    lib = _host(host)
    n, me, m = prog.dims
    args = []
    for (p, i, x), vec in zip((prog.Q, prog.A, prog.C), (prog.c, prog.b, prog.d)):
        args += [np.ascontiguousarray(p, dtype=np.int32),
                 np.ascontiguousarray(i, dtype=np.int32) if len(i) else np.zeros(1, np.int32),
                 _pad(x), _pad(vec)]
    args += [_pad(c2), _pad(b2), _pad(d2)]
    trace = np.zeros((max_iters, 8))
    niter = C.c_int(0)
    out = np.zeros(4)
    f = lib.hqpip_trace_franke_hot
    f.restype = C.c_int
    f.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int] + [np.ctypeslib.ndpointer(dtype=a.dtype, flags="C") for a in args] + \
                 [C.c_double, C.c_int, np.ctypeslib.ndpointer(dtype=np.float64, flags="C"), C.POINTER(C.c_int),
                  np.ctypeslib.ndpointer(dtype=np.float64, flags="C")]
    e = f(mat_solver.encode(), n, me, m, *args, qp_eps, max_iters, trace, C.byref(niter), out)
    if e:
        raise RefError(e, f"trace_franke_hot[{mat_solver}]")
    return trace[:niter.value], dict(iters=int(out[0]), result=int(out[1]), first_iters=int(out[2]))


def ip_solve_hot(prog, c2, b2, d2, solver="Mehrotra", mat_solver="SpBKP", host="ref", qp_eps=1e-10, max_iters=250):
    """Two QPs in a row as an SQP iteration makes them: ``prog`` from a cold start, then the same
    matrices with (c2, b2, d2) after update() + hot_start() (hqp/Hqp_IpsMehrotra.C:330-352,
    696-733).  Returns the SECOND solve: dict(x, y, z, iters, result, seconds, first_iters)."""
    lib = _host(host)
    if hasattr(lib, "hqpip_set_mu0"):  # (process-wide in the driver: back to the reference's default)
        lib.hqpip_set_mu0.argtypes = [C.c_double]
        lib.hqpip_set_mu0(0.0)
    n, me, m = prog.dims
    args = []
    for (p, i, x), vec in zip((prog.Q, prog.A, prog.C), (prog.c, prog.b, prog.d)):
        args += [np.ascontiguousarray(p, dtype=np.int32),
                 np.ascontiguousarray(i, dtype=np.int32) if len(i) else np.zeros(1, np.int32),
                 _pad(x), _pad(vec)]
    x, y, z = np.zeros(max(n, 1)), np.zeros(max(me, 1)), np.zeros(max(m, 1))
    out = np.zeros(5)
    e = lib.hqpip_solve_hot({"Mehrotra": 0, "Franke": 1, "MehrotraHip": 2, "FrankeHip": 3}[solver], mat_solver.encode(), n, me, m,
                            *args, _pad(c2), _pad(b2), _pad(d2), qp_eps, max_iters, x, y, z, out)
    if e:
        raise RefError(e, f"ip_solve_hot[{solver},{mat_solver}]")
    return dict(x=x[:n], y=y[:me], z=z[:m], iters=int(out[0]), result=int(out[1]), seconds=out[2],
                first_iters=int(out[4]))


def sqp_did(kmax, qp_solver="Mehrotra", mat_solver="SpBKP", host="ref", sqp_eps=1e-5, sqp_max_iters=100):
    """BASELINE.json configs[0]: the reference's hqp_docp demo (Prg_DID, Hqp_SqpPowell) with the
    QP solver / KKT plugin given by name (oracle/ref_sqpdrive.cc).  Returns dict(f, sqp_iters,
    qp_iters, seconds, norm_inf, norm_grd_L, rc) - rc 0 = optimal."""
    lib = _host(host)
    out = np.zeros(8)
    e = lib.hqpsqp_did(int(kmax), qp_solver.encode(), mat_solver.encode(), float(sqp_eps), int(sqp_max_iters), out)
    if e > 0:
        raise RefError(e, f"sqp_did[{qp_solver},{mat_solver}]")
    return dict(f=out[0], sqp_iters=int(out[1]), qp_iters=int(out[2]), seconds=out[3], norm_inf=out[4],
                norm_grd_L=out[5], rc=e)


def sqp_grid(gx, gy, qp_solver="Mehrotra", mat_solver="RedSpBKP", host="ref", seed=1, eq_every=3, bound_frac=0.5, hela=1,
             ordering=0, sqp_eps=1e-6, sqp_max_iters=200, far=0):
    """BASELINE.json configs[4] stand-in: Prg_GridNLP (oracle/ref_sqpdrive.cc, our program class with the sparsity
    of a discretised control problem) through the reference's Hqp_SqpPowell with the QP solver / KKT plugin given
    by name.  Returns dict(f, sqp_iters, qp_iters, seconds, norm_inf, norm_grd_L, n, me, m, rc) - rc 0 = optimal."""
    lib = _host(host)
    lib.hqpsqp_gridfar.restype = C.c_int
    lib.hqpsqp_gridfar.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_char_p, C.c_char_p, C.c_int,
                                   C.c_double, C.c_int, _dp]
    out = np.zeros(12)
    # (far > 0: that many couplings between distant cells on top of the mesh - the irregular part)
    e = lib.hqpsqp_gridfar(int(gx), int(gy), int(far), int(seed), int(eq_every), float(bound_frac), int(hela), qp_solver.encode(),
                           mat_solver.encode(), int(ordering), float(sqp_eps), int(sqp_max_iters), out)
    if e > 0:
        raise RefError(e, f"sqp_grid[{qp_solver},{mat_solver}]")
    return dict(f=out[0], sqp_iters=int(out[1]), qp_iters=int(out[2]), seconds=out[3], norm_inf=out[4],
                norm_grd_L=out[5], n=int(out[6]), me=int(out[7]), m=int(out[8]), rc=e)


def time_update(prog, mat_solver="SpBKP", host="ref", reps=7):
    """(median seconds of Hqp_Solver::update() with new values on the same pattern, seconds of init +
    first update) with the reference's Hqp_IpsMehrotra and the plugin ``mat_solver``."""
    lib = _host(host)
    out = np.zeros(4)
    args = []
    for (p, i, x) in (prog.Q, prog.A, prog.C):
        args += [np.ascontiguousarray(p, dtype=np.int32),
                 np.ascontiguousarray(i, dtype=np.int32) if len(i) else np.zeros(1, np.int32), _pad(x)]
    lib.hqpip_time_update.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int] + [_ip, _ip, _dp] * 3 + [C.c_int, _dp]
    e = lib.hqpip_time_update(mat_solver.encode(), prog.n, prog.me, prog.m, *args, reps, out)
    if e:
        raise RefError(e, "time_update")
    return float(out[0]), float(out[1])
