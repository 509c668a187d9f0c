"""TEST INFRASTRUCTURE ONLY -- ctypes binding of oracle/libkktoracle.so, the
plain-C restatement (oracle/kkt_oracle.c) of the reference's KKT path."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libkktoracle.so")
_lib = None

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")

E_SING = 4


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libkktoracle.so"])


def _load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_PATH):
        build()
    lib = C.CDLL(_PATH)
    lib.kkto_create.restype = C.c_void_p
    lib.kkto_create.argtypes = [C.c_int, C.c_double, C.c_double]
    lib.kkto_destroy.argtypes = [C.c_void_p]
    lib.kkto_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int] + [_ip, _ip, _dp] * 3
    lib.kkto_update.argtypes = [C.c_void_p, _dp, _dp, _dp]
    lib.kkto_factor.argtypes = [C.c_void_p, _dp, _dp]
    lib.kkto_step.argtypes = [C.c_void_p] + [_dp] * 10
    lib.kkto_residuum.restype = C.c_double
    lib.kkto_residuum.argtypes = [C.c_void_p] + [_dp] * 10
    lib.kkto_solve.argtypes = [C.c_void_p] + [_dp] * 10 + [C.POINTER(C.c_double), C.POINTER(C.c_int)]
    lib.kkto_sbw.argtypes = [C.c_void_p]
    lib.kkto_dim.argtypes = [C.c_void_p]
    lib.kkto_get_perm.argtypes = [C.c_void_p, _ip]
    lib.kkto_get_pivot.argtypes = [C.c_void_p, _ip]
    lib.kkto_get_dense.argtypes = [C.c_void_p, C.c_int, _dp]
    _lib = lib
    return lib


class OracleError(RuntimeError):
    def __init__(self, code, where):
        super().__init__(f"oracle status {code} in {where}")
        self.code = code


def _pad(a, dt=np.float64):
    a = np.ascontiguousarray(a, dtype=dt)
    return a if a.size else np.zeros(1, dtype=dt)


class OracleIpMatrix:
    """Same call sequence as the reference plugin (hqp/Hqp_IpMatrix.h:63-88):
    init -> update -> factor -> step / solve / residuum."""

    def __init__(self, kind="SpBKP", tol=1.0, eps=1e-10):
        self._lib = _load()
        self.kind = kind
        self._h = self._lib.kkto_create({"SpBKP": 0, "RedSpBKP": 1}[kind], tol, eps)

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.kkto_destroy(self._h)
            self._h = None

    def init(self, prog):
        self.n, self.me, self.m = prog.dims
        args = []
        for (p, i, x) in (prog.Q, prog.A, prog.C):
            args += [_pad(p, np.int32), _pad(i, np.int32), _pad(x)]
        e = self._lib.kkto_init(self._h, self.n, self.me, self.m, *args)
        if e:
            raise OracleError(e, "init")

    def update(self, prog):
        e = self._lib.kkto_update(self._h, _pad(prog.Q[2]), _pad(prog.A[2]), _pad(prog.C[2]))
        if e:
            raise OracleError(e, "update")

    def factor(self, z, w):
        e = self._lib.kkto_factor(self._h, _pad(z), _pad(w))
        if e:
            raise OracleError(e, "factor")

    def _out(self):
        return [np.zeros(max(k, 1)) for k in (self.n, self.me, self.m, self.m)]

    def _trim(self, d):
        return [d[0][: self.n], d[1][: self.me], d[2][: self.m], d[3][: self.m]]

    def step(self, z, w, r1, r2, r3, r4):
        d = self._out()
        e = self._lib.kkto_step(self._h, *map(_pad, (z, w, r1, r2, r3, r4)), *d)
        if e:
            raise OracleError(e, "step")
        return self._trim(d)

    def solve(self, z, w, r1, r2, r3, r4):
        d = self._out()
        res, rounds = C.c_double(), C.c_int()
        e = self._lib.kkto_solve(self._h, *map(_pad, (z, w, r1, r2, r3, r4)), *d,
                                 C.byref(res), C.byref(rounds))
        if e:
            raise OracleError(e, "solve")
        self.rounds = rounds.value
        return self._trim(d), res.value

    def residuum(self, z, w, r1, r2, r3, r4, dx, dy, dz, dw):
        return self._lib.kkto_residuum(self._h, *map(_pad, (z, w, r1, r2, r3, r4, dx, dy, dz, dw)))

    @property
    def sbw(self):
        return self._lib.kkto_sbw(self._h)

    @property
    def dim(self):
        return self._lib.kkto_dim(self._h)

    def perm(self):
        p = np.zeros(self.dim, dtype=np.int32)
        self._lib.kkto_get_perm(self._h, p)
        return p

    def pivot(self):
        p = np.zeros(self.dim, dtype=np.int32)
        self._lib.kkto_get_pivot(self._h, p)
        return p

    def dense(self, which="raw"):
        a = np.zeros((self.dim, self.dim))
        self._lib.kkto_get_dense(self._h, 0 if which == "raw" else 1, a)
        return a
