"""ctypes binding of the C-ABI library ``libhqpkkt.so`` (include/hqpkkt.h).

There is no CPU fallback: if the HIP extension is missing this module raises,
and every numeric entry point returns ``HQPKKT_E_DEVICE`` without a gfx950 GPU.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HQPKKT_LIB") or os.path.join(_HERE, "libhqpkkt.so")  # override: instrumented builds

OK, E_SIZES, E_MEM, E_SING, E_FORMAT, E_NULL, E_RANGE, E_INTERN, E_DEVICE = 0, 1, 3, 4, 6, 8, 10, 17, 100
MODE_FULL, MODE_REDUCED, MODE_STAGED = 0, 1, 2
LOC_HOST, LOC_DEVICE = 0, 1

# every symbol include/hqpkkt.h declares
SYMBOLS = [
    "hqpkkt_default_opts", "hqpkkt_create", "hqpkkt_destroy", "hqpkkt_analyze",
    "hqpkkt_set_values", "hqpkkt_factor", "hqpkkt_step", "hqpkkt_residual", "hqpkkt_solve",
    "hqpkkt_get_sbw", "hqpkkt_get_perm", "hqpkkt_set_tol", "hqpkkt_set_eps",
    "hqpkkt_set_stream", "hqpkkt_get_stats", "hqpkkt_strerror", "hqpkkt_debug_get",
    "hqpkkt_selftest_mfma", "hqpkkt_set_profile", "hqpkkt_get_profile",
    "hqpkkt_profile_class_name", "hqpkkt_set_shard", "hqpkkt_debug_read",
    "hqpkkt_default_ip_opts", "hqpkkt_mehrotra", "hqpkkt_franke",
    "hqpkkt_set_stages", "hqpkkt_debug_stage_ranks", "hqpkkt_debug_dgemm", "hqpkkt_debug_sk_table",
    "hqpkkt_analyze_staged", "hqpkkt_set_values_staged", "hqpkkt_set_shard_stream",
    "hqpkkt_values_staging", "hqpkkt_detect_stages", "hqpkkt_stage_staging", "hqpkkt_set_stage_block",
    "hqpkkt_debug_factor_block", "hqpkkt_debug_solve_top_stamps",
]
RCCL_LIB_PATH = os.path.join(_HERE, "libhqpkkt_rccl.so")
RCCL_SYMBOLS = ["hqpkkt_rccl_unique_id", "hqpkkt_rccl_create", "hqpkkt_rccl_create_from_env",
                "hqpkkt_rccl_comm_info", "hqpkkt_rccl_exchange", "hqpkkt_rccl_destroy", "hqpkkt_rccl_origin"]

XCHG_ALLGATHER, XCHG_ALLREDUCE_SUM, XCHG_BCAST_BASE = 0, 1, 16
# int fn(void *ctx, int op, double *buf, long long slot_elems, int nslots)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_longlong, C.c_int)


class Opts(C.Structure):
    _fields_ = [("mode", C.c_int), ("device", C.c_int), ("loc", C.c_int), ("tol", C.c_double),
                ("eps", C.c_double), ("pivot_eps", C.c_double), ("leaf_size", C.c_int),
                ("max_pivots", C.c_int), ("zd_policy", C.c_int), ("slack_policy", C.c_int),
                ("no_small_fronts", C.c_int), ("upd_pingpong_mb", C.c_int), ("amalgamation", C.c_int),
                ("ordering", C.c_int)]


class Stats(C.Structure):
    _fields_ = [("dim", C.c_int), ("sbw", C.c_int), ("n_supernodes", C.c_int),
                ("n_levels", C.c_int), ("max_front", C.c_int), ("nnz_kkt", C.c_longlong),
                ("nnz_factor", C.c_longlong), ("flops_factor", C.c_longlong),
                ("bytes_panels", C.c_longlong), ("bytes_updates", C.c_longlong),
                ("n_2x2", C.c_int), ("n_perturbed", C.c_int), ("refine_rounds", C.c_int),
                ("kmax", C.c_double), ("ms_assemble", C.c_float), ("ms_factor", C.c_float),
                ("ms_step", C.c_float), ("ms_residual", C.c_float), ("ms_solve", C.c_float),
                ("shard_rank", C.c_int), ("shard_count", C.c_int), ("n_top", C.c_int),
                ("n_exchange_blocks", C.c_int), ("flops_local", C.c_longlong),
                ("flops_top", C.c_longlong), ("bytes_exchange_factor", C.c_longlong),
                ("bytes_exchange_step", C.c_longlong), ("n_slow_pivots", C.c_int),
                ("n_poll_fallbacks", C.c_int)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class IpOpts(C.Structure):
    _fields_ = [("eps", C.c_double), ("max_iters", C.c_int), ("gammaf", C.c_double),
                ("norm_data", C.c_double), ("hot_start", C.c_int), ("max_warm_iters", C.c_int),
                ("init_method", C.c_int), ("reserved", C.c_int * 1),
                ("norm_Q", C.c_double), ("norm_C", C.c_double), ("norm_d", C.c_double), ("qp_mu0", C.c_double)]


class IpResult(C.Structure):
    _fields_ = [("result", C.c_int), ("iters", C.c_int), ("n_factor", C.c_int), ("n_solve", C.c_int),
                ("gap", C.c_double), ("mu", C.c_double), ("phi", C.c_double), ("pcost", C.c_double),
                ("alpha", C.c_double), ("ms_total", C.c_float), ("attempts", C.c_int)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_lib = None


def lib():
    """Load the HIP extension; fail loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  hqp_amd has no CPU fallback.")
    try:
        # PyTorch is the plumbing for device memory / streams / torch.distributed.  Its
        # wheel bundles its own HIP runtime; load it FIRST so that the extension binds to
        # the same libamdhip64 (two runtimes in one process lose the GPU).
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, ip, dp = C.c_void_p, C.POINTER(C.c_int), C.c_void_p  # vectors go as raw addresses
    L.hqpkkt_default_opts.argtypes = [C.POINTER(Opts)]
    L.hqpkkt_create.argtypes = [C.POINTER(Opts), C.POINTER(vp)]
    L.hqpkkt_destroy.argtypes = [vp]
    L.hqpkkt_analyze.argtypes = [vp, C.c_int, C.c_int, C.c_int] + [vp] * 6 + [ip]
    L.hqpkkt_set_values.argtypes = [vp, dp, dp, dp]
    L.hqpkkt_factor.argtypes = [vp, dp, dp]
    L.hqpkkt_step.argtypes = [vp] + [dp] * 10
    L.hqpkkt_residual.argtypes = [vp] + [dp] * 10 + [C.POINTER(C.c_double)]
    L.hqpkkt_solve.argtypes = [vp] + [dp] * 10 + [C.POINTER(C.c_double)]
    L.hqpkkt_get_sbw.argtypes = [vp, ip]
    L.hqpkkt_get_perm.argtypes = [vp, vp]
    L.hqpkkt_set_tol.argtypes = [vp, C.c_double]
    L.hqpkkt_set_eps.argtypes = [vp, C.c_double]
    L.hqpkkt_set_stream.argtypes = [vp, vp]
    L.hqpkkt_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.hqpkkt_strerror.restype = C.c_char_p
    L.hqpkkt_strerror.argtypes = [C.c_int]
    L.hqpkkt_debug_get.argtypes = [vp, C.c_int, vp, C.POINTER(C.c_longlong)]
    L.hqpkkt_selftest_mfma.argtypes = [C.c_int, C.POINTER(C.c_double)]
    L.hqpkkt_set_profile.argtypes = [vp, C.c_int]
    L.hqpkkt_get_profile.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]
    L.hqpkkt_profile_class_name.restype = C.c_char_p
    L.hqpkkt_profile_class_name.argtypes = [C.c_int]
    L.hqpkkt_set_shard.argtypes = [vp, C.c_int, C.c_int, EXCHANGE_FN, vp]
    L.hqpkkt_debug_read.argtypes = [vp, C.c_int, C.c_int, vp, C.c_longlong, C.POINTER(C.c_longlong)]
    L.hqpkkt_default_ip_opts.argtypes = [C.POINTER(IpOpts)]
    L.hqpkkt_mehrotra.argtypes = [vp, C.POINTER(IpOpts)] + [dp] * 7 + [C.POINTER(IpResult)]
    L.hqpkkt_franke.argtypes = [vp, C.POINTER(IpOpts)] + [dp] * 7 + [C.POINTER(IpResult)]
    L.hqpkkt_set_stages.argtypes = [vp, C.c_int, vp, vp]
    L.hqpkkt_debug_stage_ranks.argtypes = [vp, vp, C.c_int]
    L.hqpkkt_analyze_staged.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int] + [vp] * 6
    L.hqpkkt_set_values_staged.argtypes = [vp, dp, vp, vp, dp, dp]
    L.hqpkkt_values_staging.argtypes = [vp, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    L.hqpkkt_set_shard_stream.argtypes = [vp, C.c_int, C.c_int, vp, vp]
    L.hqpkkt_debug_dgemm.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_double)] * 2
    L.hqpkkt_debug_sk_table.argtypes = [C.c_longlong, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_longlong, C.POINTER(C.c_longlong),
                                        C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.hqpkkt_debug_solve_top_stamps.argtypes = [vp, vp, C.c_int]
    L.hqpkkt_debug_factor_block.argtypes = [C.c_int, C.c_int, vp, C.c_double, C.c_double, C.c_int, C.c_int] + [vp] * 7
    _lib = L
    return L


_rccl = None


def rccl_lib():
    """libhqpkkt_rccl.so (include/hqpkkt_rccl.h): the RCCL transport of a sharded system."""
    global _rccl
    if _rccl is None:
        lib()  # torch (and its RCCL) first
        R = C.CDLL(RCCL_LIB_PATH)
        R.hqpkkt_rccl_unique_id.argtypes = [C.c_char_p]
        R.hqpkkt_rccl_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        R.hqpkkt_rccl_exchange.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p]
        R.hqpkkt_rccl_destroy.argtypes = [C.c_void_p]
        R.hqpkkt_rccl_comm_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        R.hqpkkt_rccl_origin.restype = C.c_char_p
        _rccl = R
    return _rccl


def strerror(code):
    return lib().hqpkkt_strerror(code).decode()
