"""CPU: the C-ABI library loads, exports every symbol of include/hqpkkt.h, the
host-only analysis reproduces the reference's ordering (mat_sbw, _QP2J) and
yields a valid assembly tree; no compute call is made (there is no GPU here)."""
import os
import re

import numpy as np
import pytest

import model
from common import GOLDEN, KINDS, load_golden
from hqp_amd import _lib, ipmatrix, problems

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLS = {"SpBKP": ipmatrix.IpSpBKP, "RedSpBKP": ipmatrix.IpRedSpBKP}


def analyzed(cls, prog, **kw):
    M = cls(**kw)
    try:
        M.init(prog)  # analyze is host-only; update() needs the device
    except ipmatrix.KktError as e:
        assert e.code == _lib.E_DEVICE
    return M


def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "hqpkkt.h")).read()
    declared = set(re.findall(r"\b(hqpkkt_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS)
    L = _lib.lib()
    for s in declared:
        assert getattr(L, s) is not None


def test_rccl_header_symbols_exported():
    """include/hqpkkt_rccl.h against libhqpkkt_rccl.so (the RCCL transport of a sharded system): every
    declared symbol is exported; no communicator is made here (no GPU)."""
    hdr = open(os.path.join(ROOT, "include", "hqpkkt_rccl.h")).read()
    declared = set(re.findall(r"\b(hqpkkt_rccl_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.RCCL_SYMBOLS)
    R = _lib.rccl_lib()
    for s in declared:
        assert getattr(R, s) is not None


def test_no_cpu_fallback_in_product():
    """The product must not import, link or execute anything under oracle/."""
    banned = ("import oracle", "from oracle", "kkt_oracle", "libkktoracle", "libhqpref", "refapi", "oracleapi")
    for dirpath, _d, files in os.walk(os.path.join(ROOT, "hqp_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                for b in banned:
                    assert b not in txt, (f, b)


def test_call_order_and_argument_errors():
    M = ipmatrix.IpSpBKP()
    z = np.ones(3)
    with pytest.raises(ipmatrix.KktError) as e:
        M.factor(None, z[:0], z[:0])
    assert e.value.code == _lib.E_INTERN
    with pytest.raises(ipmatrix.KktError):
        ipmatrix.IpSpBKP(mat_tol=2.0)  # hqp/spBKP.C:389-390 E_RANGE


def test_unsorted_csr_rejected():
    p = problems.banded_qp(20, 3, 1)
    Ai = p.A[1].copy()
    Ai[0], Ai[1] = Ai[1], Ai[0]
    bad = problems.Program(p.n, p.me, p.m, p.Q, (p.A[0], Ai, p.A[2]), p.C)
    with pytest.raises(ipmatrix.KktError) as e:
        ipmatrix.IpSpBKP().init(bad)
    assert e.value.code == _lib.E_FORMAT


@pytest.mark.parametrize("name", GOLDEN)
@pytest.mark.parametrize("kind", KINDS)
def test_ordering_matches_reference(name, kind):
    prog, _st, g = load_golden(name)
    M = analyzed(CLS[kind], prog)
    assert M.mat_sbw == int(g[f"{kind}_sbw"])
    assert np.array_equal(M.perm(), g[f"{kind}_perm"])


@pytest.mark.parametrize("name", ["banded_n300_b10", "did_K50_spread4", "random_n200", "noineq_n120"])
@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("small", [False, True])
def test_assembly_tree_is_valid_and_factorisable(name, kind, small):
    """Numpy model of the kernels' algorithm on the exported structure: nothing
    couples outside the symbolic fronts and the factors solve the scaled system."""
    prog, st, _g = load_golden(name)
    kw = dict(leaf_size=24, max_pivots=12) if small else {}
    M = analyzed(CLS[kind], prog, **kw)
    s = M.structure()
    dim = M.stats()["dim"]
    e = s["elim"]
    assert sorted(e) == list(range(dim))
    assert int(s["npiv"].sum()) == dim
    par = s["parent"]
    assert all(par[k] > k or par[k] < 0 for k in range(len(par)))  # postorder
    mode = 0 if kind == "SpBKP" else 1
    K, _sc = model.scaled_kkt(prog, st[0], st[1], mode)
    mdl = model.Model(s)
    mdl.factor(K, prog.n)
    assert mdl.struct_violation == 0.0
    rhs = np.random.default_rng(0).uniform(-1, 1, dim)
    rhs_e = np.zeros(dim)
    rhs_e[e] = rhs
    x = mdl.solve(rhs_e)[e]
    # one solve, no refinement: tiny blocks may need perturbed pivots
    tol = 1e-3 if mdl.npert else 1e-7
    assert np.abs(K @ x - rhs).max() <= tol * max(1.0, np.abs(K).max() * np.abs(x).max())


def test_large_structure_counts():
    prog = problems.banded_qp(4000, 20, 1)
    M = analyzed(ipmatrix.IpSpBKP, prog)
    st = M.stats()
    assert st["sbw"] == 50 and st["dim"] == 10000
    assert st["n_levels"] <= 12 and st["max_front"] <= 96 + 3 * 50


@pytest.mark.parametrize("case", ["grid", "grid_far", "random", "banded"])
@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("ordering", [1, 2])
def test_graph_dissection_tree_is_valid_and_factorisable(case, kind, ordering):
    """hqpkkt_opts.ordering 1 / 2 (nested dissection of the KKT graph itself, hqp_amd/csrc/analysis.cpp) on the host:
    the exported tree is a postorder, nothing couples outside the symbolic fronts and the numpy model of the kernels'
    algorithm solves the scaled system on it; ordering 1 still reports the reference's mat_sbw (RCM)."""
    prog = {"grid": lambda: problems.grid_sparse_qp(14, 11, seed=2),
            "grid_far": lambda: problems.grid_sparse_qp(12, 12, seed=3, long_range=25),
            "random": lambda: problems.random_sparse_qp(150, 40, 90, row_nnz=3, seed=4),
            "banded": lambda: problems.banded_qp(200, 6, 5)}[case]()
    st = problems.ip_state(prog, 5, 1.0)
    M = analyzed(CLS[kind], prog, ordering=ordering, leaf_size=16)
    s = M.structure()
    dim = M.stats()["dim"]
    e = s["elim"]
    assert sorted(e) == list(range(dim)) and int(s["npiv"].sum()) == dim
    par = s["parent"]
    assert all(par[k] > k or par[k] < 0 for k in range(len(par)))
    if ordering == 1:
        assert M.mat_sbw == analyzed(CLS[kind], prog).mat_sbw
    mode = 0 if kind == "SpBKP" else 1
    K, _sc = model.scaled_kkt(prog, st[0], st[1], mode)
    mdl = model.Model(s)
    mdl.factor(K, prog.n)
    assert mdl.struct_violation == 0.0
    rhs = np.random.default_rng(1).uniform(-1, 1, dim)
    rhs_e = np.zeros(dim)
    rhs_e[e] = rhs
    x = mdl.solve(rhs_e)[e]
    tol = 1e-3 if mdl.npert else 1e-7
    assert np.abs(K @ x - rhs).max() <= tol * max(1.0, np.abs(K).max() * np.abs(x).max())


def test_dissection_of_a_band_with_far_couplings():
    """Host-only (hqpkkt_analyze touches no device): the symbolic phase on the irregular stand-in of BASELINE configs[4] -
    10^5 variables, 21 entries per row of Q, 1000 couplings between distant variables - through the dissection of the
    graph itself (ordering 2).  Level structures alone gave a root separator of 6154 vertices and fronts of 11 724 rows
    (round 4); with the cuts by linear order among the candidates the largest separator has ~500 vertices
    (max_front <= 4000)."""
    from hqp_amd import ipmatrix, problems
    prog = problems.banded_long_range_qp(100000, 10, 1000)
    M = ipmatrix.IpRedSpBKP(ordering=2)
    try:
        M.init(prog)
    except ipmatrix.KktError as e:  # no device here: the analysis has run, the upload of the values fails
        assert e.code == 100
    s = M.stats()
    assert s["dim"] == 150000
    assert s["max_front"] <= 4000 and s["flops_factor"] < 5e10, s


def test_large_separators_are_cut_into_longer_pieces():
    """Host-only: a separator of >= 768 vertices (the top of an irregular graph's tree: a 200 x 200 mesh with 1500 far
    couplings) is cut into pieces of up to 192 pivots when the caller leaves max_pivots at its default - every piece
    rewrites the whole update block of its front - and into pieces of <= 160 when the caller asks for 160; the tree stays
    a valid postorder either way.  (On the GPU: tests/test_gpu_ordering.py, the 10^6-cell system and
    test_update_blocks_on_128_tiles_against_the_oracle.)"""
    from hqp_amd import ipmatrix, problems
    prog = problems.grid_sparse_qp(200, 200, seed=5, long_range=1500)
    got = {}
    for name, kw in (("default", {}), ("160", dict(max_pivots=160))):
        M = ipmatrix.IpRedSpBKP(ordering=2, **kw)
        try:
            M.init(prog)
        except ipmatrix.KktError as e:  # no device here: the analysis has run, the upload of the values fails
            assert e.code == 100
        s = M.structure()
        par, npiv = s["parent"], s["npiv"]
        assert int(npiv.sum()) == M.stats()["dim"] and all(par[k] > k or par[k] < 0 for k in range(len(par)))
        got[name] = npiv
    assert 160 < int(got["default"].max()) <= 192
    assert int(got["160"].max()) <= 160
    # only the long separators change: the supernodes of <= 160 pivots are the same multiset apart from the re-cut chains
    assert abs(len(got["default"]) - len(got["160"])) <= int((got["default"] > 160).sum()) + 4
