"""CPU, build container only: pins the oracle against the reference compiled from
/root/reference (oracle/_ref/libhqpref.so) on fresh seeded inputs."""
import numpy as np
import pytest

from common import KINDS, rel_err
from hqp_amd import problems
from oracle import oracleapi, refapi

pytestmark = pytest.mark.skipif(not refapi.available(), reason="oracle/_ref not built / loadable here")

CASES = [
    (lambda: problems.banded_qp(500, 12, 21), 21, 0.0),
    (lambda: problems.banded_qp(240, 8, 22), 22, 5.0),
    (lambda: problems.did_like_qp(120), 23, 3.0),
    (lambda: problems.random_sparse_qp(160, 50, 90, seed=24), 24, 1.0),
]


@pytest.mark.parametrize("case", range(len(CASES)))
@pytest.mark.parametrize("kind", KINDS)
def test_oracle_equals_reference(case, kind):
    mk, seed, spread = CASES[case]
    prog = mk()
    st = problems.ip_state(prog, seed, spread)
    R, O = refapi.RefIpMatrix(kind), oracleapi.OracleIpMatrix(kind)
    R.init(prog), O.init(prog)
    assert R.sbw == O.sbw
    assert np.array_equal(R.perm(), O.perm())
    R.factor(st[0], st[1]), O.factor(st[0], st[1])
    assert np.array_equal(R.pivot(), O.pivot())
    rp, ci, va = R.matrix("fac")
    D = O.dense("fac")
    rows = np.repeat(np.arange(R.dim), np.diff(rp))
    assert np.abs(D[rows, ci] - va).max() <= 1e-12 * max(1.0, np.abs(va).max())
    (dr, rr), (do, ro) = R.solve(*st), O.solve(*st)
    assert rel_err(do, dr) <= 1e-12
    assert abs(rr - ro) <= 1e-13 * max(1.0, rr)


@pytest.mark.parametrize("K,spread", [(50, 0.0), (50, 3.0), (200, 1.0)])
def test_reference_lqdocp_solves_the_same_system(K, spread):
    """Hqp_IpLQDOCP (multistage Riccati, hqp/Hqp_IpLQDOCP.C) and Hqp_IpSpBKP are two
    algorithms for one KKT system: on a DOCP-structured QP they agree, which is what
    lets the full-system engine stand in for LQDOCP."""
    prog = problems.did_like_qp(K)
    st = problems.ip_state(prog, 11, spread)
    L, O = refapi.RefIpMatrix("LQDOCP"), oracleapi.OracleIpMatrix("SpBKP")
    L.init(prog), O.init(prog)
    L.factor(st[0], st[1]), O.factor(st[0], st[1])
    (dl, rl), (do, ro) = L.solve(*st), O.solve(*st)
    assert rl <= 1e-8 and ro <= 1e-8
    assert rel_err(dl, do) <= 1e-6
