"""CPU: the plain-C oracle (oracle/kkt_oracle.c) against the golden vectors the
reference itself produced (tests/golden/make_golden.py)."""
import numpy as np
import pytest

from common import GOLDEN, KINDS, load_golden, rel_err
from oracle import oracleapi


@pytest.mark.parametrize("name", GOLDEN)
@pytest.mark.parametrize("kind", KINDS)
def test_oracle_matches_reference_outputs(name, kind):
    prog, st, g = load_golden(name)
    O = oracleapi.OracleIpMatrix(kind)
    O.init(prog)
    assert O.sbw == int(g[f"{kind}_sbw"])
    assert np.array_equal(O.perm(), g[f"{kind}_perm"])
    O.factor(st[0], st[1])
    assert np.array_equal(O.pivot(), g[f"{kind}_pivot"])
    step = O.step(*st)
    gold_step = [g[f"{kind}_step_{k}"] for k in ("dx", "dy", "dz", "dw")]
    # same algorithm, same operation order: agreement to the last few ulps
    assert rel_err(step, gold_step) <= 1e-13
    sol, res = O.solve(*st)
    gold = [g[f"{kind}_solve_{k}"] for k in ("dx", "dy", "dz", "dw")]
    assert rel_err(sol, gold) <= 1e-13
    assert abs(res - float(g[f"{kind}_res"])) <= 1e-13 * max(1.0, float(g[f"{kind}_res"]))
    assert abs(O.residuum(*st, *step) - float(g[f"{kind}_res_of_step"])) <= 1e-14


def test_oracle_singular_zero_slack():
    prog, st, _ = load_golden("banded_n60_b4")
    O = oracleapi.OracleIpMatrix("SpBKP")
    O.init(prog)
    z = st[0].copy()
    z[3] = 0.0
    with pytest.raises(oracleapi.OracleError) as e:
        O.factor(z, st[1])
    assert e.value.code == oracleapi.E_SING


def test_oracle_update_changes_values():
    prog, st, _ = load_golden("banded_n300_b10")
    O = oracleapi.OracleIpMatrix("SpBKP")
    O.init(prog)
    O.factor(st[0], st[1])
    a, _ = O.solve(*st)
    prog.Q = (prog.Q[0], prog.Q[1], prog.Q[2] * 2.0)
    O.update(prog)
    O.factor(st[0], st[1])
    b, res = O.solve(*st)
    assert res < 1e-10 and rel_err(a, b) > 1e-3


@pytest.mark.parametrize("name", __import__("common").GOLDEN_LQDOCP)
def test_oracle_solves_the_lqdocp_fixtures_like_the_reference(name):
    """The multistage fixtures (results of the reference's Hqp_IpLQDOCP): the CPU oracle of the reduced system
    reaches the same solution - two different eliminations of the same KKT system."""
    from common import GOLDEN_LQDOCP_DIR, load_golden, rel_err
    prog, st, g = load_golden(name, GOLDEN_LQDOCP_DIR)
    O = oracleapi.OracleIpMatrix("RedSpBKP")
    O.init(prog)
    O.factor(st[0], st[1])
    sol, res = O.solve(*st)
    gold = [g[f"LQDOCP_solve_{k}"] for k in ("dx", "dy", "dz", "dw")]
    scale = max(1.0, max(np.abs(v).max() for v in gold if len(v)))
    assert res <= 1e-10 * scale
    assert rel_err(sol, gold) <= 1e-7, rel_err(sol, gold)
