"""CPU: the stand-in for BASELINE.json configs[4] (Prg_GridNLP, oracle/ref_sqpdrive.cc) through the reference's own
Hqp_SqpPowell + Hqp_IpsMehrotra + RedSpBKP / SpBKP - pins the harness the GPU test compares our plugin with."""
import pytest

from oracle import refapi

needs_ref = pytest.mark.skipif(not refapi.host_available("ref"), reason="oracle/_ref not built / loadable here")


@needs_ref
@pytest.mark.parametrize("mat", ["RedSpBKP", "SpBKP"])
def test_reference_solves_the_grid_nlp(mat):
    r = refapi.sqp_grid(12, 12, "Mehrotra", mat)
    assert r["rc"] == 0 and (r["n"], r["me"], r["m"]) == (144, 40, 156)
    assert r["sqp_iters"] == 10 and r["qp_iters"] == 23
    assert abs(r["f"] - 22.561897391459546) < 1e-9
    assert r["norm_inf"] < 1e-10 and r["norm_grd_L"] < 1e-6


@needs_ref
def test_both_hessian_modes_reach_the_same_point():
    a = refapi.sqp_grid(12, 12, "Mehrotra", "RedSpBKP", hela=1)
    b = refapi.sqp_grid(12, 12, "Mehrotra", "RedSpBKP", hela=0)
    assert a["rc"] == 0 and b["rc"] == 0 and abs(a["f"] - b["f"]) < 1e-8
    assert b["sqp_iters"] > a["sqp_iters"]  # the diagonal quasi-Newton scaling needs many more steps
