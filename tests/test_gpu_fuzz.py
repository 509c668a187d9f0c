"""GPU (-m gpu): deterministic subsets of the randomised campaigns under tools/ (fixed seeds, the case generators and
checks of the campaigns themselves), so that the driver's GPU test run repeats them - the large campaigns
(profiles/r0*_fuzz_*.txt) are builder-run.  About half a minute in total:

* tools/fuzz.py          tree engine (both plugins, tree options, update()) against the CPU oracle, 300 + 40 cases
* tools/fuzz_staged.py   STAGED engine against the reference's own Hqp_IpLQDOCP, 200 cases + the historic finds
* tools/fuzz_bigstage.py STAGED engine on stages of 10 ... 300 controls against the tree engine, 100 cases
* tools/fuzz_ip.py       the device-resident Mehrotra / Franke loops against the reference's solvers, 100 cases + the
                         finds of rounds 1-5 (those that still differ are listed as such, not hidden)
"""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

from oracle import refapi  # noqa: E402

pytestmark = pytest.mark.gpu


def _run(check, cases):
    bad, cnt = [], {}
    for case in cases:
        status, detail = check(case)
        cnt[status] = cnt.get(status, 0) + 1
        if status == "BAD":
            bad.append(detail)
    return bad, cnt


def test_tree_engine_campaign_subset():
    """tools/fuzz.py cases 0-299: random banded / DID-like / multistage / unstructured systems, both plugins, tree
    options, w/z spreads; mat_sbw and the RCM permutation equal to the oracle's, the oracle's residual of OUR solution
    within 1e-10 of its own, again after update() - and systems the reference does not solve are not solved here either."""
    import fuzz
    bad, cnt = _run(fuzz.check, range(300))
    assert not bad, bad
    assert cnt.get("ok", 0) >= 270, cnt  # (a few per hundred are singular on both sides)


@pytest.mark.parametrize("ordering", [1, 2])
def test_tree_engine_campaign_subset_with_graph_dissection(monkeypatch, ordering):
    """... 40 cases through the tree of the graph's own dissection (hqpkkt_opts.ordering 1; 2: without the reference's RCM
    pass - no mat_sbw / permutation to compare there)"""
    import fuzz
    monkeypatch.setenv("FUZZ_ORDERING", str(ordering))
    bad, cnt = _run(fuzz.check, range(3000, 3040))
    assert not bad, bad
    assert cnt.get("ok", 0) >= 34, cnt


def test_staged_engine_campaign_subset_against_the_reference():
    """tools/fuzz_staged.py cases 0-199 and the finds of round 2 (5597, 7983: free initial states whose path
    equalities consume the controls) and of the first sweep (672: a stiff stage) against the REFERENCE's Hqp_IpLQDOCP
    (oracle/_ref): the oracle's residual of our refined solution within 1e-10 of the reference's."""
    import fuzz_staged
    if not refapi.available():
        pytest.skip("oracle/_ref not present")
    bad, cnt = _run(fuzz_staged.check, list(range(200)) + [672, 5597, 7983])
    assert not bad, bad
    assert cnt.get("ok", 0) >= 180, cnt  # (the rest: the reference itself fails or ends above 1e-8)


def test_large_stage_campaign_subset():
    """tools/fuzz_bigstage.py cases 0-99: stages of 10 ... 300 controls (K in registers, in LDS, by the blocked
    elimination), carried final-state rows, free initial states - against the tree engine on the same QP."""
    import fuzz_bigstage
    tally = {}
    bad, cnt = _run(lambda c: fuzz_bigstage.check(c, tally), range(100))
    assert not bad, bad
    assert cnt.get("ok", 0) >= 95, cnt
    assert tally["blocked"] >= 100 and tally["fell"] <= 3, tally  # the blocked elimination ran, and did not fall back


# finds of tools/fuzz_ip.py: rounds 1-4 (profiles/r02_fuzz_big.txt: 12 000 cases, r04_fuzz_tree.txt: 6 000) - the first ten,
# all cured in round 5 (cancelled multiplier pivots replaced, up to fifteen refinement rounds behind a perturbed pivot) -
# and those that round 5's campaigns of 12 000 found instead (profiles/r05_fuzz_tree.txt: eight; three on the final code,
# profiles/r05_fuzz_ip_final.txt: 2536, 6258, 8650).  Those that differ from the
# reference are reported as expected failures, not hidden.  Their cause (shown by experiment in round 5):
# ONE blocking decision of Hqp_IpsFranke's step-length rule near the solution, on the double-integrator structure, flips
# with the rounding of the loop's own vector kernels (contracted multiply-adds where the reference's host code rounds
# twice); with the kernels compiled without contraction these three agree and four other QPs of the 12 000 differ instead.
# Round 6 (profiles/r06_fuzz_final.txt, 8000 QPs on the final code): 2536 and 8650 agree since the small fronts sum their update
# blocks in another order (another rounding: the cause named above), 6258 stays; 187 - the loop ended "degenerate" one step
# before the reference's "optimal" on an EXACTLY zero pivot (w / z of 1e-21 beside 1e+9) - agrees since the loops' second
# attempt replaces such a pivot like any other cancelled one where the run's first factorisation met none (kernels.hip.h).
IP_FINDS = [193, 2536, 5533, 5975, 6258, 7018, 7260, 7511, 8650, 10246, 803, 2419, 2532, 3015, 5466, 5921, 5954, 7818, 10258, 187]


def test_ip_loop_campaign_subset():
    """tools/fuzz_ip.py cases 0-99: hqpkkt_mehrotra / hqpkkt_franke against the reference's
    Hqp_IpsMehrotra / Hqp_IpsFranke with its own plugin - same result code, objective to 1e-6, iteration counts
    within 2 (Mehrotra) / 10 % (Franke), or spread as the reference's own two plugins are."""
    import fuzz_ip
    if not refapi.host_available("ref"):
        pytest.skip("oracle/_ref not present")
    bad, cnt = _run(fuzz_ip.check, range(100))
    assert not bad, bad
    assert cnt.get("ok", 0) >= 95, cnt


@pytest.mark.parametrize("case", IP_FINDS)
def test_ip_loop_finds_of_the_campaigns(case):
    """The QPs on which the campaigns of rounds 1-4 found hqpkkt_franke / hqpkkt_mehrotra to differ from the
    reference's solvers (iteration counts apart by more than 10 %, or "degenerate" within 3 iterations of the
    reference's "optimal"): pass where they now agree, expected failure where they still differ."""
    import fuzz_ip
    if not refapi.host_available("ref"):
        pytest.skip("oracle/_ref not present")
    status, line = fuzz_ip.check(case)
    if status == "BAD":
        pytest.xfail("known difference (one blocking decision near the solution flips with the rounding of the loop's own vector "
                     "kernels, DESIGN.md section 6): " + line)


def test_ip_loop_hot_start_subset():
    """tools/fuzz_ip.py with hot starts (FUZZ_HOT=1), cases 0-99: two QPs in a row with the same matrices, the second
    hot-started from the first as Hqp_SqpSolver does on every SQP iteration after the first - same result code and
    objective as the reference's solver on the second QP, iteration counts within the campaign's bounds or apart like
    those of the reference's own two plugins (a hot start begins at the end point of the solve before, where the last bits
    decide about the first blocking component and a cold restart: profiles/r06_franke_hot_traces.txt)."""
    import fuzz_ip
    if not refapi.host_available("ref"):
        pytest.skip("oracle/_ref not present")
    bad, cnt = _run(lambda c: fuzz_ip.check(c, hot=True), range(100))
    assert not bad, bad
    assert cnt.get("ok", 0) >= 90, cnt


@pytest.mark.parametrize("case", [444, 1075])
def test_ip_loop_hot_start_find_of_round_6(case):
    """The QPs of the hot-start campaigns of round 6 (profiles/r06_fuzz_hot.txt: 800 cases; r06_fuzz_final.txt: 2000) on
    which the device loop is further from the reference than the reference's own plugins are from each other: cases 444
    and 1075, Franke on the double-integrator structure - the hot start needs more than the 15 warm iterations
    (qp_max_warm_iters) the reference converges in (at the 15th), so the loop restarts cold: 201 / 116 against 15
    iterations, same solution."""
    import fuzz_ip
    if not refapi.host_available("ref"):
        pytest.skip("oracle/_ref not present")
    status, line = fuzz_ip.check(case, hot=True)
    if status == "BAD":
        pytest.xfail("known difference (the hot start misses qp_max_warm_iters by its first, short steps: "
                     "profiles/r06_franke_hot_traces.txt): " + line)
