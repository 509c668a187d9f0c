"""Randomised sweep of the device-resident interior-point loops (hqpkkt_mehrotra / hqpkkt_franke)
against the reference's own Hqp_IpsMehrotra / Hqp_IpsFranke with its own plugin (oracle/_ref):
random QPs of the three structured families, plugin kinds, starting points.  Same termination,
iteration counts within the tolerance of tests/test_reference_host.py, same objective.
Usage: python tools/fuzz_ip.py [cases] [seed0]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hqp_amd import problems, ipmatrix
from oracle import refapi


def objective(prog, x):
    p, i, v = prog.Q
    rows = np.repeat(np.arange(prog.n), np.diff(p))
    return float((np.where(rows == i, 0.5, 1.0) * v * x[rows] * x[i]).sum() + prog.c @ x)


HOT = os.environ.get("FUZZ_HOT", "") == "1"  # 1: hot starts (two QPs in a row) instead of cold starts
SHIM = os.environ.get("FUZZ_SHIM", "") == "1"  # 1: Hqp_IpsMehrotra / Hqp_IpsFranke + SpBKPHip / RedSpBKPHip instead of the device loops
OPTS = eval(os.environ.get("FUZZ_OPTS", "{}"))  # plugin options for every case, e.g. "dict(slack_policy=1)"


def check(case, hot=None):
    """-> (status, line) of fuzz case number ``case`` (hot: two QPs in a row, the second hot-started; default: FUZZ_HOT): 'ok', 'odd' (explained: the reference itself is not optimal / the
    final test missed by a hair on one side / counts spread like the reference's own two plugins) or 'BAD'"""
    HOT = globals()["HOT"] if hot is None else bool(hot)
    rng = np.random.default_rng(5000 + case)
    what = str(rng.choice(["banded", "did", "docp"]))
    if what == "banded":
        b = int(rng.integers(1, 30))
        args = (int(rng.integers(2 * b + 2, 1200)), b, int(rng.integers(1, 1000)))
        prog = problems.banded_qp(*args)
    elif what == "did":
        args = (int(rng.integers(2, 800)), float(rng.choice([1e-4, 1e-2, 1.0])))
        prog = problems.did_like_qp(*args)
    else:
        args = (int(rng.integers(2, 60)), int(rng.integers(1, 10)), int(rng.integers(1, 5)), int(rng.integers(1, 99)))
        prog = problems.lq_docp(*args)
    kind = str(rng.choice(["SpBKP", "RedSpBKP"]))
    solver = str(rng.choice(["Mehrotra", "Mehrotra", "Franke"]))
    im = int(rng.integers(0, 4)) if solver == "Mehrotra" else 0
    tag = f"case {case}: {what}{args} {solver} {kind} init {im}"
    try:
        if HOT:  # a second QP with the same matrices and a perturbed c, hot-started from the first
            if solver == "Mehrotra":
                im = 0
            scale = float(rng.choice([1e-4, 1e-3, 1e-2, 1e-1]))
            c2 = prog.c + scale * rng.standard_normal(prog.n) * (np.abs(prog.c).max() + 1)
            tag += f" hot {scale}"
            ref = refapi.ip_solve_hot(prog, c2, prog.b, prog.d, solver, kind, max_iters=400)
            M = (ipmatrix.IpRedSpBKP if kind == "RedSpBKP" else ipmatrix.IpSpBKP)(**OPTS)
            M.init(prog)
            prog2 = problems.Program(prog.n, prog.me, prog.m, prog.Q, prog.A, prog.C, c=c2, b=prog.b, d=prog.d)
            if solver == "Mehrotra":
                M.mehrotra(prog, max_iters=400, hot_start=2)
                x, y, z, w, info = M.mehrotra(prog2, max_iters=400, hot_start=1)
            else:
                M.franke(prog, max_iters=400)
                x, y, z, w, info = M.franke(prog2, max_iters=400, hot_start=1)
            prog = prog2
        else:
            ref = refapi.ip_solve(prog, solver, kind, init_method=im)
        if HOT:
            pass
        elif SHIM:  # the reference's own IP solver driving the plugin through shim/ (oracle/_ref/libhqphost_hip.so)
            hip = refapi.ip_solve(prog, solver, kind + "Hip", host="hip", init_method=im)
            x, info = hip["x"], dict(result=hip["result"], iters=hip["iters"])
        elif solver == "Mehrotra":
            M = (ipmatrix.IpRedSpBKP if kind == "RedSpBKP" else ipmatrix.IpSpBKP)(**OPTS)
            M.init(prog)
            x, y, z, w, info = M.mehrotra(prog, max_iters=250, init_method=im)
        else:
            M = (ipmatrix.IpRedSpBKP if kind == "RedSpBKP" else ipmatrix.IpSpBKP)(**OPTS)
            M.init(prog)
            x, y, z, w, info = M.franke(prog, max_iters=250)
    except Exception as e:
        return "BAD", f"EXCEPTION {tag} {repr(e)[:120]}"
    fr, fd = objective(prog, ref["x"]), objective(prog, x)
    same_f = abs(fr - fd) <= 1e-6 * max(1.0, abs(fr))
    slack = 2 if solver == "Mehrotra" else max(2, ref["iters"] // 10)
    same_it = abs(info["iters"] - ref["iters"]) <= slack
    line = f"{tag}: result {info['result']}/{ref['result']} iters {info['iters']}/{ref['iters']} f {fd:.10g}/{fr:.10g}"
    if info["result"] == ref["result"] and same_it and (same_f or ref["result"] != 0):
        return "ok", line
    if ref["result"] != 0 and info["result"] in (0, 3, 4):
        # the reference does not end "optimal" itself: stall / E_SING next to the solution
        return "odd", "reference not optimal: " + line
    if {info["result"], ref["result"]} == {0, 3} and same_it and same_f:
        return "odd", "final test missed by a hair on one side: " + line
    # how far apart are the reference's own two plugins on this QP?
    if HOT:
        # A hot start begins at the END POINT of the solve before: z and w of the active / inactive rows at 1e-10 .. 1e-13,
        # where the last bits of that solve decide which component blocks the first step and whether the gap rises above
        # its first value (Hqp_IpsFranke::solve restarts cold then, hqp/Hqp_IpsFranke.C:388-397).  The reference's own two
        # plugins end their first solves at different such points as well: where THEY part ways on the hot start by as
        # much as the device loop does, the difference is that sensitivity, not a rule (round 6; traces of the finds:
        # tools/franke_hot_trace.py, profiles/r06_franke_hot_traces.txt)
        other = refapi.ip_solve_hot(prog, c2, prog.b, prog.d, solver, "RedSpBKP" if kind == "SpBKP" else "SpBKP", max_iters=400)
        if info["result"] == ref["result"] == other["result"] and same_f and \
                abs(info["iters"] - ref["iters"]) <= 2 * abs(other["iters"] - ref["iters"]) + slack:
            return "odd", f"hot-start counts spread like the reference's own plugins ({other['iters']} with the other one, first solves {ref['first_iters']} / {other['first_iters']}): " + line
        return "BAD", f"MISMATCH {line} (reference with its other plugin: {other['result']}, {other['iters']}; first solves {ref['first_iters']} / {other['first_iters']})"
    other = refapi.ip_solve(prog, solver, "RedSpBKP" if kind == "SpBKP" else "SpBKP", init_method=im)
    if info["result"] == ref["result"] == other["result"] and same_f and \
            abs(info["iters"] - ref["iters"]) <= 2 * abs(other["iters"] - ref["iters"]) + slack:
        return "odd", f"counts spread like the reference's own plugins ({other['iters']} with the other one): " + line
    return "BAD", f"MISMATCH {line} (reference with its other plugin: {other['result']}, {other['iters']})"


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = odd = 0
    t0 = time.time()
    for case in range(seed0, seed0 + ncases):
        status, line = check(case)
        bad += status == "BAD"
        odd += status == "odd"
        if status != "ok":
            print(line, flush=True)
    print(f"{ncases} cases from {seed0}: {bad} bad, {odd} odd, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
