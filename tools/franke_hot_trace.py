"""Diagnosis of a find of the hot-start campaign (FUZZ_HOT=1 tools/fuzz_ip.py): the reference's hot-started Hqp_IpsFranke
(probe subclass, oracle/ref_ipdrive.cc: hqpip_trace_franke_hot) and the device loop (HQPKKT_TRACE_IP) step by step.
   python tools/franke_hot_trace.py CASE [CASE ...]"""
import os, sys, re, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hqp_amd import problems


def hot_case(case):
    """the QP pair of FUZZ_HOT case `case` (the generator of tools/fuzz_ip.py, same random stream)"""
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
    _im = int(rng.integers(0, 4)) if solver == "Mehrotra" else 0
    scale = float(rng.choice([1e-4, 1e-3, 1e-2, 1e-1]))
    c2 = prog.c + scale * rng.standard_normal(prog.n) * (np.abs(prog.c).max() + 1)
    return prog, c2, solver, kind, f"{what}{args} {solver} {kind} hot {scale}"


if len(sys.argv) > 2 and sys.argv[2] == "child":  # the device loop with its trace on stderr
    from hqp_amd import ipmatrix
    prog, c2, solver, kind, tag = hot_case(int(sys.argv[1]))
    M = (ipmatrix.IpRedSpBKP if kind == "RedSpBKP" else ipmatrix.IpSpBKP)()
    M.init(prog)
    M.franke(prog, max_iters=400)
    print("franke: SECOND", file=sys.stderr, flush=True)
    prog2 = problems.Program(prog.n, prog.me, prog.m, prog.Q, prog.A, prog.C, c=c2, b=prog.b, d=prog.d)
    x, y, z, w, info = M.franke(prog2, max_iters=400, hot_start=1)
    print("franke: END iters", info["iters"], "result", info["result"], file=sys.stderr)
    sys.exit(0)

from oracle import refapi
for case in [int(a) for a in sys.argv[1:]]:
    prog, c2, solver, kind, tag = hot_case(case)
    assert solver == "Franke", tag
    a, ia = refapi.trace_franke_hot(prog, c2, prog.b, prog.d, kind)
    env = dict(os.environ, HQPKKT_TRACE_IP="1")
    out = subprocess.run([sys.executable, __file__, str(case), "child"], env=env, capture_output=True, text=True).stderr
    lines = out.splitlines()
    k2 = [i for i, l in enumerate(lines) if l.startswith("franke: SECOND")][0]
    first = [l for l in lines[:k2] if l.startswith("franke: step")]
    c = np.array([[float(v) for v in re.findall(r"(?:gap|alpha|alphabar|zeta|rhomin|resid|hot) (\S+)", l)] for l in lines[k2:] if l.startswith("franke: step")])
    end = [l for l in lines if l.startswith("franke: END")]
    print(f"== case {case}: {tag}: first solve {ia['first_iters']} (reference) / {len(first)} (device) steps; second: reference {ia['iters']} iterations "
          f"(result {ia['result']}, {len(a)} steps), device {end[-1] if end else '?'} ({len(c)} steps)")
    print("step | hot: ref dev | gap: ref, device | alpha: ref, device | zeta: ref, device")
    for k in range(min(max(len(a), len(c)), 40)):
        g = lambda t, j: f"{t[k][j]:.9e}" if k < len(t) else "-"
        print(k + 1, "|", int(a[k][6]) if k < len(a) else "-", int(c[k][6]) if k < len(c) else "-", "|", g(a, 0), g(c, 0), "|", g(a, 1), g(c, 1), "|", g(a, 3), g(c, 3))
