"""Franke + SpBKP on DID QPs under the slack-row orders and a tighter refinement target; and what
each costs on C2 (diagnostics).  Usage: python tools/slack_study.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hqp_amd import problems, ipmatrix
from oracle import refapi
V = [dict(), dict(slack_policy=0), dict(slack_policy=1), dict(mat_eps=1e-13), dict(mat_eps=1e-12)]
for K, qx in ((207, 1.0), (252, 1.0), (105, 1.0), (596, 1.0), (356, 0.01), (772, 1.0), (745, 1.0), (400, 1e-4), (2000, 1e-4)):
    prog = problems.did_like_qp(K, qx)
    ref = refapi.ip_solve(prog, "Franke", "SpBKP")
    out = []
    for v in V:
        M = ipmatrix.IpSpBKP(**v)
        M.init(prog)
        M.franke(prog, max_iters=250)
        t0 = time.perf_counter()
        info = M.franke(prog, max_iters=250)[4]
        out.append((info["result"], info["iters"], round(info["iters"] / (time.perf_counter() - t0))))
    print(K, qx, "reference", (ref["result"], ref["iters"]), out, flush=True)
prog = problems.banded_qp(40000, 80)
st = [torch.as_tensor(a).cuda() for a in problems.ip_state(prog)]
d = [torch.zeros(k, dtype=torch.float64, device="cuda") for k in (prog.n, prog.me, prog.m, prog.m)]
for v in V:
    M = ipmatrix.IpSpBKP(device_vectors=True, **v)
    M.init(prog)
    for _ in range(5):
        M.factor(prog, st[0], st[1]); res = M.solve(prog, *st, *d)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        M.factor(prog, st[0], st[1]); res = M.solve(prog, *st, *d)
    torch.cuda.synchronize()
    s = M.stats()
    print("C2", v, "%.3f ms" % ((time.perf_counter() - t0) / 50 * 1e3), "res %.2e rounds %d slow %d" % (res, s["refine_rounds"], s["n_slow_pivots"]), flush=True)
