"""Device-resident Mehrotra loop on the double-integrator QP (C3 structure): IP iterations per second, with and without
amalgamation; the tree options in use (introspection 31).  python tools/did_rate.py [K]"""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from hqp_amd import ipmatrix, problems
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
prog = problems.did_like_qp(K)
for kw in ({}, {"amalgamation": True}):
    M = ipmatrix.IpRedSpBKP(**kw)
    M.init(prog)
    M.mehrotra(prog)
    best = None
    for _ in range(3):
        _x, _y, _z, _w, info = M.mehrotra(prog)
        r = info["iters"] / (info["ms_total"] * 1e-3)
        best = max(best or 0, r)
    s = M.structure()
    lev = np.asarray(s["level"])
    print(kw, "NO_TOP" if os.environ.get("HQPKKT_NO_SOLVE_TOP") else "top", "iters", info["iters"], "it/s %.0f" % best, "levels", lev.max() + 1, "top:", M.debug(31))
    M.set_profile(True)
    M.mehrotra(prog)
    pr = M.profile()
    print("   ", {k: (round(v[0] / info["iters"], 4), round(v[1] / info["iters"], 1)) for k, v in pr.items() if v[1]})
    M.set_profile(False)
