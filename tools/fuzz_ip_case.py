"""One Franke / Mehrotra run on a DID QP under option variants (diagnostics for tools/fuzz_ip.py).
Usage: python tools/fuzz_ip_case.py K qx [Franke|Mehrotra] [SpBKP|RedSpBKP]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hqp_amd import problems, ipmatrix
from oracle import refapi
K, qx = int(sys.argv[1]), float(sys.argv[2])
solver = sys.argv[3] if len(sys.argv) > 3 else "Franke"
kind = sys.argv[4] if len(sys.argv) > 4 else "SpBKP"
prog = problems.did_like_qp(K, qx)
ref = refapi.ip_solve(prog, solver, kind)
print("reference:", ref["result"], ref["iters"], flush=True)
for v in (dict(), dict(zd_policy=0), dict(slack_policy=0), dict(slack_policy=1), dict(small_fronts=False), dict(pivot_eps=0.0),
          dict(mat_tol=0.5), dict(slack_policy=1, zd_policy=0), dict(leaf_size=100000), dict(leaf_size=100000, slack_policy=1),
          dict(leaf_size=100000, slack_policy=0, zd_policy=0), dict(pivot_eps=1e-14), dict(mat_eps=1e-13, slack_policy=1)):
    M = (ipmatrix.IpRedSpBKP if kind == "RedSpBKP" else ipmatrix.IpSpBKP)(**v)
    M.init(prog)
    x, y, z, w, info = (M.franke if solver == "Franke" else M.mehrotra)(prog, max_iters=250)
    s = M.stats()
    print(v, info["result"], info["iters"], {k: s[k] for k in ("n_2x2", "n_perturbed", "refine_rounds", "n_slow_pivots")}, flush=True)
