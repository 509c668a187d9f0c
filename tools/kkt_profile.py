"""Per-class device times of one factor + solve of a mesh-structured KKT system (hqpkkt_set_profile: HIP events around
every launch).  Usage: python tools/kkt_profile.py <grid edge> [ordering] [max_pivots] [leaf_size]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from hqp_amd import ipmatrix, problems

g = int(sys.argv[1])
o = int(sys.argv[2]) if len(sys.argv) > 2 else 2
mp = int(sys.argv[3]) if len(sys.argv) > 3 else 0
leaf = int(sys.argv[4]) if len(sys.argv) > 4 else 0
prog = problems.grid_sparse_qp(g, g)
st = problems.ip_state(prog, 1, 1.0)
M = ipmatrix.IpRedSpBKP(ordering=o, max_pivots=mp, leaf_size=leaf)
M.init(prog)
d = [np.zeros(k) for k in (prog.n, prog.me, prog.m, prog.m)]
M.factor(prog, st[0], st[1])
M.solve(prog, *st, *d)
M.set_profile(True)
M.factor(prog, st[0], st[1])
res = M.solve(prog, *st, *d)
pr = M.profile()
s = M.stats()
print(json.dumps({"grid": g, "ordering": o, "leaf": leaf, "max_pivots": mp, "nodes": s["n_supernodes"], "dim": s["dim"], "levels": s["n_levels"], "max_front": s["max_front"],
                  "flops_factor": s["flops_factor"], "residual": res, "ms_factor": s["ms_factor"], "ms_solve": s["ms_solve"],
                  "classes": {k: v for k, v in pr.items() if v[0] > 0}}))
