"""Device time per kernel class of one factor + solve of a sparse KKT system (events around every launch):
python tools/kkt_classes.py far|mesh|band|c2 [ordering [max_pivots]]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hqp_amd import ipmatrix, problems
what = sys.argv[1] if len(sys.argv) > 1 else "far"
prog = {"far": lambda: problems.grid_sparse_qp(1000, 1000, seed=5, long_range=10000), "mesh": lambda: problems.grid_sparse_qp(1000, 1000),
        "band": lambda: problems.banded_long_range_qp(100000, 10, 1000), "c2": lambda: problems.banded_qp(40000, 80)}[what]()
st = [torch.as_tensor(a).cuda() for a in problems.ip_state(prog, 1, 1.0)]
M = (ipmatrix.IpSpBKP if what == "c2" else ipmatrix.IpRedSpBKP)(device_vectors=True, ordering=int(sys.argv[2]) if len(sys.argv) > 2 else (0 if what == "c2" else 2),
                                                                  max_pivots=int(sys.argv[3]) if len(sys.argv) > 3 else 0)
M.init(prog)
d = [torch.zeros(k, dtype=torch.float64, device="cuda") for k in (prog.n, prog.me, prog.m, prog.m)]
try:
    M.factor(prog, st[0], st[1]); M.solve(prog, *st, *d)
except Exception:
    pass
M.set_profile(True)
res = None
try:
    M.factor(prog, st[0], st[1]); res = M.solve(prog, *st, *d)
except Exception as e:  # (timing experiments with wrong arithmetic)
    print("failed:", repr(e)[:100])
s = M.stats()
print(json.dumps({"what": what, "res": res, "flops_factor": s["flops_factor"], "max_front": s["max_front"], "levels": s["n_levels"], "supernodes": s["n_supernodes"],
                  "profile_ms": {k: (round(v[0], 3), v[1]) for k, v in M.profile().items() if v[1]}}))
