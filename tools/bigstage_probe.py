import os
import sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from hqp_amd import problems, ipmatrix
from common import new_d, rel_err
def solve(M, prog, st):
    M.init(prog); M.factor(prog, st[0], st[1]); d = new_d(prog); res = M.solve(prog, *st, *d); return d, res
for name, mk in {"final140": lambda: problems.lq_docp(10, 160, 20, final_eq=140, seed=3),
                 "final60": lambda: problems.lq_docp(6, 80, 20, final_eq=60, seed=3),
                 "final100_nu50": lambda: problems.lq_docp(5, 120, 50, final_eq=100, seed=3),
                 "free_x0_250": lambda: problems.lq_docp(3, 250, 6, x0_fixed=False, final_eq=3, seed=4),
                 "nu300_path40": lambda: problems.lq_docp(3, 200, 300, path_eq=40, seed=5),
                 "nu512": lambda: problems.lq_docp(2, 520, 512, seed=6)}.items():
    prog = mk(); st = problems.ip_state(prog, 6, 1.0)
    try:
        S, F = ipmatrix.IpLQDOCP(), ipmatrix.IpLQDOCPFull()
        ds, rs = solve(S, prog, st); df, rf = solve(F, prog, st)
        print(name, "staged res", rs, "full res", rf, "relerr", rel_err(ds, df), "ranks", S.stage_ranks()[:, :].T.tolist()[0][:12], S.stats()["ms_factor"], flush=True)
    except Exception as e:
        print(name, "FAILED", repr(e), flush=True)
