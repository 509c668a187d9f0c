import os
"""Time of the control-sized elimination (k_st_small / k_st_init_factor) of stages whose matrices live in global
memory (StagedPlan::big): per launch, from the per-class event profile.  Usage: python tools/bigstage_time.py"""
import sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from hqp_amd import problems, ipmatrix
from common import new_d, rel_err
CASES = {"nu100": lambda: problems.lq_docp(3, 200, 100, seed=2),
         "nu200": lambda: problems.lq_docp(3, 260, 200, seed=2),
         "nu300_path40": lambda: problems.lq_docp(3, 200, 300, path_eq=40, seed=5),
         "nu512": lambda: problems.lq_docp(2, 520, 512, seed=6),
         "final140": lambda: problems.lq_docp(10, 160, 20, final_eq=140, seed=3),
         "free_x0_250": lambda: problems.lq_docp(3, 250, 6, x0_fixed=False, final_eq=3, seed=4),
         "free_x0_1000": lambda: problems.lq_docp(2, 1000, 4, x0_fixed=False, seed=4)}
for name, mk in CASES.items():
    prog = mk(); st = problems.ip_state(prog, 6, 1.0)
    S = ipmatrix.IpLQDOCP()
    S.init(prog); S.factor(prog, st[0], st[1]); d = new_d(prog); res = S.solve(prog, *st, *d)
    S.set_profile(True)
    S.factor(prog, st[0], st[1]); res = S.solve(prog, *st, *d)
    pr = {k: (round(v[0], 3), v[1]) for k, v in S.profile().items() if v[1]}
    print(name, "res %.1e" % res, "staged_small", pr.get("staged_small"), "all", pr, flush=True)
