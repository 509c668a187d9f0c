import os
"""One big-stage case under rocprofv3 (per-kernel durations).  Usage: python tools/bigstage_one.py nu controls [path_eq]"""
import sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from hqp_amd import problems, ipmatrix
from common import new_d
nx, nu = int(sys.argv[1]), int(sys.argv[2])
pe = int(sys.argv[3]) if len(sys.argv) > 3 else 0
prog = problems.lq_docp(3, nx, nu, path_eq=pe, seed=2); st = problems.ip_state(prog, 6, 1.0)
S = ipmatrix.IpLQDOCP()
S.init(prog)
d = new_d(prog)
for _ in range(3):
    S.factor(prog, st[0], st[1]); res = S.solve(prog, *st, *d)
print(nx, nu, pe, res)
