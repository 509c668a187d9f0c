"""C2 system: tree levels (fronts, pivots, border rows), what the solve fuses (introspection 31) and the device time per
kernel class of a factor + solve (events around every launch), for the default and for 160 pivots per supernode."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hqp_amd import ipmatrix, problems
prog = problems.banded_qp(40000, 80, seed=12345)
st = problems.ip_state(prog, seed=1)
for mp in (0, 160):
    M = ipmatrix.IpSpBKP(device=0, device_vectors=True, max_pivots=mp)
    M.init(prog)
    s = M.structure()
    lev, npiv, nb = np.asarray(s["level"]), np.asarray(s["npiv"]), np.asarray(s["nborder"])
    print("max_pivots", mp, "top:", M.debug(31))
    for l in range(lev.max() + 1):
        m = lev == l
        print(f"  level {l:3d}: {m.sum():6d} fronts, pivots {npiv[m].min():4d}..{npiv[m].max():4d}, border {nb[m].min():4d}..{nb[m].max():4d}")
    dev = [torch.as_tensor(a).cuda() for a in st]
    d = [torch.zeros(k, dtype=torch.float64, device="cuda") for k in (prog.n, prog.me, prog.m, prog.m)]
    for _ in range(3):
        M.factor(prog, dev[0], dev[1]); M.solve(prog, *dev, *d)
    M.set_profile(True)
    for _ in range(5):
        M.factor(prog, dev[0], dev[1]); M.solve(prog, *dev, *d)
    pr = M.profile()
    print("  ", {k: (round(v[0] / 5, 4), v[1] / 5) for k, v in pr.items() if v[1]})
