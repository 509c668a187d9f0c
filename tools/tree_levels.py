"""Level structure of the assembly tree of a workload (needs the GPU box: init() creates the device handle): per level the number of
fronts, pivots and border rows (min / max).  python tools/tree_levels.py [n band max_pivots]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hqp_amd import ipmatrix, problems  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
band = int(sys.argv[2]) if len(sys.argv) > 2 else 80
mp = int(sys.argv[3]) if len(sys.argv) > 3 else 0
prog = problems.banded_qp(n, band, seed=12345)
mat = ipmatrix.IpRedSpBKP(device=0, device_vectors=False, max_pivots=mp)
mat.init(prog)
s = mat.structure()
lev, npiv, nb = np.asarray(s["level"]), np.asarray(s["npiv"]), np.asarray(s["nborder"])
for l in range(lev.max() + 1):
    m = lev == l
    print(f"level {l:3d}: {m.sum():6d} fronts, pivots {npiv[m].min():4d}..{npiv[m].max():4d}, border {nb[m].min():4d}..{nb[m].max():4d}")
