import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from hqp_amd import ipmatrix, problems
K = 2000
prog = problems.did_like_qp(K)
for ls in (0, 8, 12, 16, 20, 24, 48):
    M = ipmatrix.IpRedSpBKP(leaf_size=ls)
    M.init(prog)
    M.mehrotra(prog)
    best = 0
    for _ in range(4):
        _x, _y, _z, _w, info = M.mehrotra(prog)
        best = max(best, info["iters"] / (info["ms_total"] * 1e-3))
    st = M.stats()
    print("leaf_size", ls, "iters", info["iters"], "it/s %.0f" % best, "levels", st["n_levels"], "fronts", st["n_supernodes"], "max_front", st["max_front"], M.debug(31)[4:])
