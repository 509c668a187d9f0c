"""factor + solve of the same KKT system through the tree of the RCM band (ordering 0) and of the graph's own
dissection (1, 2): python tools/ordering_compare.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from hqp_amd import ipmatrix, problems

CASES = [("C2 banded n=40000 b=80", lambda: problems.banded_qp(40000, 80), ipmatrix.IpSpBKP),
         ("banded n=400000 b=80", lambda: problems.banded_qp(400000, 80), ipmatrix.IpSpBKP),
         ("DID K=33333", lambda: problems.did_like_qp(33333), ipmatrix.IpRedSpBKP),
         ("DOCP K=200 nx=100 nu=10", lambda: problems.lq_docp(200, 100, 10), ipmatrix.IpRedSpBKP),
         ("mesh 300x300", lambda: problems.grid_sparse_qp(300, 300), ipmatrix.IpRedSpBKP)]
for name, make, cls in CASES:
    prog = make()
    st = [torch.as_tensor(a).cuda() for a in problems.ip_state(prog, 1, 1.0)]
    for o in (0, 1, 2):
        try:
            M = cls(device_vectors=True, ordering=o)
            t0 = time.perf_counter()
            M.init(prog)
            init = time.perf_counter() - t0
            d = [torch.zeros(k, dtype=torch.float64, device="cuda") for k in (prog.n, prog.me, prog.m, prog.m)]
            ts = []
            for _ in range(6):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                M.factor(prog, st[0], st[1])
                res = M.solve(prog, *st, *d)
                ts.append(time.perf_counter() - t0)
            s = M.stats()
            print(f"{name}: ordering {o}: {1e3 * np.median(ts[1:]):8.2f} ms  res {res:.1e}  init {init:.2f} s  levels {s['n_levels']} "
                  f"flops {s['flops_factor']:.3g} nnzL {s['nnz_factor']:.3g} rounds {s['refine_rounds']} perturbed {s['n_perturbed']}", flush=True)
            del M
        except Exception as e:
            print(f"{name}: ordering {o}: {e}", flush=True)
