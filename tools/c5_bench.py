"""BASELINE.json configs[4] stand-in ("CUTE-style sparse NLP, full SQP loop"): Prg_GridNLP (oracle/ref_sqpdrive.cc)
through the reference's own Hqp_SqpPowell + Hqp_IpsMehrotra, once with the reference's RedSpBKP (CPU, one core)
and once with RedSpBKPHip (this repo) with the tree of the RCM band and of the graph itself.  The reference runs
up to the size given (its time grows ~ n^2).  One JSON line per size.
Usage: python tools/c5_bench.py [max grid edge for the reference] [grid edges ...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from hqp_amd import ipmatrix, problems
from oracle import refapi


def kkt_rate(g, ordering, reps=5):
    """factor + solve per second of the KKT system of the mesh QP of the same structure (resident vectors)"""
    import torch
    prog = problems.grid_sparse_qp(g, g)
    st = problems.ip_state(prog, 1, 1.0)
    M = ipmatrix.IpRedSpBKP(ordering=ordering)
    t0 = time.perf_counter()
    M.init(prog)
    init_s = time.perf_counter() - t0
    d = [np.zeros(k) for k in (prog.n, prog.me, prog.m, prog.m)]
    M.factor(prog, st[0], st[1])
    res = M.solve(prog, *st, *d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        M.factor(prog, st[0], st[1])
        M.solve(prog, *st, *d)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    s = M.stats()
    return dict(ms_per_factor_solve=1e3 * dt, residual=res, init_s=init_s, nnz_factor=s["nnz_factor"],
                flops_factor=s["flops_factor"], max_front=s["max_front"], levels=s["n_levels"], dim=s["dim"])


def main():
    ref_max = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    sizes = [int(a) for a in sys.argv[2:]] or [60, 100, 150, 200, 300]
    for g in sizes:
        line = {"workload": f"Prg_GridNLP {g} x {g} cells, Hqp_SqpPowell + Hqp_IpsMehrotra, analytic Hessian + Gerschgorin",
                "cores": os.cpu_count()}
        if g <= ref_max:
            r = refapi.sqp_grid(g, g, "Mehrotra", "RedSpBKP", host="hip")
            line["reference_RedSpBKP_cpu"] = {k: (float(v) if isinstance(v, (float, np.floating)) else v) for k, v in r.items()}
        for o in ((0, 1, 2) if g <= 300 else (2,)):  # the band's tree beyond 300 x 300: tens of GB of factor
            try:
                r = refapi.sqp_grid(g, g, "Mehrotra", "RedSpBKPHip", host="hip", ordering=o)
                line[f"RedSpBKPHip_ordering{o}"] = {k: (float(v) if isinstance(v, (float, np.floating)) else v) for k, v in r.items()}
            except refapi.RefError as e:
                line[f"RedSpBKPHip_ordering{o}"] = {"error": str(e)}
            try:
                line[f"kkt_ordering{o}"] = kkt_rate(g, o)
            except ipmatrix.KktError as e:
                line[f"kkt_ordering{o}"] = {"error": str(e)}
        try:  # our device-resident Mehrotra loop (shim/Hqp_IpsMehrotraHip.C) under the same SQP host
            r = refapi.sqp_grid(g, g, "MehrotraHip", "RedSpBKPHip", host="hip", ordering=2)
            line["MehrotraHip_RedSpBKPHip_ordering2"] = {k: (float(v) if isinstance(v, (float, np.floating)) else v) for k, v in r.items()}
        except refapi.RefError as e:
            line["MehrotraHip_RedSpBKPHip_ordering2"] = {"error": str(e)}
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
