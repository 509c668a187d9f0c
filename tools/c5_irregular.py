"""BASELINE.json configs[4] ("CUTE-style random sparse NLP n = 1e6, full SQP loop (Hqp_SqpPowell)") at FULL size on the
irregular generator: Prg_GridNLP (oracle/ref_sqpdrive.cc) on g x g cells with `far` couplings between distant cells
through the reference's unmodified Hqp_SqpPowell, with the device-resident MehrotraHip and with the reference's own
Hqp_IpsMehrotra, both driving RedSpBKPHip (dissection of the KKT graph, mat_ordering 2); and factor + solve of the
KKT system of the QP of the same structure.  One JSON line per case.
Usage: python tools/c5_irregular.py [g far] ...   (default: 316 1000, 1000 1000, 1000 10000)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hqp_amd import ipmatrix, problems
from oracle import refapi

args = [int(a) for a in sys.argv[1:]] or [316, 1000, 1000, 1000, 1000, 10000]
for g, far in zip(args[::2], args[1::2]):
    line = {"workload": f"Prg_GridNLP {g} x {g} cells + {far} far couplings; Hqp_SqpPowell, analytic Hessian + Gerschgorin; RedSpBKPHip, mat_ordering 2"}
    for solver in ("MehrotraHip", "Mehrotra"):
        if solver == "Mehrotra" and g * g > 200000:
            continue  # (the reference's own interior-point solver on one host core: minutes at 10^6 variables)
        t0 = time.perf_counter()
        try:
            r = refapi.sqp_grid(g, g, solver, "RedSpBKPHip", host="hip", ordering=2, far=far)
            r = {k: (float(v) if isinstance(v, (float, np.floating)) else v) for k, v in r.items()}
            r["wall_s"] = time.perf_counter() - t0
            line[solver] = r
        except refapi.RefError as e:
            line[solver] = {"error": str(e)}
    prog = problems.grid_sparse_qp(g, g, seed=5, long_range=far)
    st = [torch.as_tensor(a).cuda() for a in problems.ip_state(prog, 1, 1.0)]
    M = ipmatrix.IpRedSpBKP(device_vectors=True, ordering=2)
    t0 = time.perf_counter()
    M.init(prog)
    init_s = time.perf_counter() - t0
    d = [torch.zeros(k, dtype=torch.float64, device="cuda") for k in (prog.n, prog.me, prog.m, prog.m)]
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        M.factor(prog, st[0], st[1])
        res = M.solve(prog, *st, *d)
        ts.append(time.perf_counter() - t0)
    s = M.stats()
    line["kkt"] = {"variables": prog.n, "kkt_dim": s["dim"], "ms_per_factor_solve": 1e3 * float(np.median(ts[1:])), "residual": res, "init_s": init_s,
                   "flops_factor": s["flops_factor"], "max_front": s["max_front"], "tree_levels": s["n_levels"],
                   "hbm_gb": (s["bytes_panels"] + s["bytes_updates"]) / 1e9,
                   "tflops": s["flops_factor"] / float(np.median(ts[1:])) / 1e12}
    del M
    torch.cuda.empty_cache()
    print(json.dumps(line), flush=True)
