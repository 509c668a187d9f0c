"""BASELINE configs[3] (multistage LQ DOCP, K stages of nx states) through the dense hand-over of
the STAGED engine, everything generated on the device: factor + solve times, rate of the
recursion's matrix products, residual.  python tools/c4_bench.py K nx nu [reps]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hqp_amd import problems, ipmatrix


def make(K, nx, nu, seed=0, x_bounds=0):
    """fx dense random with spectral radius ~0.9 (different for every stage), fu dense random,
    Q = diag (1 on states, 0.1 on controls), x_0 fixed, box bounds on every control (and on the
    first x_bounds states of every stage)."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    nz = nx + nu
    F = []
    for k in range(K):
        blk = torch.empty((nx, nz), dtype=torch.float64, device="cuda")
        blk.uniform_(-1.0, 1.0, generator=g)
        blk[:, :nx] *= 0.9 / np.sqrt(nx / 3.0)
        F.append(blk)
    n = K * nz + nx
    qd = np.ones(n)
    for k in range(K):
        qd[k * nz + nx:(k + 1) * nz] = 0.1
    Q = (np.arange(n + 1, dtype=np.int32), np.arange(n, dtype=np.int32), qd)
    E = (np.arange(nx + 1, dtype=np.int32), np.arange(nx, dtype=np.int32), np.ones(nx))
    ucols = np.concatenate([k * nz + nx + np.arange(nu) for k in range(K)])
    cols = np.concatenate([ucols, ucols])
    vals = np.concatenate([np.ones(ucols.size), -np.ones(ucols.size)])
    if x_bounds:
        xb = np.concatenate([k * nz + np.arange(x_bounds) for k in range(1, K + 1)])
        cols, vals = np.concatenate([cols, xb]), np.concatenate([vals, -np.ones(xb.size)])
    m = cols.size
    C = (np.arange(m + 1, dtype=np.int32), cols.astype(np.int32), vals)
    return problems.DenseDocp([nx] * (K + 1), [nu] * K, Q, E, C, F, nx, m)


def run(K, nx, nu, reps=3, x_bounds=0, profile=False):
    t0 = time.time()
    dq = make(K, nx, nu, x_bounds=x_bounds)
    torch.cuda.synchronize()
    t1 = time.time()
    M = ipmatrix.IpLQDOCP(device_vectors=True)
    M.init_dense(dq)
    dq.F = None  # the engine holds its own copy
    torch.cuda.empty_cache()
    t2 = time.time()
    g = torch.Generator(device="cuda").manual_seed(1)
    rnd = lambda k, lo, hi: torch.empty(k, dtype=torch.float64, device="cuda").uniform_(lo, hi, generator=g)
    n, me, m = dq.dims
    z, w = rnd(m, 0.1, 1.1), rnd(m, 0.1, 1.1)
    r = [rnd(k, -0.5, 0.5) for k in (n, me, m, m)]
    d = [torch.zeros(k, dtype=torch.float64, device="cuda") for k in (n, me, m, m)]
    out = []
    if profile:
        M.set_profile(True)
    for it in range(reps + 1):
        torch.cuda.synchronize()
        ta = time.time()
        M.factor(None, z, w)
        tb = time.time()
        res = M.solve(None, z, w, *r, *d)
        tc = time.time()
        s = M.stats()
        if it:
            out.append((tb - ta, tc - tb, s["ms_factor"], s["ms_solve"], res, s["refine_rounds"]))
    tf = float(np.median([o[0] for o in out])), float(np.median([o[1] for o in out]))
    s = M.stats()
    rec = {"K": K, "nx": nx, "nu": nu, "n": n, "me": me, "m": m, "gen_s": round(t1 - t0, 2), "init_s": round(t2 - t1, 2),
           "factor_s": tf[0], "solve_s": tf[1], "factor_solve_per_s": 1.0 / (tf[0] + tf[1]),
           "ms_factor_dev": float(np.median([o[2] for o in out])), "ms_solve_dev": float(np.median([o[3] for o in out])),
           "res": out[-1][4], "refine_rounds": out[-1][5], "flops_factor": s["flops_factor"],
           "tflops_factor": s["flops_factor"] / tf[0] / 1e12, "frac_fp64_peak": s["flops_factor"] / tf[0] / 78.6e12,
           "hbm_gb": (s["bytes_panels"] + s["bytes_updates"]) / 1e9}
    if profile:
        rec["profile_ms"] = {k: (round(v[0], 3), v[1]) for k, v in M.profile().items() if v[1]}
    return rec


if __name__ == "__main__":
    K, nx, nu = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    print(json.dumps(run(K, nx, nu, reps, profile="--profile" in sys.argv)), flush=True)
