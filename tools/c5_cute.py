"""SURVEY.md 8(d) C5's row density (problems.cute_like_qp: 10 ... 100 entries per row) at n variables through the tree of the
graph's own dissection: init time, factor + solve time, structure.   python tools/c5_cute.py [n]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hqp_amd import ipmatrix, problems
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
prog = problems.cute_like_qp(n)
st = [torch.as_tensor(a).cuda() for a in problems.ip_state(prog, 1, 1.0)]
M = ipmatrix.IpRedSpBKP(device_vectors=True, ordering=2)
t0 = time.perf_counter()
M.init(prog)
t_init = time.perf_counter() - t0
d = [torch.zeros(k, dtype=torch.float64, device="cuda") for k in (prog.n, prog.me, prog.m, prog.m)]
for _ in range(2):
    M.factor(prog, st[0], st[1]); res = M.solve(prog, *st, *d)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    M.factor(prog, st[0], st[1]); res = M.solve(prog, *st, *d)
torch.cuda.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / reps
s = M.stats()
print(json.dumps({"workload": f"cute_like_qp({n}): n {prog.n} me {prog.me} m {prog.m}, nnz Q {len(prog.Q[2])} A {len(prog.A[2])}; RedSpBKP, ordering 2",
                  "ms_per_factor_solve": ms, "residual": res, "init_s": t_init, "tflops": s["flops_factor"] / ms / 1e9,
                  **{k: s[k] for k in ("dim", "max_front", "n_levels", "n_supernodes", "flops_factor", "bytes_panels", "bytes_updates")}}))
