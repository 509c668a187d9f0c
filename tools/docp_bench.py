"""C4 structure (multistage LQ DOCP, K stages, nx states, nu controls): one factor + solve of the
KKT system and the whole QP with the device-resident Mehrotra loop.
   python3 tools/docp_bench.py [nx ...]       (K = 200, nu = 10)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hqp_amd import problems, ipmatrix

for nx in [int(a) for a in sys.argv[1:]] or [50, 100, 200]:
    prog = problems.lq_docp(200, nx, 10)
    st = problems.ip_state(prog, seed=1)
    M = ipmatrix.IpSpBKP(device_vectors=True)
    t0 = time.perf_counter()
    M.init(prog)
    t_init = time.perf_counter() - t0
    dev = [torch.as_tensor(a).cuda() for a in st]
    d = [torch.zeros(k, dtype=torch.float64, device="cuda") for k in (prog.n, prog.me, prog.m, prog.m)]
    for _ in range(2):
        M.factor(prog, dev[0], dev[1]); res = M.solve(prog, *dev, *d)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        M.factor(prog, dev[0], dev[1]); res = M.solve(prog, *dev, *d)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
    s = M.stats()
    M2 = ipmatrix.IpSpBKP()
    M2.init(prog)
    M2.mehrotra(prog)
    x, y, z, w, info = M2.mehrotra(prog)
    print(f"nx={nx} dim={s['dim']} levels={s['n_levels']} max_front={s['max_front']} init {t_init:.1f} s  factor+solve {ms:.2f} ms "
          f"res {res:.2e}  |  QP: {info['iters']} iterations, result {info['result']}, {info['ms_total']:.1f} ms", flush=True)
