"""First look at the STAGED engine on the GPU: parity numbers and the dgemm rate."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from hqp_amd import problems, ipmatrix
from model_staged import StagedModel, kkt_residual

def relerr(a, b):
    return max(np.abs(x - y).max() / max(np.abs(y).max(), 1e-300) for x, y in zip(a, b) if len(y))

if "gemm" in sys.argv:
    for (M, N, K, lo) in [(512, 512, 512, 0), (1024, 1024, 1024, 0), (2048, 2048, 2048, 0), (4096, 4096, 4096, 0), (5000, 5050, 5000, 0),
                          (5050, 5050, 5000, 1), (8192, 8192, 8192, 0), (1000, 1050, 1000, 0), (5000, 640, 5000, 0), (640, 640, 5000, 1), (4360, 640, 5000, 0), (400, 410, 400, 0), (200, 210, 200, 0), (60, 5000, 60, 0)]:
        ms, tf, err = ipmatrix.bench_dgemm(M, N, K, lo, lo, reps=3)
        print(f"dgemm M={M} N={N} K={K} lower={lo}: {ms:.3f} ms  {tf:.2f} TFLOP/s  ({tf/78.6*100:.1f}% of 78.6)  err {err:.1e}", flush=True)
if "parity" in sys.argv:
    cases = {"plain": problems.lq_docp(10, 6, 2), "final5": problems.lq_docp(12, 5, 3, final_eq=5),
             "mix": problems.lq_docp(12, 5, 3, path_eq=2, final_eq=3, x_bounds=2),
             "free": problems.lq_docp(8, 4, 2, x0_fixed=False, final_eq=2), "did50": problems.did_like_qp(50),
             "tiles": problems.lq_docp(3, 150, 20, seed=5)}
    for name, prog in cases.items():
        st = problems.ip_state(prog, 3, 1.0)
        M = ipmatrix.IpLQDOCP()
        try:
            M.init(prog)
            M.factor(prog, st[0], st[1])
            d = [np.zeros(k) for k in (prog.n, prog.me, prog.m, prog.m)]
            M.step(prog, *st, *d)
            R = StagedModel(prog); R.factor(st[0], st[1]); md = R.step(*st[2:])
            print(name, "step res", kkt_residual(prog, st[0], st[1], st[2:], d), "vs model", relerr(d, md), "ranks", M.stage_ranks()[:4].tolist(), flush=True)
            res = M.solve(prog, *st, *d)
            print(name, "solve res", res, M.stats()["refine_rounds"], flush=True)
        except Exception as e:
            print(name, "FAILED", repr(e), flush=True)
if "time" in sys.argv:
    for nx in (50, 100, 200, 400, 800):
        prog = problems.lq_docp(200, nx, 10, seed=11)
        st = problems.ip_state(prog, 5, 1.0)
        M = ipmatrix.IpLQDOCP()
        t0 = time.time(); M.init(prog); t1 = time.time()
        d = [np.zeros(k) for k in (prog.n, prog.me, prog.m, prog.m)]
        for _ in range(3):
            M.factor(prog, st[0], st[1]); res = M.solve(prog, *st, *d)
        s = M.stats()
        print(f"K=200 nx={nx}: init {t1-t0:.2f} s factor {s['ms_factor']:.2f} ms solve {s['ms_solve']:.2f} ms res {res:.2e} rounds {s['refine_rounds']} "
              f"flops {s['flops_factor']:.3e} -> {s['flops_factor']/s['ms_factor']/1e9:.2f} TFLOP/s", flush=True)
