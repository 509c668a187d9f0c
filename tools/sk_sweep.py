"""The cut form of the STAGED engine's fp64 product under different plans (same box, one process):
   python3 tools/sk_sweep.py [reps]
every line: the environment of the plan, then ms / TFLOP/s of W (5000 x 5050 x 5000) and G (5050 x 5050 x 5000 lower)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hqp_amd import ipmatrix
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
shapes = [(5000, 5050, 5000, 0), (5050, 5050, 5000, 1)]
plans = [{"HQPKKT_SK_RATIO": "0"}]
for ratio in ("1.3", "1.4", "1.5", "1.6", "1.7", "1.8", "2.0"):
    for E in ("2", "4", "8"):
        for tol in ("0", "0.08", "0.2"):
            plans.append({"HQPKKT_SK_RATIO": ratio, "HQPKKT_SK_E": E, "HQPKKT_SK_TOL": tol})
plans.append({"HQPKKT_SK_RATIO": "0"})
keys = sorted({k for p in plans for k in p})
os.environ["HQPKKT_SK_VERBOSE"] = "1"
for p in plans:
    for k in keys:
        os.environ.pop(k, None)
    os.environ.update(p)
    for (M, N, K, lo) in shapes:
        sys.stderr.flush()
        ms, tf, err = ipmatrix.bench_dgemm(M, N, K, lo, lo, reps=reps)
        print(" ".join(f"{k[10:]}={v}" for k, v in sorted(p.items())).ljust(28), "WG"[lo], f"{ms:7.3f} ms {tf:5.1f} TF err {err:.1e}", flush=True)
