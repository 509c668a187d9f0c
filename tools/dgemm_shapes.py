"""hqpkkt_debug_dgemm on the shapes given as MxNxKxlower[xmirror] (default: the two products of a C4 stage)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hqp_amd import ipmatrix
shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(5000, 5050, 5000, 0, 0), (5050, 5050, 5000, 1, 0)]
for sh in shapes:
    M, N, K, lo = sh[:4]
    mir = sh[4] if len(sh) > 4 else 0
    ms, tf, err = ipmatrix.bench_dgemm(M, N, K, lo, mir, reps=10)
    print(f"dgemm M={M} N={N} K={K} lower={lo} mirror={mir}: {ms:.3f} ms  {tf:.2f} TFLOP/s  ({tf/78.6*100:.1f}% of 78.6)  err {err:.1e}", flush=True)
