"""Time stamps per workgroup of the STAGED engine's fp64 product (HQPKKT_DGEMM_STAMPS=1): where a launch spends its time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hqp_amd import ipmatrix
shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(5000, 5050, 5000, 0), (5050, 5050, 5000, 1)]
for (M, N, K, lo) in shapes:
    print(M, N, K, lo, ipmatrix.bench_dgemm(M, N, K, lo, lo, reps=3), flush=True)
