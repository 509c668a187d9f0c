"""A/B of the STAGED engine's fp64 MFMA product: LDS-DMA staging (default) against register staging
(HQPKKT_NO_LDSDMA=1), same process order, shapes of the C4 recursion and the square reference sizes."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys; sys.path.insert(0, %r)
from hqp_amd import ipmatrix
for (M, N, K, lo) in [(1024,1024,1024,0),(2048,2048,2048,0),(4096,4096,4096,0),(5000,5050,5000,0),(5050,5050,5000,1),(8192,8192,8192,0),
                      (5000,640,5000,0),(640,640,5000,1),(4360,640,5000,0),(3000,3050,3000,0),(3050,3050,3000,1)]:
    ms, tf, err = ipmatrix.bench_dgemm(M, N, K, lo, lo, reps=5)
    print(f"dgemm M={M} N={N} K={K} lower={lo}: {ms:.3f} ms  {tf:.2f} TFLOP/s  ({tf/78.6*100:.1f}%% of 78.6)  err {err:.1e}", flush=True)
''' % ROOT
for name, env in (("lds-dma, 2 x 4 waves of 64 x 32 (4 waves per SIMD)", {}), ("lds-dma, 2 x 2 waves of 64 x 64 (2 waves per SIMD)", {"HQPKKT_DGEMM_WAVES": "4"}),
                  ("register-staged, 2 x 2 waves", {"HQPKKT_NO_LDSDMA": "1"})):
    print("==", name, flush=True)
    subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, **env))
