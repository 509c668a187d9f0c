"""The symbolic phase alone (host code: runs without a GPU; the upload that follows it fails there and is ignored):
fronts, levels and flop count of the tree for the irregular test structures.   python3 tools/symbolic_stats.py [small]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hqp_amd import ipmatrix, problems  # noqa: E402

small = len(sys.argv) > 1
cases = [("mesh 200 x 200 + 400 far", lambda: problems.grid_sparse_qp(200, 200, seed=5, long_range=400)),
         ("band of 21, 1000 far, n = 1e5", lambda: problems.banded_long_range_qp(100000, 10, 1000)),
         ("cute-like rows, n = 2e4", lambda: problems.cute_like_qp(20000))]
if not small:
    cases += [("mesh 1000 x 1000 + 10000 far", lambda: problems.grid_sparse_qp(1000, 1000, seed=5, long_range=10000)),
              ("cute-like rows, n = 1e5", lambda: problems.cute_like_qp(100000))]
for name, make in cases:
    prog = make()
    M = ipmatrix.IpRedSpBKP(ordering=2)
    t0 = time.perf_counter()
    try:
        M.init(prog)
    except Exception:  # noqa: BLE001  (no device here: the symbolic phase has run)
        pass
    s = M.stats()
    print(f"{name}: {time.perf_counter() - t0:.1f} s  dim {s['dim']} fronts {s['n_supernodes']} levels {s['n_levels']} max_front {s['max_front']} "
          f"Gflop {s['flops_factor'] / 1e9:.1f} nnzL {s['nnz_factor'] / 1e6:.1f} M  updates {s['bytes_updates'] / 1e9:.2f} GB", flush=True)
