"""Device-resident Mehrotra loop on the C3-like DID structure, for rocprofv3:
   rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ip -o ip -- python3 tools/ip_profile.py [K] [mode]
prints afterwards (from its own run) the factorisation statistics."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hqp_amd import problems, ipmatrix

K = int(sys.argv[1]) if len(sys.argv) > 1 else 33333
mode = sys.argv[2] if len(sys.argv) > 2 else "RedSpBKP"
amalg = len(sys.argv) > 3 and sys.argv[3] == "amalg"
prog = problems.did_like_qp(K)
M = (ipmatrix.IpRedSpBKP if mode == "RedSpBKP" else ipmatrix.IpSpBKP)(amalgamation=amalg)
M.init(prog)
for rep in range(3):
    x, y, z, w, info = M.mehrotra(prog)
    print(rep, {k: info[k] for k in ("result", "iters", "n_factor", "n_solve", "ms_total")},
          "it/s", round(info["iters"] / (info["ms_total"] * 1e-3), 1), flush=True)
st = M.stats()
print({k: st[k] for k in ("dim", "sbw", "n_supernodes", "n_levels", "nnz_kkt", "nnz_factor", "n_2x2", "n_perturbed",
                          "n_slow_pivots", "max_front") if k in st})
