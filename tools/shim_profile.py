"""The reference's Hqp_IpsMehrotra on our plugin through the shim (host vectors, as in a real HQP run) on the
double-integrator QP: IP iterations per second, a few runs.   python3 tools/shim_profile.py [K] [runs]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import refapi  # noqa: E402
from hqp_amd import problems  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
prog = problems.did_like_qp(K)
for i in range(runs):
    r = refapi.ip_solve(prog, "Mehrotra", "RedSpBKPHip", host="hip")
    print(i, r["iters"], r["result"], f"{r['seconds'] * 1e3:.3f} ms", f"{r['iters'] / r['seconds']:.1f} it/s", flush=True)
r = refapi.ip_solve(prog, "Mehrotra", "RedSpBKP", host="hip")
print("reference plugin:", r["iters"], f"{r['seconds'] * 1e3:.3f} ms", f"{r['iters'] / r['seconds']:.1f} it/s")
