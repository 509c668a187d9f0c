"""Diagnosis: Hqp_IpsFranke on a DID QP step by step - the reference's loop with its own plugin, the reference's loop with
OUR plugin (through the shim) and the device-resident loop (HQPKKT_TRACE_IP) - gap / alpha / zeta / rhomin side by side.
python tools/franke_trace.py K qx [SpBKP|RedSpBKP]"""
import os, sys, re, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
K, qx = int(sys.argv[1]), float(sys.argv[2])
kind = sys.argv[3] if len(sys.argv) > 3 else "SpBKP"
if len(sys.argv) > 4:  # child: the device loop with its trace on stderr
    from hqp_amd import problems, ipmatrix
    prog = problems.did_like_qp(K, qx)
    M = (ipmatrix.IpRedSpBKP if kind == "RedSpBKP" else ipmatrix.IpSpBKP)()
    M.init(prog)
    M.franke(prog, max_iters=250)
    sys.exit(0)
from hqp_amd import problems
from oracle import refapi
prog = problems.did_like_qp(K, qx)
a = refapi.trace_franke(prog, kind, host="hip")
b = refapi.trace_franke(prog, kind + "Hip", host="hip")
env = dict(os.environ, HQPKKT_TRACE_IP="1", HQPKKT_TINY_IN_LOOP="0")
out = subprocess.run([sys.executable, __file__, str(K), str(qx), kind, "child"], env=env, capture_output=True, text=True).stderr
c = np.array([[float(v) for v in re.findall(r"(?:gap|alpha|alphabar|zeta|rhomin|resid) (\S+)", l)] for l in out.splitlines() if l.startswith("franke:")])
print(f"steps: reference {len(a)}, reference loop + our plugin {len(b)}, device loop {len(c)}")
print("step | gap: ref, ref+ours, device | alpha: ref, ref+ours, device | rhomin ref+ours, device | device resid")
for k in range(max(len(a), len(b), len(c))):
    g = lambda t, j: f"{t[k][j]:.6e}" if k < len(t) else "-"
    print(k + 1, "|", g(a, 0), g(b, 0), g(c, 0), "|", g(a, 1), g(b, 1), g(c, 1), "|", g(b, 4), g(c, 4), "|", g(c, 5))
