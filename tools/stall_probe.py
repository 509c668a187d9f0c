"""Where does a stalled Franke run block?  Re-solves the step at the final iterate with the plugin
and with the CPU oracle and prints the components that limit the step length (diagnostics)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hqp_amd import problems, ipmatrix
from oracle import oracleapi
K, qx, iters = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
kw = eval(sys.argv[4]) if len(sys.argv) > 4 else {}
prog = problems.did_like_qp(K, qx)
M = ipmatrix.IpSpBKP(**kw); M.init(prog)
x, y, z, w, info = M.franke(prog, max_iters=iters)
print({k: info[k] for k in ("result", "iters", "gap", "mu", "alpha")})
m = prog.m
mu = info["gap"] / m
r = [np.zeros(prog.n), np.zeros(prog.me), np.zeros(m), z * w - mu]
P = ipmatrix.IpSpBKP(**kw); P.init(prog); P.factor(prog, z, w)
d = [np.zeros(k) for k in (prog.n, prog.me, m, m)]
res = P.solve(prog, z, w, *r, *d)
O = oracleapi.OracleIpMatrix("SpBKP"); O.init(prog); O.factor(z, w)
od, ores = O.solve(z, w, *r)
print("residual: plugin %.3e oracle %.3e; stats" % (res, ores), {k: P.stats()[k] for k in ("n_2x2", "n_perturbed", "refine_rounds")})
def ratio(v, dv):
    with np.errstate(divide="ignore", invalid="ignore"):
        q = np.where(dv > 0, v / dv, np.inf)
    i = int(np.argmin(q)); return i, q[i]
for name, (dz, dw) in (("plugin", (d[2], d[3])), ("oracle", (od[2], od[3]))):
    iz, qz = ratio(z, dz); iw, qw = ratio(w, dw)
    print(name, "z blocks at %d: ratio %.3e (z %.3e dz %.3e w %.3e)" % (iz, qz, z[iz], dz[iz], w[iz]),
          "| w blocks at %d: ratio %.3e (w %.3e dw %.3e z %.3e)" % (iw, qw, w[iw], dw[iw], z[iw]))
i = ratio(w, d[3])[0]
print("component", i, "dw plugin %.6e oracle %.6e; dz plugin %.6e oracle %.6e" % (d[3][i], od[3][i], d[2][i], od[2][i]))
i = ratio(z, d[2])[0]
print("component", i, "dz plugin %.6e oracle %.6e; dw plugin %.6e oracle %.6e" % (d[2][i], od[2][i], d[3][i], od[3][i]))
print("max |dx - dx_oracle| %.3e of %.3e" % (np.abs(d[0] - od[0]).max(), np.abs(od[0]).max()))
