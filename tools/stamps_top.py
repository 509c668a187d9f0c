"""Time stamps inside k_solve_top (hqpkkt_debug_solve_top_stamps) on the C2 system: per tree level the mean times of
its fronts (us after the launch's first stamp): start, static data in, children arrived, forward done, border
solution arrived, backward done.  python tools/stamps_top.py [n band]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402,F401
from hqp_amd import _lib, ipmatrix, problems  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
band = int(sys.argv[2]) if len(sys.argv) > 2 else 80
prog = problems.banded_qp(n, band, seed=12345)
st = problems.ip_state(prog, seed=1)
M = ipmatrix.IpSpBKP(device=0)
M.init(prog)
d = [np.zeros(k) for k in (prog.n, prog.me, prog.m, prog.m)]
M.factor(prog, st[0], st[1])
for _ in range(3):
    M.solve(prog, *st, *d)
top = M.debug(31)
print("fused fronts", top[0], "from level", top[1], "LDS", top[2], "instance NS", top[3])
out = np.zeros(8 * top[0])
for rep in range(3):
    rc = _lib.lib().hqpkkt_debug_solve_top_stamps(M._h, out.ctypes.data, len(out))
    assert rc == 0, rc
o = out.reshape(-1, 8)
print("level fronts |   start  static-in  children   fwd-done | bwd-start border-in  bwd-done   (mean us after the first start; max in brackets for fwd / bwd done;")
print("             |  split form: the backward sweep is a launch of its own and loads its static data again)")
for lv in sorted(set(o[:, 0].astype(int))):
    m = o[:, 0].astype(int) == lv
    mean = o[m, 1:8].mean(0)
    print(f"{lv:5d} {m.sum():6d} | " + " ".join(f"{x:9.2f}" for x in mean[:4]) + " | " + " ".join(f"{x:9.2f}" for x in (mean[6], mean[4], mean[5])) +
          f"   [{o[m, 4].max():.2f} {o[m, 6].max():.2f}]")
