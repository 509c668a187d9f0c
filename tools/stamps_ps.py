"""s_memtime stamps inside k_panel_solve (instrumented build, -DHQPKKT_STAMPS): workgroup 0 of the LAST launch of a C2
factorisation (the top level with a border), per wavefront.
   HQPKKT_LIB=hqp_amd/libhqpkkt_stamps.so python3 tools/stamps_ps.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hqp_amd import problems, ipmatrix, _lib
prog = problems.banded_qp(40000, 80)
M = ipmatrix.IpSpBKP()
M.init(prog)
st = problems.ip_state(prog, seed=1)
for rep in range(3):
    M.factor(prog, st[0], st[1])
w = (C.c_int * 64)()
_lib.lib().hqpkkt_debug_ps_stamps(w)
w = np.array(w[:], dtype=np.int64).reshape(4, 16)
u = lambda a, b: int((b - a) & 0xffffffff)
names = ["node data", "gather", "barrier", "products", "barrier", "store+barrier", "epilogue"]
print("wave " + " ".join(f"{n:>14s}" for n in names) + "   total (shader cycles)")
for wv in range(4):
    d = [u(w[wv, j], w[wv, j + 1]) for j in range(7)]
    print(f"{wv:4d} " + " ".join(f"{x:14d}" for x in d) + f"   {u(w[wv, 0], w[wv, 7])}")
