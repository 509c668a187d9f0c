"""s_memtime stamps inside k_factor_diag_small<true> (instrumented build, -DHQPKKT_STAMPS):
   HQPKKT_LIB=hqp_amd/libhqpkkt_stamps.so python3 tools/stamps_small.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hqp_amd import problems, ipmatrix, _lib

prog = problems.did_like_qp(int(sys.argv[1]) if len(sys.argv) > 1 else 2000)
M = ipmatrix.IpRedSpBKP()
M.init(prog)
z, w, r1, r2, r3, r4 = problems.ip_state(prog, seed=3)
for rep in range(3):
    M.factor(prog, z, w)
    out = (C.c_int * 64)()
    _lib.lib().hqpkkt_debug_stamps(M._h, out)
    st = np.array(out[9:9 + 10], dtype=np.int64)
    print("p", out[9 + 50], "b", out[9 + 51], "cycles:", np.diff(st).tolist(), "total", int(st[-1] - st[0]))
    pv = np.array(out[19:19 + min(out[9 + 50], 20)], dtype=np.int64)
    print("   pivot starts rel. to stamp 2:", (pv - st[2]).tolist(), "loop end", int(st[3] - st[2]))
