"""s_memtime stamps inside k_st_small (instrumented build, -DHQPKKT_STAMPS; 100 MHz clock):
   HQPKKT_LIB=hqp_amd/libhqpkkt_stamps.so python3 tools/stamps_stsmall.py nx nu"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hqp_amd import problems, ipmatrix, _lib
nx, nu = int(sys.argv[1]), int(sys.argv[2])
prog = problems.lq_docp(3, nx, nu, seed=2)
st = problems.ip_state(prog, 6, 1.0)
M = ipmatrix.IpLQDOCP()
M.init(prog)
for rep in range(3):
    M.factor(prog, st[0], st[1])
    out = (C.c_int * 64)()
    _lib.lib().hqpkkt_debug_stamps(M._h, out)
    t = np.array(out[9:9 + 7], dtype=np.int64)
    print(nx, nu, "phases (x10 ns): start->(A) end, build K, write Kmat, scale, inverse, write Kinv:", np.diff(t).tolist(), "total", int(t[-1] - t[0]))
    g = (C.c_int * 32)()
    if _lib.lib().hqpkkt_debug_gj_stamps(g) == 0:
        print("   step 10 of the LDS / global inverse (cycles): argmax, exchange, pivot row/col, sweep:", np.diff(np.array(g[0:5], dtype=np.int64)).tolist())
        print("   k_st_rm, workgroup 0 (cycles): K^-1 and K to LDS, Y (and carried rows), three products, store:", np.diff(np.array(g[16:20], dtype=np.int64)).tolist())
