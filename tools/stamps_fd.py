"""s_memtime stamps inside k_factor_diag (instrumented build, -DHQPKKT_STAMPS) for the last
launch of a C2 factorisation (the root supernode, p = 80):
   HQPKKT_LIB=hqp_amd/libhqpkkt_stamps.so python3 tools/stamps_fd.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hqp_amd import problems, ipmatrix, _lib
prog = problems.banded_qp(40000, 80)
M = ipmatrix.IpSpBKP()
M.init(prog)
st = problems.ip_state(prog, seed=1)
for rep in range(2):
    M.factor(prog, st[0], st[1])
    out = (C.c_int * 64)()
    _lib.lib().hqpkkt_debug_stamps(M._h, out)
    s = np.array(out[9:9 + 50], dtype=np.int64)
    p = out[9 + 50]
    npan = (p + 15) // 16
    print("p", p, "stage", s[1] - s[0])
    for k in range(npan):
        a = s[2 + 4 * k: 6 + 4 * k]
        print("  panel", k, "publish", a[0] - (s[1] if k == 0 else s[5 + 4 * (k - 1)]), "wave", a[1] - a[0], "sync", a[2] - a[1], "sweep", a[3] - a[2])
    print("  tail:", "to 44", s[44] - s[5 + 4 * (npan - 1)], "44-45", s[45] - s[44], "45-46", s[46] - s[45], "46-47", s[47] - s[46], "47-48", s[48] - s[47], "total", s[48] - s[0])
