"""Handle life-cycle stress: many create / analyze / solve / destroy cycles over all entry points
(incl. the policy switch inside a solve); prints the device memory in use before and after."""
import sys, os, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hqp_amd import problems, ipmatrix
free0 = torch.cuda.mem_get_info()[0]
did, banded = problems.did_like_qp(400), problems.banded_qp(2000, 20, 5)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    for kind in (ipmatrix.IpRedSpBKP, ipmatrix.IpSpBKP):
        M = kind()
        M.init(did)
        M.mehrotra(did, init_method=2, hot_start=2)   # switches the zero-diagonal policy on the way
        M.mehrotra(did, hot_start=1)
        M.franke(did, max_iters=300)
        M.franke(did, max_iters=300, hot_start=1)
        B = kind(amalgamation=bool(rep & 1))
        B.init(banded)
        st = problems.ip_state(banded, rep)
        B.factor(banded, st[0], st[1])
        d = [np.zeros(k) for k in (banded.n, banded.me, banded.m, banded.m)]
        assert B.solve(banded, *st, *d) <= 1e-10
        B.init(did)  # re-init with another structure
        B.mehrotra(did)
        del M, B
    # round 2: the STAGED engine (CSR and dense hand-over, a stage with constraints), the graph's own dissection
    docp = problems.lq_docp(6, 40, 3, seed=rep + 1, final_eq=4, path_eq=1, x_bounds=10)
    S = ipmatrix.IpLQDOCP()
    S.init(docp)
    st = problems.ip_state(docp, rep, 1.0)
    d = [np.zeros(k) for k in (docp.n, docp.me, docp.m, docp.m)]
    S.factor(docp, st[0], st[1])
    assert S.solve(docp, *st, *d) <= 1e-10
    S.mehrotra(docp)
    plain = problems.lq_docp(5, 30, 2, seed=rep + 3)
    dq = problems.dense_docp_from_program(plain, [30] * 6, [2] * 5)
    S2 = ipmatrix.IpLQDOCP()
    S2.init_dense(dq)
    st = problems.ip_state(plain, rep, 0.0)
    d = [np.zeros(k) for k in (plain.n, plain.me, plain.m, plain.m)]
    S2.factor(dq, st[0], st[1])
    assert S2.solve(dq, *st, *d) <= 1e-10
    mesh = problems.grid_sparse_qp(30, 30, seed=rep)
    G = ipmatrix.IpRedSpBKP(ordering=1 + (rep & 1))
    G.init(mesh)
    G.mehrotra(mesh)
    del S, S2, G
    gc.collect()
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print("device memory in use: before %.1f MB, after %.1f MB" % ((torch.cuda.mem_get_info()[1] - free0) / 2**20, (torch.cuda.mem_get_info()[1] - free1) / 2**20))
