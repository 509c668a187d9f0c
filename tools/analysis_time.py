import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hqp_amd import problems, ipmatrix
prog = problems.lq_docp(200, int(sys.argv[1]), 10)
M=ipmatrix.IpSpBKP()
t0=time.time()
try: M.init(prog)
except ipmatrix.KktError: pass
print("analyze", round(time.time()-t0,2), M.stats()["nnz_kkt"], M.stats()["n_supernodes"], M.stats()["n_levels"])
