import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from hqp_amd import ipmatrix
for (K, nx, nu) in [(4, 101, 4), (4, 2001, 40), (4, 2001, 400), (4, 2000, 400), (4, 1999, 401)]:
    try:
        dq = bench.c4_dense(K, nx, nu, seed=0)
        M = ipmatrix.IpLQDOCP(device_vectors=True)
        M.init_dense(dq)
        print(K, nx, nu, "ok")
    except Exception as e:
        print(K, nx, nu, "FAIL", e)
