"""The pieces ONE rank of a system sharded over P ranks runs per stage, timed on the one GPU of the test box: the
handle is made for rank `r` of `P` with a transport that moves nothing (the other ranks' slots of the exchange buffers
hold zeros: the numbers are meaningless, the shapes, launches and memory are exactly the rank's).  Run under
rocprofv3 --kernel-trace --stats for the per-kernel durations; alone it prints the factorisation's wall time per
stage and the arenas' sizes.  python tools/shard_pieces.py P r [K nx nu]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from hqp_amd import ipmatrix

P, r = int(sys.argv[1]), int(sys.argv[2])
K, nx, nu = (int(a) for a in sys.argv[3:6]) if len(sys.argv) > 5 else (8, 5000, 50)
dq = bench.c4_dense(K, nx, nu, seed=0)
shard = (r, P, lambda *a: None) if P > 1 else None
M = ipmatrix.IpLQDOCP(device_vectors=True, **({"shard": shard} if shard else {}))
M.init_dense(dq)
dq.F = None
torch.cuda.empty_cache()
n, me, m = dq.dims
g = torch.Generator(device="cuda").manual_seed(1)
z = torch.empty(m, dtype=torch.float64, device="cuda").uniform_(0.1, 1.1, generator=g)
w = torch.empty(m, dtype=torch.float64, device="cuda").uniform_(0.1, 1.1, generator=g)
ts = []
for it in range(4):
    torch.cuda.synchronize()
    t0 = time.time()
    try:
        M.factor(None, z, w)
    except ipmatrix.KktError:
        pass  # (zeros where the other ranks' blocks would be: the control-sized chain may call the stage singular)
    torch.cuda.synchronize()
    ts.append(time.time() - t0)
s = M.stats()
print(json.dumps(dict(ranks=P, rank=r, K=K, nx=nx, nu=nu, ms_per_stage=round(1e3 * min(ts[1:]) / K, 4), bytes_panels=s["bytes_panels"],
                      bytes_updates=s["bytes_updates"], flops_local=s["flops_local"], bytes_exchange_factor=s["bytes_exchange_factor"])), flush=True)
