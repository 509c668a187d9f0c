"""Cost of Hqp_Solver::update() (new values on the analysed pattern, once per SQP iteration) at C2
size: the reference's own plugin, ours through the shim (threaded walk of the row lists into the
library's pinned staging + one DMA per block), ours with mat_update_threads 0 (round 1's path).
python tools/update_time.py [n band]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hqp_amd import problems
from oracle import refapi

n, band = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (40000, 80)
prog = problems.banded_qp(n, band)
nnz = len(prog.Q[2]) + len(prog.A[2]) + len(prog.C[2])
out = {"n": n, "band": band, "nnz_values": nnz, "value_bytes": 8 * nnz, "host_cores": os.cpu_count()}
for name, host in (("SpBKP", "hip"), ("SpBKPHip", "hip"), ("RedSpBKPHip", "hip")):
    t, t0 = refapi.time_update(prog, name, host=host, reps=9)
    out[name] = {"update_s": t, "init_plus_first_update_s": t0}
print(json.dumps(out))
