"""bench.py's stand-ins for BASELINE configs[4] on their own (mesh 300 x 300, 1000 x 1000, band of 21 with 1000 far
couplings at 10^5 variables): python tools/mesh_bench.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
print(json.dumps(bench.mesh_kkt_sizes(0)), flush=True)
