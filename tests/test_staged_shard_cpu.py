"""CPU (no GPU): the N > 1 path of the STAGED engine with world_size 2, 3, 4 and 8 over gloo.  The library's plan of the
memory-sharded partition drives a numpy restatement of the stage's data flow with REAL collectives
(tests/staged_shard_cpu_worker.py); separately the plan's invariants for 2 ... 8 ranks: the blocks cover the lower
triangle of G_xx exactly once, every rank owns the same number of them, the tile lists are the owners' blocks."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from hqp_amd import ipmatrix

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,nx", [(2, 300), (3, 300), (3, 140), (4, 520), (8, 1100)])  # (8 ranks: the pairs P / 2 apart are cut in two)
def test_sharded_stage_data_flow_over_gloo(world, nx):
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", NX=str(nx))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "staged_shard_cpu_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    line = [l for l in out.stdout.splitlines() if l.startswith("STAGED_SHARD_CPU ")][-1]
    recs = json.loads(line[len("STAGED_SHARD_CPU "):])
    assert len(recs) == world
    for r in recs:
        assert r["worst"] < 1e-12, r  # every rank ends with the V_k of the recursion on whole matrices
        assert r["cuts"] == recs[0]["cuts"] and r["nrects"] == recs[0]["nrects"]


@pytest.mark.parametrize("world", [2, 3, 4, 5, 8])
def test_blocks_of_gxx_cover_the_triangle_once(world):
    K, nx, nu = 2, 5000, 50
    n = K * (nx + nu) + nx
    Q = (np.arange(n + 1, dtype=np.int32), np.arange(n, dtype=np.int32))
    E = (np.arange(nx + 1, dtype=np.int32), np.arange(nx, dtype=np.int32))
    nxa, nua = np.full(K + 1, nx, np.int32), np.full(K, nu, np.int32)
    T = (nx + 127) // 128
    cover = np.zeros((T, T), int)
    per_rank_tiles, rects0 = [], None
    for rank in range(world):
        M = ipmatrix.IpLQDOCP(shard=(rank, world, lambda *a: None))
        e = M._L.hqpkkt_analyze_staged(M._h, K, C.c_void_p(nxa.ctypes.data), C.c_void_p(nua.ctypes.data), n, nx, 0,
                                       C.c_void_p(Q[0].ctypes.data), C.c_void_p(Q[1].ctypes.data),
                                       C.c_void_p(E[0].ctypes.data), C.c_void_p(E[1].ctypes.data), None, None)
        assert e == 0
        rects = M.debug(33).reshape(-1, 10)
        if rects0 is None:
            rects0 = rects
        assert np.array_equal(rects, rects0)  # every rank derives the same table
        cut = M.debug(27).reshape(K + 1, world + 1)[0]
        tl = M.debug(34)
        tiles = tl[1:1 + tl[0]]
        per_rank_tiles.append(len(tiles))
        for t in tiles:  # work orientation: rows = the rank's own columns; the block is stored in lower orientation
            tm, tn = (int(t) >> 16) + cut[rank] // 128, int(t) & 0xffff
            cover[max(tm, tn), min(tm, tn)] += 1
    low = np.tril(np.ones((T, T), bool))
    assert np.all(cover[low] == 1) and np.all(cover[~low] == 0), cover
    # slots inside an owner do not overlap, and all boundaries are tile boundaries
    r0 = rects0[rects0[:, 0] == 0]
    for o in range(world):
        mine = sorted((int(r[9]), int((r[4] - r[3]) * (r[6] - r[5]))) for r in r0 if r[7] == o)
        for (a, la), (b, _lb) in zip(mine, mine[1:]):
            assert a + la <= b
    assert np.all(r0[:, 3] % 128 == 0) and np.all(r0[:, 5] % 128 == 0)
    assert max(per_rank_tiles) <= 1.12 * (sum(per_rank_tiles) / world) + 8, per_rank_tiles  # (the last strip is the narrow one)
