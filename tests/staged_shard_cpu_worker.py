"""Worker of tests/test_staged_shard_cpu.py (no GPU): ONE multistage system over the ranks of a gloo group - the library's
plan of the memory-sharded partition (hqpkkt_analyze_staged with hqpkkt_set_shard: column cuts, the blocks of G_xx and
their owners, the tile lists) drives a numpy restatement of the data flow of staged_stage_sharded (staged_host.hip.h):
every rank holds its column strip of F_k and its row strip of V_k only, W_p = V+ F_p stays local, the F strips are
gathered (hqp_amd.dist.exchange_tensor: the collectives the C-ABI callback runs), the blocks are computed as W_p' F_q by
their owners - in the owner's row strip, transposed where the plan says so -, gathered, and V_k assembled from them.
The result is compared with the recursion on the whole matrices."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hqp_amd import dist as hdist, ipmatrix  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    K, nx, nu = 3, int(os.environ.get("NX", "300")), 4
    n = K * (nx + nu) + nx
    Q = (np.arange(n + 1, dtype=np.int32), np.arange(n, dtype=np.int32))
    E = (np.arange(nx + 1, dtype=np.int32), np.arange(nx, dtype=np.int32))
    nxa, nua = np.full(K + 1, nx, np.int32), np.full(K, nu, np.int32)
    M = ipmatrix.IpLQDOCP(shard=(rank, world, lambda *a: None))
    e = M._L.hqpkkt_analyze_staged(M._h, K, C.c_void_p(nxa.ctypes.data), C.c_void_p(nua.ctypes.data), n, nx, 0,
                                   C.c_void_p(Q[0].ctypes.data), C.c_void_p(Q[1].ctypes.data),
                                   C.c_void_p(E[0].ctypes.data), C.c_void_p(E[1].ctypes.data), None, None)
    assert e == 0
    cuts = M.debug(27).reshape(K + 1, world + 1)
    rects = M.debug(33).reshape(-1, 10)
    tl = M.debug(34)
    tiles, p = [], 0
    for k in range(K):
        tiles.append(tl[p + 1:p + 1 + tl[p]])
        p += 1 + tl[p]
    # the data: the same on every rank (seeded), every rank KEEPS its strips only
    rng = np.random.default_rng(5)
    F = [rng.uniform(-1, 1, (nx, nx + nu)) * 0.1 for _ in range(K)]
    H = [np.diag(1.0 + rng.uniform(0, 1, nx + nu)) for _ in range(K)]
    VK = np.diag(1.0 + rng.uniform(0, 1, nx))
    # reference: the recursion on whole matrices
    V = VK.copy()
    Vref = [None] * K + [V]
    for k in range(K - 1, -1, -1):
        G = F[k].T @ V @ F[k] + H[k]
        Gxx, Gux, Guu = G[:nx, :nx], G[nx:, :nx], G[nx:, nx:]
        V = Gxx - Gux.T @ np.linalg.solve(Guu, Gux)
        V = 0.5 * (V + V.T)
        Vref[k] = V
    # sharded: rank `rank`
    Vfull = VK.copy()  # the transient full block of the stage before
    worst = 0.0
    for k in range(K - 1, -1, -1):
        cut = cuts[k]
        c0, c1 = cut[rank], cut[rank + 1]
        Fp, Fu = F[k][:, c0:c1], F[k][:, nx:]          # what this rank holds of F_k
        Wp = Vfull @ Fp                                 # local; NOT exchanged
        # gather of the F strips (padded to the common width, as the slots of the library are)
        xw = int((cut[1:] - cut[:-1]).max())
        slots = torch.zeros(world, nx * xw, dtype=torch.float64)
        slots[rank, :nx * (c1 - c0)] = torch.from_numpy(np.ascontiguousarray(Fp).ravel())
        hdist.exchange_tensor(hdist.XCHG_ALLGATHER, slots.view(-1), nx * xw, world, rank)
        Fg = [slots[q, :nx * (cut[q + 1] - cut[q])].numpy().reshape(nx, cut[q + 1] - cut[q]) for q in range(world)]
        # this rank's blocks, in its row strip of the work block: rows = its own columns
        Gs = np.full((c1 - c0, nx), np.nan)
        for t in tiles[k]:
            tm, tn = int(t) >> 16, int(t) & 0xffff
            r0, r1, j0, j1 = tm * 128, min((tm + 1) * 128, c1 - c0), tn * 128, min((tn + 1) * 128, nx)
            q = int(np.searchsorted(cut, j0, side="right") - 1)
            Gs[r0:r1, j0:j1] = Wp[:, r0:r1].T @ Fg[q][:, j0 - cut[q]:j1 - cut[q]]
        Gs += H[k][c0:c1, :nx]
        # pack (lower orientation; transposed where the block was computed for the partner's rows), gather
        mine = [r for r in rects if r[0] == k and r[7] == rank]
        allr = [r for r in rects if r[0] == k]
        slot_len = max(sum((r[4] - r[3]) * (r[6] - r[5]) + 16 for r in allr if r[7] == o) for o in range(world)) + 16
        xs = torch.full((world, slot_len), float("nan"), dtype=torch.float64)
        for r in mine:
            _k, a, b, r0, r1, cc0, cc1, owner, mine_rows, off = (int(v) for v in r)
            if mine_rows:
                blk = Gs[r0 - c0:r1 - c0, cc0:cc1]
            else:
                blk = Gs[cc0 - c0:cc1 - c0, r0:r1].T
            if a == b:  # (above the diagonal of a diagonal block nothing was computed)
                blk = np.where(np.arange(r0, r1)[:, None] // 128 >= np.arange(cc0, cc1)[None, :] // 128, blk, 0.0)
            xs[rank, off:off + blk.size] = torch.from_numpy(np.ascontiguousarray(blk).ravel())
        hdist.exchange_tensor(hdist.XCHG_ALLGATHER, xs.view(-1), slot_len, world, rank)
        Gxx = np.full((nx, nx), np.nan)
        for r in allr:
            _k, a, b, r0, r1, cc0, cc1, owner, mine_rows, off = (int(v) for v in r)
            Gxx[r0:r1, cc0:cc1] = xs[owner, off:off + (r1 - r0) * (cc1 - cc0)].numpy().reshape(r1 - r0, cc1 - cc0)
        low = np.tril(np.ones((nx, nx), bool))
        assert not np.isnan(Gxx[low]).any(), "the blocks do not cover the lower triangle"
        Gxx = np.where(low, Gxx, 0.0)
        Gxx = Gxx + np.tril(Gxx, -1).T
        # the control-sized chain from the gathered F (replicated)
        Ffull = np.concatenate(Fg + [Fu], axis=1)
        Wu = Vfull @ Fu
        Gu = Wu.T @ Ffull + H[k][nx:, :]
        Vfull = Gxx - Gu[:, :nx].T @ np.linalg.solve(Gu[:, nx:], Gu[:, :nx])
        Vfull = 0.5 * (Vfull + Vfull.T)
        worst = max(worst, float(np.abs(Vfull - Vref[k]).max() / np.abs(Vref[k]).max()))
    out = [None] * world
    dist.all_gather_object(out, dict(rank=rank, worst=worst, nrects=len(rects), cuts=cuts[0].tolist()))
    if rank == 0:
        print("STAGED_SHARD_CPU " + json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
