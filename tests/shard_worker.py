"""Worker of test_gpu_shard.py / bench.py --shard-system check: every rank factors
and solves the SAME KKT system with one handle sharded over the ranks
(hqpkkt_set_shard) and rank 0 compares with an unsharded handle in-process.
Launched by torchrun; all ranks may share cuda:0 (backend gloo, host-staged
exchange) or own a GPU each (backend nccl = RCCL)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def c4dense_case(case, rank, world, dev):
    """BASELINE configs[3]'s structure at FULL stage width through the dense hand-over (bench.c4_dense:
    K stages of nx states, nu controls, everything on the device): the sharded handle against an
    unsharded one on rank 0, identical vectors on every rank."""
    import torch
    import torch.distributed as tdist
    import bench
    from hqp_amd import dist, ipmatrix
    _, K, nx, nu, _kind = case
    dq = bench.c4_dense(K, nx, nu, seed=0, device=f"cuda:{dev}")
    n, me, m = dq.dims
    g = torch.Generator(device=f"cuda:{dev}").manual_seed(100)
    rnd = lambda k, lo, hi: torch.empty(k, dtype=torch.float64, device=f"cuda:{dev}").uniform_(lo, hi, generator=g)
    z, w = rnd(m, 0.1, 1.1), rnd(m, 0.1, 1.1)
    r = [rnd(k, -0.5, 0.5) for k in (n, me, m, m)]
    if os.environ.get("SHARD_TRANSPORT") == "rccl":
        shard = dist.RcclShard(rank, world, dev)
    else:
        shard = (rank, world, dist.make_exchange(rank, dev))
    d = [torch.zeros(k, dtype=torch.float64, device=f"cuda:{dev}") for k in (n, me, m, m)]
    torch.cuda.synchronize()
    tdist.barrier()
    free0 = torch.cuda.mem_get_info(dev)[0]
    M = ipmatrix.IpLQDOCP(device=dev, device_vectors=True, shard=shard)
    M.init_dense(dq)
    for _rep in range(2):
        M.factor(None, z, w)
        res = M.solve(None, z, w, *r, *d)
    torch.cuda.synchronize()
    tdist.barrier()
    # device memory the handles of ALL ranks on this device took (the ranks of the test box share one GPU)
    hbm_all_ranks = free0 - torch.cuda.mem_get_info(dev)[0]
    tdist.barrier()
    s = M.stats()
    cuts = M.debug(27).reshape(-1, world + 1)
    rec = dict(case=case, rank=rank, res=res, staged=True, cuts=cuts[0].tolist(), flops_local=s["flops_local"],
               bytes_factor=s["bytes_exchange_factor"], ranks=s["shard_count"], refine_rounds=s["refine_rounds"],
               bytes_panels=s["bytes_panels"], hbm_all_ranks=int(hbm_all_ranks), bytes_updates=s["bytes_updates"])
    del M
    torch.cuda.empty_cache()
    tdist.barrier()  # (the unsharded partner below needs the memory the other ranks have just released)
    if rank == 0:
        R = ipmatrix.IpLQDOCP(device=dev, device_vectors=True)
        R.init_dense(dq)
        d0 = [torch.zeros_like(t) for t in d]
        R.factor(None, z, w)
        rec["res_single"] = R.solve(None, z, w, *r, *d0)
        rec["bytes_panels_single"] = R.stats()["bytes_panels"]
        torch.cuda.synchronize()
        rec["hbm_single"] = int(free0 - torch.cuda.mem_get_info(dev)[0])
        rec["diff"] = max(float((a - b).abs().max() / b.abs().max().clamp_min(1e-300)) for a, b in zip(d, d0) if b.numel())
        del R
    t = torch.cat(d).cpu()
    ref = t.clone()
    tdist.broadcast(ref, src=0)
    rec["same_as_rank0"] = bool(torch.equal(t, ref))
    return rec


def main():
    import torch
    import torch.distributed as tdist
    from hqp_amd import dist, ipmatrix, problems
    from common import new_d

    backend = os.environ.get("SHARD_BACKEND", "gloo")
    rank, local_rank, world = dist.env_world()
    dev = local_rank if backend == "nccl" else 0
    torch.cuda.set_device(dev)
    if backend == "nccl":
        tdist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    else:
        tdist.init_process_group("gloo")
    out = []
    cases = json.loads(os.environ.get("SHARD_CASES", '[["banded", 1500, 12, "SpBKP"]]'))
    for case in cases:
        kind = case[-1]
        if case[0] == "c4dense":
            out.append(c4dense_case(case, rank, world, dev))
            continue
        if case[0] == "fuzzst":  # ["fuzzst", first, count, "LQDOCP"]: cases of tools/fuzz_staged.py, sharded against unsharded
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import fuzz_staged
            worst, ncmp, codes = 0.0, 0, []
            for c in range(case[1], case[1] + case[2]):
                prog, st, tag = fuzz_staged.make_case(c)
                res = []
                for sh in (True, False):
                    if not sh and rank != 0:
                        res.append(None)
                        continue
                    kw = dict(shard=(rank, world, dist.make_exchange(rank, dev))) if sh else {}
                    M = ipmatrix.IpLQDOCP(device=dev, **kw)
                    try:
                        M.init(prog)
                        M.factor(prog, st[0], st[1])
                        d = new_d(prog)
                        r = M.solve(prog, *st, *d)
                        res.append((0, r, d))
                    except ipmatrix.KktError as err:
                        res.append((err.code, None, None))
                    del M
                code = torch.tensor([res[0][0]])
                ref = code.clone()
                tdist.broadcast(ref, src=0)
                codes.append(bool(torch.equal(code, ref)))  # every rank ends with the same status
                if rank == 0:
                    (cs, rs, ds), (cu, ru, du) = res
                    if cs == 1:
                        continue  # E_SIZES: an odd number of states - the sharded plan refuses such stages, by design (staged_plan.cpp)
                    if cs != cu:
                        worst = max(worst, 1.0 if (cs == 0) != (cu == 0) and not (cu == 0 and ru is not None and not ru <= 1e-8) else 0.0)
                        continue
                    if cs == 0 and ru <= 1e-10:
                        ncmp += 1
                        scale = max(1.0, max(np.abs(v).max() for v in du if len(v)))
                        worst = max(worst, max(float(np.abs(a - b).max()) for a, b in zip(ds, du) if len(b)) / scale)
            out.append(dict(case=case, rank=rank, worst=worst, compared=ncmp, same_status=all(codes)))
            continue
        if case[0] == "singular":
            # a duplicated equality row: the zero pivot turns up inside ONE rank's subtree (or in the top);
            # every rank must return the same status - nobody may be left waiting in a collective
            prog = problems.banded_qp(case[1], case[2])
            p, i, x = prog.A
            r = case[1] // 8
            lo, hi = p[r], p[r + 1]
            lo2, hi2 = p[r + 1], p[r + 2]
            i, x = i.copy(), x.copy()
            k = min(hi - lo, hi2 - lo2)
            i[lo2:lo2 + k], x[lo2:lo2 + k] = i[lo:lo + k], x[lo:lo + k]
            if hi2 - lo2 > k:
                x[lo2 + k:hi2] = 0.0
            prog = problems.Program(prog.n, prog.me, prog.m, prog.Q, (p, i, x), prog.C)
            st = problems.ip_state(prog, 7, 1.0)
            cls = {"SpBKP": ipmatrix.IpSpBKP, "RedSpBKP": ipmatrix.IpRedSpBKP}[kind]
            M = cls(device=dev, shard=(rank, world, dist.make_exchange(rank, dev)))
            M.init(prog)
            code = 0
            try:
                M.factor(prog, st[0], st[1])
                d = new_d(prog)
                M.solve(prog, *st, *d)
            except ipmatrix.KktError as err:
                code = err.code
            out.append(dict(case=case, rank=rank, code=code))
            continue
        if case[0] == "banded":
            prog = problems.banded_qp(case[1], case[2])
        elif case[0] == "docp":
            prog = problems.lq_docp(case[1], case[2], case[3])
        elif case[0] == "docpx":  # ["docpx", K, nx, nu, {options of problems.lq_docp}, spread, kind]
            prog = problems.lq_docp(case[1], case[2], case[3], **case[4])
        elif case[0] == "grid":  # mesh QP through the tree of the graph's own dissection (opts.ordering)
            prog = problems.grid_sparse_qp(case[1], case[2])
        elif case[0] == "did_spread":  # w/z over 12 decades on weak Hessian diagonals: the refinement fails and the
            prog = problems.did_like_qp(case[1])  # handle switches its zero-diagonal placement (every rank must)
        else:
            prog = problems.did_like_qp(case[1])
        st = problems.ip_state(prog, case[2], case[3]) if case[0] == "did_spread" else problems.ip_state(prog, 7, case[5] if case[0] == "docpx" else 1.0)
        cls = {"SpBKP": ipmatrix.IpSpBKP, "RedSpBKP": ipmatrix.IpRedSpBKP, "LQDOCP": ipmatrix.IpLQDOCP}[kind]
        kw = dict(ordering=case[3]) if case[0] == "grid" else {}
        if os.environ.get("SHARD_TRANSPORT") == "rccl":  # libhqpkkt_rccl.so: stream-ordered collectives
            M = cls(device=dev, shard=dist.RcclShard(rank, world, dev), **kw)
        else:
            M = cls(device=dev, shard=(rank, world, dist.make_exchange(rank, dev)), **kw)
        M.init(prog)
        d = new_d(prog)
        for rep in range(2):  # second round replays the captured graphs
            M.factor(prog, st[0], st[1])
            res = M.solve(prog, *st, *d)
        s = M.stats()
        if kind == "LQDOCP":
            cuts = M.debug(27).reshape(-1, world + 1)
            rec = dict(case=case, rank=rank, res=res, staged=True, cuts=cuts[0].tolist(), flops_local=s["flops_local"],
                       bytes_factor=s["bytes_exchange_factor"], ranks=s["shard_count"], bytes_panels=s["bytes_panels"])
        else:
            owner = M.debug(10)
            rec = dict(case=case, rank=rank, res=res, n_top=s["n_top"], xblocks=s["n_exchange_blocks"],
                       flops_local=s["flops_local"], flops_top=s["flops_top"], nodes=s["n_supernodes"],
                       owned=int((owner == rank).sum()), top=int((owner < 0).sum()),
                       bytes_factor=s["bytes_exchange_factor"], bytes_step=s["bytes_exchange_step"])
        if rank == 0:
            R = cls(device=dev, **kw)
            R.init(prog)
            R.factor(prog, st[0], st[1])
            d0 = new_d(prog)
            res0 = R.solve(prog, *st, *d0)
            rec["res_single"] = res0
            rec["bytes_panels_single"] = R.stats()["bytes_panels"]
            rec["diff"] = max(float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
                              for a, b in zip(d, d0) if len(b))
        # all ranks must hold the same full solution
        t = torch.tensor(np.concatenate(d))
        ref = t.clone()
        tdist.broadcast(ref, src=0)
        rec["same_as_rank0"] = bool(torch.equal(t, ref))
        out.append(rec)
    gathered = [None] * world
    tdist.all_gather_object(gathered, out)
    if rank == 0:
        print("SHARD_RESULT " + json.dumps(gathered))
    tdist.barrier()
    tdist.destroy_process_group()


if __name__ == "__main__":
    main()
