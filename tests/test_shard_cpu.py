"""CPU: the one-system shard plan (hqpkkt_set_shard) and its exchange
steps.  No GPU: the plan is host code, the two collectives run over gloo with
world_size 2, and the numeric kernels are stood in for by the numpy model of the
supernodal LDL' (tests/model.py) working on the structure the library exports."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from common import load_golden
from hqp_amd import ipmatrix, problems

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLS = {"SpBKP": ipmatrix.IpSpBKP, "RedSpBKP": ipmatrix.IpRedSpBKP}


def plan(cls, prog, rank, count, **kw):
    M = cls(shard=(rank, count, lambda *a: None), **kw)
    try:
        M.init(prog)  # analysis is host code; the value upload needs a device
    except ipmatrix.KktError as e:
        assert e.code == 100
    return M


@pytest.mark.parametrize("count", [2, 3, 8])
@pytest.mark.parametrize("kind", ["SpBKP", "RedSpBKP"])
def test_plan_partitions_the_tree(kind, count):
    prog = problems.banded_qp(3000, 16, 3)
    owners, stats = [], []
    for rank in range(count):
        M = plan(CLS[kind], prog, rank, count)
        owners.append(M.debug(10))
        stats.append(M.stats())
        s = M.structure()
    own = owners[0]
    assert all(np.array_equal(own, o) for o in owners)  # every rank derives the same plan
    par = s["parent"]
    top = own < 0
    assert top.any() and set(np.unique(own[~top])) == set(range(count))
    for k in range(len(par)):
        if par[k] >= 0:
            if top[k]:
                assert top[par[k]]  # the replicated part is closed under "parent"
            elif not top[par[k]]:
                assert own[par[k]] == own[k]  # subtrees are not split
    xr = s["exchange_roots"]
    assert sorted(xr) == sorted(k for k in range(len(par)) if not top[k] and par[k] >= 0 and top[par[k]])
    st0 = stats[0]
    assert sum(st["flops_local"] for st in stats) + st0["flops_top"] == st0["flops_factor"]
    assert max(st["flops_local"] for st in stats) <= 1.25 * st0["flops_factor"] / count
    assert st0["n_top"] == int(top.sum()) and st0["n_exchange_blocks"] == len(xr)
    assert st0["bytes_exchange_factor"] > 0 and st0["bytes_exchange_step"] >= 8 * st0["dim"]


def test_single_rank_plan_is_trivial():
    prog = problems.banded_qp(600, 10, 1)
    M = plan(ipmatrix.IpSpBKP, prog, 0, 1)
    assert (M.debug(10) == 0).all() and len(M.debug(11)) == 0
    st = M.stats()
    assert st["n_top"] == 0 and st["flops_local"] == st["flops_factor"] and st["bytes_exchange_step"] == 0


def test_set_shard_argument_checks():
    M = ipmatrix.IpSpBKP()
    with pytest.raises(ipmatrix.KktError):
        M.set_shard(2, 2, lambda *a: None)  # rank out of range
    with pytest.raises(ipmatrix.KktError):
        M.set_shard(0, 2, None)  # several ranks need an exchange function
    M.set_shard(0, 1, None)


WORKER = textwrap.dedent("""
    import json, os, sys
    import numpy as np
    sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
    import torch, torch.distributed as tdist
    from hqp_amd import dist, ipmatrix
    from common import load_golden
    import model
    rank, _lr, world = dist.init(backend="gloo")

    # 1. the two collectives on host tensors
    slot, t = 5, torch.full((5 * world + 3,), -1.0, dtype=torch.float64)
    t[rank * slot:(rank + 1) * slot] = torch.arange(slot, dtype=torch.float64) + 10 * rank
    dist.exchange_tensor(dist.XCHG_ALLGATHER, t, slot, world, rank)
    want = torch.cat([torch.arange(slot, dtype=torch.float64) + 10 * r for r in range(world)])
    ok_gather = bool(torch.equal(t[:slot * world], want)) and bool((t[slot * world:] == -1).all())
    u = torch.arange(7, dtype=torch.float64) * (rank + 1)
    dist.exchange_tensor(dist.XCHG_ALLREDUCE_SUM, u, 6, 1, rank)
    tot = sum(range(1, world + 1))
    ok_reduce = bool(torch.equal(u[:6], torch.arange(6, dtype=torch.float64) * tot)) and float(u[6]) == 6.0 * (rank + 1)

    # 2. the sharded elimination, numpy model in place of the kernels
    out = dict(ok_gather=ok_gather, ok_reduce=ok_reduce, cases=[])
    for name, kind in (("banded_n300_b10", "SpBKP"), ("banded_n300_b10", "RedSpBKP"), ("did_K50_spread4", "SpBKP")):
        prog, st, _g = load_golden(name)
        cls = ipmatrix.IpSpBKP if kind == "SpBKP" else ipmatrix.IpRedSpBKP
        M = cls(shard=(rank, world, lambda *a: None), leaf_size=24, max_pivots=12)
        try:
            M.init(prog)
        except ipmatrix.KktError:
            pass
        s = M.structure()
        own = s["node_owner"]
        mine = [k for k in range(len(own)) if own[k] == rank]
        top = [k for k in range(len(own)) if own[k] < 0]
        topidx = np.concatenate([np.arange(s["piv_start"][k], s["piv_start"][k] + s["npiv"][k]) for k in top])
        K, _sc = model.scaled_kkt(prog, st[0], st[1], 0 if kind == "SpBKP" else 1)
        dim = K.shape[0]
        e = s["elim"]
        mdl = model.Model(s)
        mdl.begin(K, prog.n)
        S0 = mdl.S[np.ix_(topidx, topidx)].copy()
        mdl.eliminate(mine)                                   # factor phase 1
        delta = torch.from_numpy(mdl.S[np.ix_(topidx, topidx)] - S0)
        tdist.all_reduce(delta)                               # = all-gather + extend-add of the update blocks
        mdl.S[np.ix_(topidx, topidx)] = S0 + delta.numpy()
        for k in range(len(own)):                             # the other ranks' subtrees are eliminated too
            if own[k] >= 0:
                mdl.done[s["piv_start"][k]:s["piv_start"][k] + s["npiv"][k]] = True
        mdl.eliminate(top)                                    # factor phase 2 (replicated)
        rhs = np.random.default_rng(5).uniform(-1, 1, dim)
        x = np.zeros(dim); x[e] = rhs
        r0 = x[topidx].copy()
        mdl.forward(x, mine)                                  # step phase 1
        dx = torch.from_numpy(x[topidx] - r0)
        tdist.all_reduce(dx)                                  # contribution vectors
        x[topidx] = r0 + dx.numpy()
        mdl.forward(x, top); mdl.backward(x, top); mdl.backward(x, mine)   # step phase 2
        keep = np.zeros(dim, dtype=bool)
        for k in mine + (top if rank == 0 else []):
            keep[s["piv_start"][k]:s["piv_start"][k] + s["npiv"][k]] = True
        xs = torch.from_numpy(np.where(keep, x, 0.0))
        dist.exchange_tensor(dist.XCHG_ALLREDUCE_SUM, xs, dim, 1, rank)    # the solution
        sol = xs.numpy()[e]
        err = float(np.abs(K @ sol - rhs).max() / max(1.0, np.abs(K).max() * np.abs(sol).max()))
        full = model.Model(s); full.factor(K, prog.n)
        xr = np.zeros(dim); xr[e] = rhs
        ref = full.solve(xr)[e]
        out["cases"].append(dict(name=name, kind=kind, err=err, viol=mdl.struct_violation, npert=mdl.npert,
                                 diff=float(np.abs(sol - ref).max() / np.abs(ref).max()), ntop=len(top), nmine=len(mine)))
    res = [None] * world
    tdist.all_gather_object(res, out)
    if rank == 0:
        print("RESULT " + json.dumps(res))
    dist.finalize()
""") % (ROOT, ROOT)


def test_two_ranks_gloo_exchange_and_sharded_elimination(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29547", str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1]
    res = json.loads(line[len("RESULT "):])
    assert len(res) == 2
    for r in res:
        assert r["ok_gather"] and r["ok_reduce"]
        for c in r["cases"]:
            assert c["viol"] == 0.0 and c["ntop"] > 0 and c["nmine"] > 0, c
            # same elimination as the unsharded model; one solve, no refinement
            assert c["diff"] < 1e-9, c
            assert c["err"] < (1e-3 if c["npert"] else 1e-7), c
