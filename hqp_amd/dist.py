"""One process per GPU over torch.distributed (backend "nccl" is RCCL on ROCm;
"gloo" on CPU for the tests).

The KKT path shards by KKT SYSTEM: independent interior-point problems (scenarios,
MPC instances, the QPs of separate SQP runs) are dealt round-robin to the ranks
and need no data-path collective; the only collectives are the barrier and the
max-over-ranks of the wall time that the benchmark contract asks for.  (Sharding
ONE system over GPUs - subtrees of the assembly tree per rank, separator update
matrices exchanged over xGMI - is the next step)
"""
from __future__ import annotations

import os


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None):
    """Initialise the default process group from the torchrun environment.
    Returns (rank, local_rank, world); a no-op group-less (0, 0, 1) for one process."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = env_world()
    if world == 1:
        return rank, local_rank, world
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)
    return rank, local_rank, world


def shard_units(n_units, rank, world):
    """Indices of the KKT systems this rank owns (round robin)."""
    return list(range(rank, n_units, world))


def fence(device_sync=True):
    """barrier + device synchronise, the bracket of the timed region."""
    import torch
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    if device_sync and torch.cuda.is_available():
        torch.cuda.synchronize()


def max_over_ranks(value):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def finalize():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
