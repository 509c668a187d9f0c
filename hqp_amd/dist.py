"""One process per GPU over torch.distributed (backend "nccl" is RCCL on ROCm;
"gloo" on CPU for the tests).

Two ways to use several GPUs on the KKT path:

* by KKT SYSTEM: independent interior-point problems (scenarios, MPC instances,
  the QPs of separate SQP runs) are dealt round-robin to the ranks and need no
  data-path collective (``shard_units``); the only collectives are the barrier
  and the max-over-ranks of the wall time that the benchmark contract asks for;
* ONE system over the ranks: every rank analyses the same system,
  the symbolic phase deals the subtrees of the assembly tree to the ranks and
  replicates the top separators; one all-gather per factorisation and one
  all-gather + one all-reduce per solve are delegated to ``make_exchange`` below
  through the C-ABI callback ``hqpkkt_set_shard`` (include/hqpkkt.h).
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
    # the device is drained BEFORE the barrier too: the barrier of an RCCL process group is itself a kernel, and it
    # must never meet this library's own collectives (another communicator, libhqpkkt_rccl.so) in flight - kernels of
    # two communicators queued in different orders on different ranks can wait for each other for ever
    if device_sync and torch.cuda.is_available():
        torch.cuda.synchronize()
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


# ----------------------------------------------------------- one system, P ranks
XCHG_ALLGATHER, XCHG_ALLREDUCE_SUM, XCHG_BCAST_BASE = 0, 1, 16


class _DevicePtr:
    """Zero-copy view of device memory owned by libhqpkkt (CUDA array interface)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2}


def exchange_tensor(op, t, slot, nslots, rank, group=None):
    """The two collectives of a sharded system on a 1-D float64 tensor ``t``.

    ALLGATHER: ``t`` holds ``nslots`` slots of ``slot`` values, slot ``rank`` is
    filled; on return all are.  ALLREDUCE_SUM: the first ``slot`` values are
    replaced by their sum over the ranks.  Device tensors go straight to RCCL
    when the group's backend is nccl; with gloo they are staged through the host
    (CPU tests, and two ranks sharing one GPU)."""
    import torch
    import torch.distributed as dist
    direct = (not t.is_cuda) or dist.get_backend(group) == "nccl"
    buf = t if direct else t.cpu()
    if op == XCHG_ALLGATHER:
        slots = buf[: slot * nslots].view(nslots, slot)
        mine = slots[rank].clone()
        if buf.is_cuda:
            dist.all_gather_into_tensor(buf[: slot * nslots], mine, group=group)
        else:
            dist.all_gather(list(slots.unbind(0)), mine, group=group)
    elif op == XCHG_ALLREDUCE_SUM:
        dist.all_reduce(buf[:slot], op=dist.ReduceOp.SUM, group=group)
    elif op >= XCHG_BCAST_BASE:  # slot values that rank (op - base) has filled, to everybody
        dist.broadcast(buf[:slot], src=op - XCHG_BCAST_BASE, group=group)
    else:
        raise ValueError(f"unknown exchange op {op}")
    if not direct:
        t.copy_(buf)
    if t.is_cuda:
        torch.cuda.synchronize(t.device)
    return t


def make_exchange(rank, device=0, group=None):
    """Callable for ``Hqp_IpMatrix(shard=(rank, count, make_exchange(rank, device)))``:
    invoked by the library as ``fn(op, device_pointer, slot_elems, nslots)`` with its
    stream drained; must return after the result is complete."""
    import torch

    def fn(op, ptr, slot, nslots):
        n = slot * nslots
        t = torch.as_tensor(_DevicePtr(ptr, n), device=torch.device("cuda", device))
        exchange_tensor(op, t, slot, nslots, rank, group)

    return fn


class RcclShard:
    """The native transport of a sharded system: one RCCL communicator per process
    (libhqpkkt_rccl.so, include/hqpkkt_rccl.h), its collectives put into the handle's HIP
    stream by the library itself (hqpkkt_set_shard_stream) - no callback into Python, no
    drained stream.  The ncclUniqueId travels over the default torch.distributed group.
    ``Hqp_IpMatrix(shard=RcclShard(rank, world, device))``."""

    def __init__(self, rank, world, device=0, group=None):
        import ctypes as C
        import torch.distributed as dist
        from . import _lib
        R = _lib.rccl_lib()
        uid = C.create_string_buffer(128)
        if rank == 0:
            e = R.hqpkkt_rccl_unique_id(uid)
            if e:
                raise RuntimeError(f"hqpkkt_rccl_unique_id: {e}")
        box = [uid.raw if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        self._ctx = C.c_void_p()
        e = R.hqpkkt_rccl_create(box[0], world, rank, device, C.byref(self._ctx))
        if e:
            raise RuntimeError(f"hqpkkt_rccl_create: {e}")
        self._R, self.rank, self.world = R, rank, world
        self.fn = C.cast(R.hqpkkt_rccl_exchange, C.c_void_p)
        # what the communicator itself says (ncclCommCount / UserRank / CuDevice): a bench line quotes it
        nr, ur, dv = C.c_int(-1), C.c_int(-1), C.c_int(-1)
        if R.hqpkkt_rccl_comm_info(self._ctx, C.byref(nr), C.byref(ur), C.byref(dv)):
            raise RuntimeError("hqpkkt_rccl_comm_info failed")
        self.comm_ranks, self.comm_rank, self.comm_device = nr.value, ur.value, dv.value
        if (self.comm_ranks, self.comm_rank) != (world, rank):
            raise RuntimeError(f"RCCL communicator reports rank {ur.value} of {nr.value}, expected {rank} of {world}")
        # self-test of the three collectives on 8 values per rank (wrong data here must not reach a solve)
        import torch
        dev = torch.device("cuda", device)
        with torch.cuda.device(dev):
            t = torch.zeros(8 * world, dtype=torch.float64, device=dev)
            t[8 * rank:8 * rank + 8] = rank + 1.0
            stream = torch.cuda.current_stream(dev).cuda_stream
            if R.hqpkkt_rccl_exchange(self._ctx, 0, t.data_ptr(), 8, world, stream):
                raise RuntimeError("hqpkkt_rccl_exchange: all-gather failed")
            want = torch.arange(1, world + 1, dtype=torch.float64, device=dev).repeat_interleave(8)
            ok = bool(torch.equal(t, want))
            if R.hqpkkt_rccl_exchange(self._ctx, 1, t.data_ptr(), 8, 1, stream):
                raise RuntimeError("hqpkkt_rccl_exchange: all-reduce failed")
            ok = ok and bool(torch.allclose(t[:8], torch.full((8,), float(world), dtype=torch.float64, device=dev)))
            bs = [torch.full((8,), float(rank), dtype=torch.float64, device=dev) for _ in range(world)]
            for r in range(world):  # one group: opened by root 0, closed by the last root
                if R.hqpkkt_rccl_exchange(self._ctx, 16 + r, bs[r].data_ptr(), 8, 1, stream):
                    raise RuntimeError("hqpkkt_rccl_exchange: broadcast failed")
            torch.cuda.synchronize(dev)
            ok = ok and all(bool((bs[r] == float(r)).all()) for r in range(world))
            if not ok:
                raise RuntimeError("libhqpkkt_rccl.so: self-test of the collectives returned wrong data")

    def close(self):
        if getattr(self, "_ctx", None):
            self._R.hqpkkt_rccl_destroy(self._ctx)
            self._ctx = None
