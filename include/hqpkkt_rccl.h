/*
 * hqpkkt_rccl.h -- the collectives of a KKT system that is sharded over several GPUs
 * (hqpkkt_set_shard_stream, include/hqpkkt.h) on RCCL over xGMI: one process per GPU, one
 * communicator per process, ncclAllGather / ncclAllReduce in the handle's HIP stream.
 *
 * The reference has no counterpart (hqp/Hqp_Client.C is an unimplemented stub).
 * libhqpkkt_rccl.so is separate from libhqpkkt.so so that the latter keeps depending on the HIP
 * runtime only.  It is NOT linked against librccl: RCCL's entry points are resolved at run time from the
 * RCCL that is already in the process (PyTorch's wheel brings its own librccl.so - a second copy from
 * /opt/rocm would be a second communicator runtime on the same GPUs), else from librccl.so.1 on the
 * library path (a C++ host), or from the file HQPKKT_RCCL_LIB names.  Every entry point returns -5 when
 * no RCCL can be loaded.
 */
#ifndef HQPKKT_RCCL_H
#define HQPKKT_RCCL_H

#ifdef __cplusplus
extern "C" {
#endif

#define HQPKKT_RCCL_ID_BYTES 128

/* rank 0: a fresh ncclUniqueId, to be handed to the other ranks by whatever the host has
 * (torch.distributed broadcast, MPI_Bcast, a file) */
int hqpkkt_rccl_unique_id(char id[HQPKKT_RCCL_ID_BYTES]);
/* every rank: ncclCommInitRank on HIP device `device` */
int hqpkkt_rccl_create(const char id[HQPKKT_RCCL_ID_BYTES], int nranks, int rank, int device, void **ctx);
/* the same from the environment, for hosts without a transport of their own (the C++ HQP host with
 * mat_ngpu > 1, started once per GPU): HQPKKT_RANK / HQPKKT_WORLD_SIZE (default: RANK / WORLD_SIZE of
 * torchrun, OMPI_COMM_WORLD_RANK / _SIZE of mpirun), device = HQPKKT_DEVICE or LOCAL_RANK or the rank;
 * rank 0 writes the id to the file HQPKKT_ID_FILE (default: rccl_id.<MASTER_PORT or 0>.<TORCHELASTIC_RUN_ID or
 * none>.<HQPKKT_RUN_NONCE or the pid of the ranks' parent process> inside $XDG_RUNTIME_DIR, or inside /tmp/hqpkkt-<uid>, mode 0700), the others wait for it.  Rank 0
 * removes the file before it writes the new id and again once the communicator is up; the others take only a
 * file of their own user that is not older than their own start: the file of an earlier run is never used */
int hqpkkt_rccl_create_from_env(void **ctx, int *rank, int *nranks, int *device);
/* what the communicator itself reports: ncclCommCount / ncclCommUserRank / ncclCommCuDevice (any may be null) */
int hqpkkt_rccl_comm_info(void *ctx, int *nranks, int *rank, int *device);
/* hqpkkt_exchange_stream_fn: HQPKKT_XCHG_ALLGATHER in place (slot `rank` of `buf` is the send part),
 * HQPKKT_XCHG_ALLREDUCE_SUM in place, HQPKKT_XCHG_BCAST_BASE + root: the broadcasts of one gather, roots
 * 0 .. nranks-1 in this order, form ONE RCCL group (opened by root 0, closed by the last root, and closed on
 * any error before the call returns); returns 0 or the ncclResult_t */
int hqpkkt_rccl_exchange(void *ctx, int op, double *buf, long long slot_elems, int nslots, void *hip_stream);
int hqpkkt_rccl_destroy(void *ctx);
/* where RCCL's entry points came from: "already loaded in the process", "loaded from the library path",
 * "HQPKKT_RCCL_LIB", "no librccl found" */
const char *hqpkkt_rccl_origin(void);

#ifdef __cplusplus
}
#endif
#endif
