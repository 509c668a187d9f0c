// libhqpkkt_rccl.so: see include/hqpkkt_rccl.h.
#include "../../include/hqpkkt_rccl.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

namespace {
struct Ctx {
  ncclComm_t comm = nullptr;
  int rank = 0, nranks = 1, device = 0;
};
static_assert(sizeof(ncclUniqueId) <= HQPKKT_RCCL_ID_BYTES, "ncclUniqueId does not fit");
int env_int(const char *a, const char *b, const char *c, int dflt) {
  for (const char *k : {a, b, c}) {
    const char *v = k ? getenv(k) : nullptr;
    if (v && *v) return atoi(v);
  }
  return dflt;
}
}  // namespace

extern "C" {

int hqpkkt_rccl_unique_id(char id[HQPKKT_RCCL_ID_BYTES]) {
  if (!id) return -1;
  ncclUniqueId u;
  std::memset(id, 0, HQPKKT_RCCL_ID_BYTES);
  const ncclResult_t r = ncclGetUniqueId(&u);
  if (r != ncclSuccess) return (int)r;
  std::memcpy(id, &u, sizeof(u));
  return 0;
}

int hqpkkt_rccl_create(const char id[HQPKKT_RCCL_ID_BYTES], int nranks, int rank, int device, void **ctx) {
  if (!id || !ctx || nranks < 1 || rank < 0 || rank >= nranks) return -1;
  Ctx *c = new (std::nothrow) Ctx;
  if (!c) return -1;
  c->rank = rank, c->nranks = nranks, c->device = device;
  if (hipSetDevice(device) != hipSuccess) {
    delete c;
    return -2;
  }
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof(u));
  const ncclResult_t r = ncclCommInitRank(&c->comm, nranks, u, rank);
  if (r != ncclSuccess) {
    delete c;
    return (int)r;
  }
  *ctx = c;
  return 0;
}

int hqpkkt_rccl_create_from_env(void **ctx, int *rank_out, int *nranks_out, int *device_out) {
  const int rank = env_int("HQPKKT_RANK", "RANK", "OMPI_COMM_WORLD_RANK", 0);
  const int nranks = env_int("HQPKKT_WORLD_SIZE", "WORLD_SIZE", "OMPI_COMM_WORLD_SIZE", 1);
  const int device = env_int("HQPKKT_DEVICE", "LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", rank);
  std::string path = getenv("HQPKKT_ID_FILE") ? getenv("HQPKKT_ID_FILE")
                                              : std::string("/tmp/hqpkkt_rccl_id.") + (getenv("MASTER_PORT") ? getenv("MASTER_PORT") : "0");
  char id[HQPKKT_RCCL_ID_BYTES];
  if (rank == 0) {
    int e = hqpkkt_rccl_unique_id(id);
    if (e) return e;
    const std::string tmp = path + ".tmp";
    FILE *f = std::fopen(tmp.c_str(), "wb");
    if (!f) return -3;
    const size_t k = std::fwrite(id, 1, sizeof(id), f);
    std::fclose(f);
    if (k != sizeof(id) || std::rename(tmp.c_str(), path.c_str()) != 0) return -3;
  } else {
    FILE *f = nullptr;
    for (int tries = 0; tries < 6000 && !f; tries++) {  // up to ten minutes
      f = std::fopen(path.c_str(), "rb");
      if (!f) usleep(100000);
    }
    if (!f) return -4;
    const size_t k = std::fread(id, 1, sizeof(id), f);
    std::fclose(f);
    if (k != sizeof(id)) return -4;
  }
  const int e = hqpkkt_rccl_create(id, nranks, rank, device, ctx);
  if (e) return e;
  if (rank_out) *rank_out = rank;
  if (nranks_out) *nranks_out = nranks;
  if (device_out) *device_out = device;
  return 0;
}

int hqpkkt_rccl_exchange(void *ctx, int op, double *buf, long long slot_elems, int nslots, void *hip_stream) {
  Ctx *c = (Ctx *)ctx;
  if (!c || !buf || slot_elems < 0) return -1;
  hipStream_t s = (hipStream_t)hip_stream;
  ncclResult_t r;
  if (op == 0) {  // HQPKKT_XCHG_ALLGATHER, in place: the send part is this rank's slot of the receive buffer
    if (nslots != c->nranks) return -1;
    r = ncclAllGather(buf + (size_t)c->rank * slot_elems, buf, (size_t)slot_elems, ncclDouble, c->comm, s);
  } else if (op == 1) {  // HQPKKT_XCHG_ALLREDUCE_SUM
    r = ncclAllReduce(buf, buf, (size_t)slot_elems, ncclDouble, ncclSum, c->comm, s);
  } else if (op >= 16 && op < 16 + c->nranks) {  // HQPKKT_XCHG_BCAST_BASE + root
    // the broadcasts of one gather come back to back: the first opens a group, the last (root = nranks-1) closes it
    const int root = op - 16;
    if (root == 0) (void)ncclGroupStart();
    r = slot_elems > 0 ? ncclBroadcast(buf, buf, (size_t)slot_elems, ncclDouble, root, c->comm, s) : ncclSuccess;
    if (root == c->nranks - 1) {
      const ncclResult_t g = ncclGroupEnd();
      if (r == ncclSuccess) r = g;
    }
  } else
    return -1;
  return r == ncclSuccess ? 0 : (int)r;
}

int hqpkkt_rccl_destroy(void *ctx) {
  Ctx *c = (Ctx *)ctx;
  if (!c) return 0;
  if (c->comm) (void)ncclCommDestroy(c->comm);
  delete c;
  return 0;
}

}  // extern "C"
