// libhqpkkt_rccl.so: see include/hqpkkt_rccl.h.
#include "../../include/hqpkkt_rccl.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types only: the entry points are resolved at run time (see rccl_api below)
#include <sys/stat.h>
#include <sys/types.h>
#include <time.h>
#include <unistd.h>

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

namespace {
// RCCL's entry points, taken from the RCCL that is ALREADY in the process when there is one (PyTorch's wheel
// ships its own librccl.so: a second copy from /opt/rocm in the same process would mean two independent
// communicator runtimes on the same GPUs), otherwise from the system's librccl.so.1 (a C++ host without
// PyTorch).  This library is therefore not linked against librccl.
struct RcclApi {
  void *lib = nullptr;
  const char *origin = "";
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*CommCuDevice)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  bool ok = false;
};
const RcclApi &rccl_api() {
  static RcclApi a = [] {
    RcclApi r;
    const char *names[] = {"librccl.so.1", "librccl.so"};
    if (const char *f = getenv("HQPKKT_RCCL_LIB"))  // an explicit path wins
      if (*f) r.lib = dlopen(f, RTLD_NOW | RTLD_LOCAL), r.origin = "HQPKKT_RCCL_LIB";
    for (const char *n : names)
      if (!r.lib && (r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD))) r.origin = "already loaded in the process";
    for (const char *n : names)
      if (!r.lib && (r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) r.origin = "loaded from the library path";
    if (!r.lib) return r;
#define HQPKKT_SYM(field, name) r.field = (decltype(r.field))dlsym(r.lib, name)
    HQPKKT_SYM(GetUniqueId, "ncclGetUniqueId");
    HQPKKT_SYM(CommInitRank, "ncclCommInitRank");
    HQPKKT_SYM(CommDestroy, "ncclCommDestroy");
    HQPKKT_SYM(CommCount, "ncclCommCount");
    HQPKKT_SYM(CommUserRank, "ncclCommUserRank");
    HQPKKT_SYM(CommCuDevice, "ncclCommCuDevice");
    HQPKKT_SYM(AllGather, "ncclAllGather");
    HQPKKT_SYM(AllReduce, "ncclAllReduce");
    HQPKKT_SYM(Broadcast, "ncclBroadcast");
    HQPKKT_SYM(GroupStart, "ncclGroupStart");
    HQPKKT_SYM(GroupEnd, "ncclGroupEnd");
#undef HQPKKT_SYM
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.CommCount && r.CommUserRank && r.CommCuDevice &&
           r.AllGather && r.AllReduce && r.Broadcast && r.GroupStart && r.GroupEnd;
    return r;
  }();
  return a;
}
struct Ctx {
  ncclComm_t comm = nullptr;
  int rank = 0, nranks = 1, device = 0;
  bool group_open = false;  // inside the broadcast sequence of one gather (root 0 .. nranks-1)
};
static_assert(sizeof(ncclUniqueId) <= HQPKKT_RCCL_ID_BYTES, "ncclUniqueId does not fit");
int env_int(const char *a, const char *b, const char *c, int dflt) {
  for (const char *k : {a, b, c}) {
    const char *v = k ? getenv(k) : nullptr;
    if (v && *v) return atoi(v);
  }
  return dflt;
}
// an open group must be closed on every way out: otherwise each later RCCL call of this thread is queued
// and never launched
int close_group(Ctx *c, ncclResult_t r) {
  if (c->group_open) {
    c->group_open = false;
    const ncclResult_t g = rccl_api().GroupEnd();
    if (r == ncclSuccess) r = g;
  }
  return r == ncclSuccess ? 0 : (int)r;
}
// the file that carries the ncclUniqueId of hqpkkt_rccl_create_from_env: HQPKKT_ID_FILE, or a name made of
// the launcher's rendezvous port inside a directory only this user can write
std::string id_file_path() {
  if (const char *f = getenv("HQPKKT_ID_FILE"))
    if (*f) return f;
  std::string dir;
  if (const char *x = getenv("XDG_RUNTIME_DIR"))
    if (*x) dir = x;
  if (dir.empty()) {
    dir = "/tmp/hqpkkt-" + std::to_string((long)getuid());
    if (mkdir(dir.c_str(), 0700) != 0 && errno != EEXIST) return std::string();
    struct stat st;
    if (lstat(dir.c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != getuid() || (st.st_mode & 022)) return std::string();
  }
  const char *port = getenv("MASTER_PORT");
  const char *run = getenv("TORCHELASTIC_RUN_ID");
  // ... and of the process that started the ranks (their common parent under torchrun / mpirun), so that a file a
  // crashed run left behind on the same port is never this run's
  const char *nonce = getenv("HQPKKT_RUN_NONCE");
  return dir + "/rccl_id." + (port && *port ? port : "0") + "." + (run && *run ? run : "none") + "." +
         (nonce && *nonce ? std::string(nonce) : std::to_string((long)getppid()));
}
}  // namespace

extern "C" {

int hqpkkt_rccl_unique_id(char id[HQPKKT_RCCL_ID_BYTES]) {
  if (!id) return -1;
  if (!rccl_api().ok) return -5;  // no usable librccl in the process or on the library path
  ncclUniqueId u;
  std::memset(id, 0, HQPKKT_RCCL_ID_BYTES);
  const ncclResult_t r = rccl_api().GetUniqueId(&u);
  if (r != ncclSuccess) return (int)r;
  std::memcpy(id, &u, sizeof(u));
  return 0;
}

int hqpkkt_rccl_create(const char id[HQPKKT_RCCL_ID_BYTES], int nranks, int rank, int device, void **ctx) {
  if (!id || !ctx || nranks < 1 || rank < 0 || rank >= nranks) return -1;
  if (!rccl_api().ok) return -5;
  Ctx *c = new (std::nothrow) Ctx;
  if (!c) return -1;
  c->rank = rank, c->nranks = nranks, c->device = device;
  if (hipSetDevice(device) != hipSuccess) {
    delete c;
    return -2;
  }
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof(u));
  const ncclResult_t r = rccl_api().CommInitRank(&c->comm, nranks, u, rank);
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
  const std::string path = id_file_path();
  if (path.empty()) return -3;
  // A file left by an earlier run (same port) must never be taken for this run's: rank 0 removes the
  // path before it makes the id and again once the communicator is up (every rank has read it by
  // then); the others accept only a file that is not older than their own start (minus the skew a
  // launcher may put between its ranks)
  const time_t started = time(nullptr);
  char id[HQPKKT_RCCL_ID_BYTES];
  if (rank == 0) {
    (void)unlink(path.c_str());
    int e = hqpkkt_rccl_unique_id(id);
    if (e) return e;
    const std::string tmp = path + ".tmp." + std::to_string((long)getpid());
    (void)unlink(tmp.c_str());
    const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW, 0600);
    if (fd < 0) return -3;
    const ssize_t k = write(fd, id, sizeof(id));
    close(fd);
    if (k != (ssize_t)sizeof(id) || std::rename(tmp.c_str(), path.c_str()) != 0) {
      (void)unlink(tmp.c_str());
      return -3;
    }
  } else {
    bool got = false;
    for (int tries = 0; tries < 6000 && !got; tries++) {  // up to ten minutes
      const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW);
      if (fd >= 0) {
        struct stat st;
        if (fstat(fd, &st) == 0 && st.st_uid == getuid() && st.st_mtime + 120 >= started &&
            read(fd, id, sizeof(id)) == (ssize_t)sizeof(id))
          got = true;
        close(fd);
      }
      if (!got) usleep(100000);
    }
    if (!got) return -4;
  }
  const int e = hqpkkt_rccl_create(id, nranks, rank, device, ctx);
  if (rank == 0) (void)unlink(path.c_str());
  if (e) return e;
  if (rank_out) *rank_out = rank;
  if (nranks_out) *nranks_out = nranks;
  if (device_out) *device_out = device;
  return 0;
}

int hqpkkt_rccl_comm_info(void *ctx, int *nranks, int *rank, int *device) {
  Ctx *c = (Ctx *)ctx;
  if (!c || !c->comm) return -1;
  int n = 0, r = 0, d = 0;
  ncclResult_t e = rccl_api().CommCount(c->comm, &n);
  if (e == ncclSuccess) e = rccl_api().CommUserRank(c->comm, &r);
  if (e == ncclSuccess) e = rccl_api().CommCuDevice(c->comm, &d);
  if (e != ncclSuccess) return (int)e;
  if (nranks) *nranks = n;
  if (rank) *rank = r;
  if (device) *device = d;
  return 0;
}

int hqpkkt_rccl_exchange(void *ctx, int op, double *buf, long long slot_elems, int nslots, void *hip_stream) {
  Ctx *c = (Ctx *)ctx;
  if (!c) return -1;
  if (!buf || slot_elems < 0) return close_group(c, ncclInvalidArgument);
  hipStream_t s = (hipStream_t)hip_stream;
  ncclResult_t r;
  if (op == 0) {  // HQPKKT_XCHG_ALLGATHER, in place: the send part is this rank's slot of the receive buffer
    if (c->group_open || nslots != c->nranks) return close_group(c, ncclInvalidArgument);
    r = rccl_api().AllGather(buf + (size_t)c->rank * slot_elems, buf, (size_t)slot_elems, ncclDouble, c->comm, s);
  } else if (op == 1) {  // HQPKKT_XCHG_ALLREDUCE_SUM
    if (c->group_open) return close_group(c, ncclInvalidArgument);
    r = rccl_api().AllReduce(buf, buf, (size_t)slot_elems, ncclDouble, ncclSum, c->comm, s);
  } else if (op >= 16 && op < 16 + c->nranks) {  // HQPKKT_XCHG_BCAST_BASE + root
    // the broadcasts of one gather come back to back, roots 0 .. nranks-1 in this order: the first opens a
    // group, the last closes it; anything out of sequence, or any error, closes the group before returning
    const int root = op - 16;
    if (root == 0) {
      if (c->group_open) return close_group(c, ncclInvalidUsage);
      r = rccl_api().GroupStart();
      if (r != ncclSuccess) return (int)r;
      c->group_open = true;
    } else if (!c->group_open)
      return (int)ncclInvalidUsage;
    r = slot_elems > 0 ? rccl_api().Broadcast(buf, buf, (size_t)slot_elems, ncclDouble, root, c->comm, s) : ncclSuccess;
    if (r != ncclSuccess || root == c->nranks - 1) return close_group(c, r);
  } else
    return close_group(c, ncclInvalidArgument);
  return r == ncclSuccess ? 0 : (int)r;
}

const char *hqpkkt_rccl_origin(void) { return rccl_api().ok ? rccl_api().origin : "no librccl found"; }

int hqpkkt_rccl_destroy(void *ctx) {
  Ctx *c = (Ctx *)ctx;
  if (!c) return 0;
  (void)close_group(c, ncclSuccess);
  if (c->comm) (void)rccl_api().CommDestroy(c->comm);
  delete c;
  return 0;
}

}  // extern "C"
