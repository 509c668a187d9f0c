// C ABI of the MI355X KKT path (include/hqpkkt.h): handle management, device
// residency of the symbolic structure, kernel sequencing for
// assemble -> factor -> step -> residuum -> solve.
#include "../../include/hqpkkt.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <atomic>
#include <mutex>
#include <new>
#include <vector>

#include "analysis.hpp"
#include "staged_plan.hpp"
#include "kernels.hip.h"
#include "factor_blk.hip.h"
#include "solve_top.hip.h"
#ifndef FB_NS160
#define FB_NS160 5  // blocks per block-holding wavefront of k_factor_blk for fronts of 129 .. 160 pivots (7 with FB_OWNSIMD: nine such wavefronts)
#endif
#ifndef FB_OWNSIMD
#define FB_OWNSIMD false  // true: the elimination wavefront of k_factor_blk shares its SIMD with no block-holding wavefront (measured slower)
#endif
#include "ipdriver.hip.h"
#include "staged.hip.h"

using namespace kktdev;

#define HIPCHK(call)                                                              \
  do {                                                                            \
    hipError_t e_ = (call);                                                       \
    if (e_ != hipSuccess) {                                                       \
      std::snprintf(g_last_hip_error, sizeof(g_last_hip_error), "%s:%d %s: %s", __FILE__, \
                    __LINE__, #call, hipGetErrorString(e_));                      \
      return HQPKKT_E_DEVICE;                                                     \
    }                                                                             \
  } while (0)

static char g_last_hip_error[512] = "";

namespace {

template <class T>
struct DBuf {
  T *p = nullptr;
  size_t count = 0;
  int alloc(size_t k) {
    release();
    count = k;
    if (hipMalloc((void **)&p, sizeof(T) * (k ? k : 1)) != hipSuccess) {
      p = nullptr;
      return HQPKKT_E_MEM;
    }
    return 0;
  }
  int upload(const std::vector<T> &v) {
    int e = alloc(v.size());
    if (e) return e;
    if (!v.empty() &&
        hipMemcpy(p, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice) != hipSuccess)
      return HQPKKT_E_DEVICE;
    return 0;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    count = 0;
  }
};

struct CsrBuf {
  DBuf<int> ptr, col, src;
  DBuf<double> val;  // values in CSR order, refreshed by hqpkkt_set_values
  int upload(const Analysis::Csr &c) {
    int e;
    if ((e = ptr.upload(c.ptr)) || (e = col.upload(c.col)) || (e = src.upload(c.src)) || (e = val.alloc(c.src.size())))
      return e;
    return 0;
  }
  CsrDev dev() const { return CsrDev{ptr.p, col.p, src.p, val.p}; }
  void release() { ptr.release(), col.release(), src.release(), val.release(); }
};

}  // namespace

// per-kernel-class device timing (hqpkkt_set_profile): HIP events on the
// handle's stream around every launch, summed per class after the call
enum { KC_ASSEMBLE = 0, KC_FACTOR_DIAG, KC_PANEL_SOLVE, KC_SCHUR_UPDATE,
       KC_SOLVE_FWD, KC_SOLVE_BWD, KC_VECTOR, KC_RESIDUAL, KC_ST_GEMM, KC_ST_SMALL, KC_ST_VEC, KC_ST_GEMM_UPD, KC_XCHG, KC_SOLVE_TOP, KC_COUNT };
static const char *const kc_names[KC_COUNT] = {"assemble", "factor_diag", "panel_solve",
                                               "schur_update", "solve_fwd", "solve_bwd", "vector",
                                               "residual", "staged_gemm", "staged_small", "staged_gemv", "staged_gemm_upd",
                                               "exchange", "solve_top"};
struct Prof {
  bool on = false;
  std::vector<hipEvent_t> pool;
  std::vector<int> cls;
  size_t used = 0;
  double ms[KC_COUNT] = {0};
  long long launches[KC_COUNT] = {0};
  hipEvent_t get() {
    if (used == pool.size()) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return nullptr;
      pool.push_back(e);
    }
    return pool[used++];
  }
  void begin(int c, hipStream_t s) {
    if (!on) return;
    hipEvent_t e = get();
    cls.push_back(c);
    if (e) (void)hipEventRecord(e, s);
  }
  void end(hipStream_t s) {
    if (!on) return;
    hipEvent_t e = get();
    if (e) (void)hipEventRecord(e, s);
  }
  // call after the stream has been synchronised
  void collect() {
    for (size_t k = 0; k + 1 < used && k / 2 < cls.size(); k += 2) {
      float t = 0.f;
      if (hipEventElapsedTime(&t, pool[k], pool[k + 1]) == hipSuccess) {
        ms[cls[k / 2]] += t;
        launches[cls[k / 2]]++;
      }
    }
    used = 0;
    cls.clear();
  }
  void reset() {
    for (int c = 0; c < KC_COUNT; c++) ms[c] = 0, launches[c] = 0;
  }
  void destroy() {
    for (auto e : pool) (void)hipEventDestroy(e);
    pool.clear();
  }
};
#define KLAUNCH(h, c, ...)        \
  do {                            \
    (h)->prof.begin(c, (h)->stream); \
    __VA_ARGS__;                  \
    (h)->prof.end((h)->stream);   \
  } while (0)

struct StagedDev;
static void staged_release(StagedDev *sd, bool destroy);

struct hqpkkt {
  hqpkkt_opts opts;
  Prof prof;
  Analysis an;
  StagedDev *sd = nullptr;  // HQPKKT_MODE_STAGED: the stage blocks (staged_host.hip.h)
  double ge_tol = 1.0e-6;   // rank decision of the stage constraints (_ge_tol, hqp/Hqp_IpLQDOCP.C:113)
  bool analyzed = false, uploaded = false, have_values = false, factored = false;
  hipStream_t own_stream = nullptr, stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, evs0 = nullptr, evs1 = nullptr;
  hipEvent_t evt0 = nullptr, evt1 = nullptr;  // total time of an interior-point run (hqpkkt_mehrotra / _franke)
  hqpkkt_stats st;

  // symbolic structure on the device
  DBuf<int> piv_start, npiv, nbor, parent, bidx, rel, child_ptr, child_idx, ent_a, ent_b,
      term_ptr, diag_ent, q2e, pinv;
  struct DevSched {  // device copy of an Analysis::Sched
    DBuf<int> level_nodes, upd_tiles, slabs, gslabs, cblks;
    void release() {
      level_nodes.release(), upd_tiles.release(), slabs.release();
      gslabs.release(), cblks.release();
    }
  } ds[2];
  DBuf<long long> zero_panel;  // (offset, length) pairs, sharded mode
  DBuf<int> simple_src, simple_wi;  // FULL: compact single-term records of the entries (k_assemble_simple)
  DBuf<signed char> keep_e;
  // one system over several ranks: collectives are delegated to the caller
  int shard_rank = 0, shard_count = 1;
  hqpkkt_exchange_fn xchg_fn = nullptr;
  hqpkkt_exchange_stream_fn xchg_sfn = nullptr;  // stream-ordered form (RCCL): nothing is drained
  void *xchg_ctx = nullptr;
  DBuf<long long> bptr, panel_off, upd_off, x_off, cb_off, ent_dst, linv_off, pinv_off;
  DBuf<TermDev> terms;
  DBuf<signed char> esign;
  CsrBuf Qf, A, AT, C, CT;
  // numeric state
  DBuf<double> vals, wt, sc, ent_val, panel, upd, xar, dinv, rhs, xsol, cb, ytmp, vtmp, linv;
  DBuf<int> ptype, lperm, flags;  // flags: [0] status, [1] n_2x2, [2] n_perturbed
  // [0] kmax, [1] residual max: inside the flags buffer (ints 120..123) so that status and
  // maxima come back in ONE copy; hpin: pinned host memory those copies land in
  struct {
    unsigned long long *p = nullptr;
  } bits;
  double *hpin = nullptr;  // 128 doubles: 0..63 status words (as ints), 64.. the IP loop's scalars
  // Read-backs without a copy and without hipStreamSynchronize (round 6): hpin is mapped, coherent host memory; a
  // one-wavefront kernel at the point of the stream where the words are final stores them there and a sequence number
  // behind them (k_post_words, kernels.hip.h), the host spins on that number (post_wait).  Measured (tools/post_probe.hip):
  // 6 us per read-back behind a queue of small kernels against 16 for hipMemcpyAsync + hipStreamSynchronize - the
  // device-resident interior-point loops read back three times per iteration.
  double *hpin_dev = nullptr;  // the device's address of hpin
  unsigned post_seq = 0;       // the number the last posting kernel in the stream will store (hpin word HPIN_SEQ)
  // host vectors of a small system: packed into / out of pinned memory by the CPU, ONE
  // transfer each way instead of six + four staged copies from pageable memory
  double *hvals = nullptr;   // pinned host staging of Qx | Ax | Cx (hqpkkt_values_staging), nq + na + nc doubles
  size_t hvals_elems = 0;    // ... as allocated: a new analysis with another pattern allocates again
  double *hstage = nullptr;
  size_t hstage_in = 0, hstage_out = 0;  // doubles; 0 = system too large, copy vector by vector
  const double *out_pending = nullptr;   // results wait in hstage + hstage_in for unstage()
  bool out_by_kernel = false;            // ... written there by a kernel in front of the posting kernel (no stream synchronisation needed)
  bool host_graph_call = false;          // inside a solve whose first part ran as hqpkkt::ghost_step (no timing events in the stream)
  double *hstage_dev = nullptr;          // the device's address of hstage (pinned, coherent: kernels copy in and out of it)
  // vectors: staging for host pointers + refinement work vectors
  DBuf<double> vin;   // z w r1 r2 r3 r4
  DBuf<double> vout;  // dx dy dz dw
  DBuf<double> vres;  // residual vectors _r1.._r4
  DBuf<double> vcor;  // corrections _dx.._dw
  DBuf<double> tz;    // REDUCED temporary (m)
  DBuf<double> ipv;   // interior-point driver: x y z w | r1..r4 | dxa..dwa | dx..dw | c b d | partials | scalars
  size_t lds_diag = 0, lds_panel = 0, lds_bwdb = 0;
  // pivot blocks of the general fronts: k_factor_blk (round 4) unless HQPKKT_OLD_FD asks for k_factor_diag;
  // per schedule and tree level the largest pivot count among the general fronts of the level
  bool old_fd = false;
  std::vector<int> level_maxp[2], level_maxb[2];
  // the device-resident interior-point loops: cancelled multiplier pivots are replaced (kernels.hip.h, TINY_REPLACE_WORD)
  // only in the SECOND attempt of a run whose first attempt - without the replacement, i.e. with the factors the
  // reference's own loop gets from this plugin through the shim - ended "degenerate" or singular
  bool tiny_replace_in_loop = false;
  int xcd_ps = -1, xcd_su = -1;  // chunk of the border kernels' work lists per XCD (kernels.hip.h, xcd_order); -1: from the level's fronts; 0: list order (HQPKKT_XCD_PS / _SU)
  int su1_max = 768;   // levels of at most this many 64 x 64 update tiles run k_schur_update with 32 x 16 per wave (HQPKKT_SU1_MAX)
  // the top levels of the tree solved in one launch (solve_top.hip.h): fronts of the levels >= top_lt, root first
  int top_n = 0, top_lt = 1 << 30, top_ns = 3;  // top_ns: 3 = k_solve_top<3, 11>, 4 = <4, 10>
  size_t top_lds = 0;
  DBuf<int> top_nodes, top_idx, top_bpos, top_up;  // top_up: the fronts leaves first (top_split)
  bool top_split = false;  // more fronts than one launch may hold at once: the two sweeps as launches of their own
  unsigned long long *top_stamps = nullptr;  // (hqpkkt_debug_solve_top_stamps)
  // trees of small fronts only (the double-integrator structure): each sweep of the solve is ONE launch over all levels
  // (k_solve_fwd_small<true> / k_solve_bwd_small<true>); tree_x: the exchange arrays (2 x cb_elems, then 2 x dim)
  bool small_tree = false, tree_factor = false;  // tree_factor: ... and the factorisation too (k_factor_diag_small<true, true>)
  DBuf<double> tree_x, tree_u;   // tree_u: the exchange copies of the update arena (2 x upd_elems)
  DBuf<int> tree_words, tree_down;  // [0] solves so far, [1] factorisations so far; the fronts root first
  DBuf<double> top_x;  // the exchange arrays of the launch: 2 x top_n x ST_CS contributions, then 2 x top_n x ST_XS solution
  // captured kernel sequences (factor; step on the caller's vectors; step on the
  // refinement's residual vectors): replayed with hipGraphLaunch
  struct GraphSlot {
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    unsigned n_posts = 0;  // posted read-backs inside (k_post_words counts on the device; the host counts along at every replay)
    void drop() {
      if (ge) (void)hipGraphExecDestroy(ge);
      if (g) (void)hipGraphDestroy(g);
      ge = nullptr, g = nullptr, n_posts = 0;
    }
  } gfactor[2], gstep[2][3];  // [phase], [caller's / refinement's vectors][phase]
  // A caller with HOST vectors (the reference's solvers through the shim): the packed vectors are read out of the pinned
  // staging buffer by a kernel, the results written into it by a kernel, and the status words posted - a whole call is
  // one graph on the compute queue (no copy engine between the launches: 9 - 13 us at each change of engine,
  // profiles/r06_shim_timeline.txt) and ends with the posted words, not a stream synchronisation
  GraphSlot ghost_factor, ghost_step;
  // The device-resident interior-point loops hand over the same device vectors in every iteration: their sequences are
  // captured ON those vectors (no copies into and out of the handle's staging buffers), one graph per set of pointers.
  struct DirectGraph {
    const void *key[10];
    GraphSlot g;
  };
  std::vector<DirectGraph> gdirect_step, gdirect_factor;
  // ... and whole SEGMENTS of an iteration of the device-resident loops - everything between two read-backs: the
  // factorisation, a solve, its residual and the posting kernel - as one graph (ip_segment)
  std::vector<DirectGraph> gdirect_seg;
  // ... and a caller's own factor / solve call on its device vectors with its residual and the posted words
  std::vector<DirectGraph> gdirect_call;
  GraphSlot &direct_slot(std::vector<DirectGraph> &cache, const void *const (&key)[10]) {
    for (auto &d : cache)
      if (std::memcmp(d.key, key, sizeof(key)) == 0) return d.g;
    if (cache.size() >= 6) {  // (Mehrotra's loop has two sets + the refinement's, Franke's one + the refinement's)
      cache.front().g.drop();
      cache.erase(cache.begin());
    }
    cache.emplace_back();
    std::memcpy(cache.back().key, key, sizeof(key));
    return cache.back().g;
  }
  bool use_graphs = true, capturing = false;
  unsigned cap_posts = 0;  // posted read-backs of the capture in progress
  DBuf<unsigned> post_seq_dev;  // the sequence number of the posted read-backs, counted by k_post_words
  // inside hqpkkt_mehrotra: factor() returns without waiting for its status (read with the
  // residual of the solve that follows), solve() leaves its result in the stream
  bool lazy = false, factor_unchecked = false;
  // hqpkkt_factor / hqpkkt_solve of a caller with DEVICE vectors: the second call in a row with the same pointers works
  // on the caller's vectors themselves (no staging copies; the sequences are captured on them, DirectGraph) - set for
  // the duration of that call.  last_f / last_s: the pointers of the previous call of either kind
  bool direct_now = false;
  const void *last_f[2] = {nullptr, nullptr}, *last_s[10] = {};
  // hqpkkt_franke: the first residual of a solve is not waited for - it comes back with the scalars of the iteration
  // (one read-back per iteration); residual_pending: such a residual is in the stream, collect_residual() reads it
  bool defer_residual = false, residual_pending = false;
  int res_read = 122;  // the word of the flags buffer the residual kernels leave their maximum in (cleared by k_post_words)
  bool no_polled = false;      // a polled launch gave up once: per-level launches for the rest of the handle's life (poll_fallback)
  bool soft_singular = false;  // the factorisation perturbed an exactly zero pivot (counters[3])
  bool soft_tiny = false;      // ... or met a pivot below 1e-13 max|K| on a multiplier-type row (counters[4])
  double refine_target = 0.0;  // > 0: the refinement of hqpkkt_solve aims below mat_eps (set by hqpkkt_franke)
  // hqpkkt_mehrotra left x, y and the hot-start candidates of z, w in ipv (same dimensions)
  bool ip_hot_valid = false;
  bool fr_hot_valid = false;  // hqpkkt_franke left x, y, z, w in ipv (same dimensions)
  double fr_rhomin = 0.0;     // ... and its qp_rhomin, which hot_start keeps (hqp/Hqp_IpsFranke.C:222-266)
  // the caller's pattern (hqpkkt_analyze), kept for the one repetition of the symbolic phase
  // that zd_policy -1 may ask for when the first values arrive; zd_used: policy of h->an
  std::vector<int> pQp, pQi, pAp, pAi, pCp, pCi;
  int zd_used = 2;
  bool zd_decided = true;
  // zd_policy -1 on a QP with weak Hessian diagonals: the values on the host, so that a solve
  // whose refinement fails can switch the handle to policy 0 (symbolic phase, upload, values,
  // factorisation again) and repeat itself
  bool zd_weak = false;
  std::vector<double> hQ, hA, hC;
  bool short_rows = false;  // CSR rows of a handful of entries: 4 lanes per row in the SpMV kernels
  void drop_graphs() {
    for (auto &g : gfactor) g.drop();
    ghost_factor.drop(), ghost_step.drop();
    for (auto &gs : gstep)
      for (auto &g : gs) g.drop();
    for (auto &d : gdirect_step) d.g.drop();
    for (auto &d : gdirect_factor) d.g.drop();
    for (auto &d : gdirect_seg) d.g.drop();
    for (auto &d : gdirect_call) d.g.drop();
    gdirect_step.clear(), gdirect_factor.clear(), gdirect_seg.clear(), gdirect_call.clear();
  }

  DevTree tree() const {
    return DevTree{piv_start.p, npiv.p,     nbor.p,  parent.p, bptr.p,      bidx.p,     rel.p,
                   panel_off.p, upd_off.p, x_off.p, cb_off.p, child_ptr.p, child_idx.p, pinv.p, pinv_off.p};
  }
  void release_device(bool keep_ip = false) {  // keep_ip: hqpkkt_mehrotra's vectors and the pinned words stay
    DBuf<int> *ib[] = {&piv_start, &npiv, &nbor, &parent, &bidx, &rel, &child_ptr, &child_idx,
                       &ent_a, &ent_b, &term_ptr, &diag_ent, &q2e, &pinv, &ptype, &lperm, &flags,
                       &top_nodes, &top_idx, &top_bpos, &top_up, &tree_words, &tree_down};
    for (auto b : ib) b->release();
    ds[0].release(), ds[1].release(), keep_e.release(), simple_src.release(), simple_wi.release();
    DBuf<long long> *lb[] = {&bptr, &panel_off, &upd_off, &x_off, &cb_off, &ent_dst, &linv_off, &pinv_off,
                             &zero_panel};
    for (auto b : lb) b->release();
    DBuf<double> *db[] = {&vals, &wt, &sc, &ent_val, &panel, &upd, &xar, &dinv, &rhs, &xsol,
                          &cb, &vin, &vout, &vres, &vcor, &tz, &ytmp, &vtmp, &linv, &top_x, &tree_x, &tree_u};
    for (auto b : db) b->release();
    if (!keep_ip) ipv.release();
    terms.release(), esign.release(), bits.p = nullptr;
    if (hpin && !keep_ip) (void)hipHostFree(hpin), hpin = nullptr, hpin_dev = nullptr, post_seq_dev.release(), post_seq = 0;
    if (hstage) (void)hipHostFree(hstage), hstage = nullptr, hstage_dev = nullptr;
    // (keep_ip = the re-analysis inside hqpkkt_solve, switch_to_policy0: the pattern and with it the sizes of the
    // pinned value staging stay, and a host may hold the pointers of hqpkkt_values_staging)
    if (hvals && !keep_ip) (void)hipHostFree(hvals), hvals = nullptr, hvals_elems = 0;
    hstage_in = hstage_out = 0;
    Qf.release(), A.release(), AT.release(), C.release(), CT.release();
    if (sd) staged_release(sd, false);
    drop_graphs();
    uploaded = have_values = factored = false;
  }
};

static inline int nblk(long long work, int bs = 256) { return (int)((work + bs - 1) / bs); }
// grid of k_copy_vectors: enough workgroups for the longest vector, at most 1024
static inline int copy_blocks(const CopyList &L) {
  int mx = 1;
  for (int v = 0; v < 6; v++) mx = std::max(mx, L.len[v]);
  return std::min(1024, std::max(1, nblk(mx)));
}

static int ensure_device(hqpkkt_t *h) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= h->opts.device) {
    std::snprintf(g_last_hip_error, sizeof(g_last_hip_error),
                  "no HIP device %d (gfx950 required; there is no CPU fallback)", h->opts.device);
    return HQPKKT_E_DEVICE;
  }
  HIPCHK(hipSetDevice(h->opts.device));
  if (!h->own_stream) {
    HIPCHK(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreate(&h->ev0));
    HIPCHK(hipEventCreate(&h->ev1));
    HIPCHK(hipEventCreate(&h->evs0));
    HIPCHK(hipEventCreate(&h->evs1));
    HIPCHK(hipEventCreate(&h->evt0));
    HIPCHK(hipEventCreate(&h->evt1));
  }
  if (!h->stream) h->stream = h->own_stream;
  return 0;
}

// the exchange arrays of k_solve_top in their idle state: every word the sentinel, counters zero
static int reset_solve_top(hqpkkt_t *h) {
  HIPCHK(hipStreamSynchronize(h->stream));
  if (h->tree_words.p) HIPCHK(hipMemset(h->tree_words.p, 0, sizeof(int) * 2));
  auto fill = [&](DBuf<double> &buf, size_t count) -> int {
    std::vector<double> f(count);
    for (auto &x : f) std::memcpy(&x, &XW_SENTINEL, sizeof(double));
    HIPCHK(hipMemcpy(buf.p, f.data(), sizeof(double) * count, hipMemcpyHostToDevice));
    return 0;
  };
  int e;
  if (h->top_n > 0) {
    if ((e = fill(h->top_x, 2 * (size_t)h->top_n * (ST_CS + ST_XS)))) return e;
  }
  if (h->small_tree) {
    if ((e = fill(h->tree_x, 2 * (size_t)(h->an.cb_elems + h->an.dim)))) return e;
    if (h->tree_factor && (e = fill(h->tree_u, 2 * (size_t)std::max<long long>(h->an.upd_elems, 1)))) return e;

  }
  return 0;
}

// A polled launch gave up (flags[XW_GAVE_UP] in the words `hs` read back from the device): the exchange arrays go back
// to their idle state and the handle switches - for good - to the per-level launches, which wait for nothing inside a
// launch (the polled launches rest on workgroups being dispatched in index order, which HIP does not promise:
// DESIGN.md section 2a).  Returns true when that happened: the caller's operation has to be run again.
static const int HQPKKT_E_POLL = -7001;  // internal: never leaves the library (the entry points run the call again)
static bool poll_fallback(hqpkkt_t *h, const int *hs) {
  if (!hs[XW_GAVE_UP]) return false;
  (void)reset_solve_top(h);
  (void)hipMemsetAsync(h->flags.p + XW_GAVE_UP, 0, sizeof(int), h->stream);
  (void)hipStreamSynchronize(h->stream);
  h->top_n = 0, h->small_tree = false, h->tree_factor = false, h->no_polled = true;  // (no_polled: a later upload stays there)
  h->drop_graphs();
  h->st.n_poll_fallbacks++;
  if (getenv("HQPKKT_TRACE_SOLVE")) fprintf(stderr, "a polled launch gave up: per-level launches from now on\n");
  return true;
}

// the mapped, coherent host words of the read-backs (hqpkkt::hpin; both engines)
static int alloc_hpin(hqpkkt_t *h) {
  if (h->hpin) return 0;
  HIPCHK(hipHostMalloc((void **)&h->hpin, sizeof(double) * HPIN_DOUBLES, hipHostMallocMapped | hipHostMallocCoherent));
  std::memset(h->hpin, 0, sizeof(double) * HPIN_DOUBLES);
  HIPCHK(hipHostGetDevicePointer((void **)&h->hpin_dev, h->hpin, 0));
  h->post_seq = 0;
  int e = h->post_seq_dev.alloc(1);
  if (e) return e;
  HIPCHK(hipMemset(h->post_seq_dev.p, 0, sizeof(unsigned)));
  return 0;
}
static int upload(hqpkkt_t *h) {
  int e = ensure_device(h);
  if (e) return e;
  Analysis &an = h->an;
#define UP(buf, vec) \
  if ((e = h->buf.upload(an.vec))) return e
  UP(piv_start, piv_start);
  UP(npiv, npiv);
  UP(nbor, nbor);
  UP(parent, parent);
  UP(bidx, bidx);
  UP(rel, rel);
  UP(child_ptr, child_ptr);
  UP(child_idx, child_idx);
  for (int w = 0; w < 2; w++) {
    UP(ds[w].level_nodes, sched[w].level_nodes);
    UP(ds[w].upd_tiles, sched[w].upd_tiles);
    UP(ds[w].slabs, sched[w].slabs);
    UP(ds[w].gslabs, sched[w].gslabs);
    UP(ds[w].cblks, sched[w].cblks);
  }
  UP(zero_panel, zero_panel);
  UP(keep_e, keep_e);
  UP(linv_off, linv_off);
  UP(pinv, pinv);
  UP(pinv_off, pinv_off);
  UP(ent_a, ent_a);
  UP(ent_b, ent_b);
  UP(term_ptr, term_ptr);
  UP(diag_ent, diag_ent);
  UP(q2e, q2e);
  UP(bptr, bptr);
  UP(panel_off, panel_off);
  UP(upd_off, upd_off);
  UP(x_off, x_off);
  UP(cb_off, cb_off);
  UP(ent_dst, ent_dst);
#undef UP
  {
    std::vector<TermDev> t(an.terms.size());
    for (size_t k = 0; k < t.size(); k++)
      t[k] = TermDev{an.terms[k].s1, an.terms[k].s2, an.terms[k].wi, an.terms[k].sgn};
    if ((e = h->terms.upload(t))) return e;
    // all entries single terms sgn * vals[s1] * wt[wi] with s2 = the constant 1 (FULL plugin)?
    const int one = an.nq + an.na + an.nc;
    bool simple = an.mode == 0 && an.terms.size() == an.ent_a.size();
    for (size_t k = 0; simple && k < t.size(); k++)
      simple = t[k].s2 == one && (t[k].sgn == 1.0 || t[k].sgn == -1.0);
    if (simple) {
      std::vector<int> ss(t.size()), ww(t.size());
      for (size_t k = 0; k < t.size(); k++) ss[k] = t[k].s1 | (t[k].sgn < 0 ? (int)0x80000000 : 0), ww[k] = t[k].wi;
      if ((e = h->simple_src.upload(ss)) || (e = h->simple_wi.upload(ww))) return e;
    }
    // sign a perturbed pivot takes: x rows belong to the -Q block, y / slack rows
    // to the zero / +W/Z blocks
    std::vector<signed char> sg(an.dim);
    for (int q = 0; q < an.dim; q++) sg[an.q2e[q]] = q < an.n ? -1 : 1;
    // +-2: no diagonal of its own (see zero_pivot_slot in kernels.hip.h)
    std::vector<char> in_c(an.n, 0);  // REDUCED: C' (Z/W) C gives x_i a diagonal as well
    if (an.mode != 0)
      for (int c : h->pCi) in_c[c] = 1;
    for (int q = 0; q < an.n; q++) {
      bool diag = in_c[q] != 0;
      for (int k = h->pQp[q]; k < h->pQp[q + 1]; k++) diag = diag || h->pQi[k] == q;
      if (!diag) sg[an.q2e[q]] = -2;
    }
    for (int q = an.n; q < an.n + an.me; q++) sg[an.q2e[q]] = 2;
    if ((e = h->esign.upload(sg))) return e;
  }
  if ((e = h->Qf.upload(an.Qfull)) || (e = h->A.upload(an.A)) || (e = h->AT.upload(an.AT)) ||
      (e = h->C.upload(an.C)) || (e = h->CT.upload(an.CT)))
    return e;
  const int n = an.n, me = an.me, m = an.m, dim = an.dim;
  const size_t nv = (size_t)an.nq + an.na + an.nc + 1;
  if ((e = h->vals.alloc(nv)) || (e = h->wt.alloc(m + 1)) || (e = h->sc.alloc(dim)) ||
      (e = h->ent_val.alloc(an.ent_a.size())) || (e = h->panel.alloc(an.panel_elems)) ||
      (e = h->upd.alloc(an.upd_elems)) || (e = h->xar.alloc(an.x_elems)) ||
      (e = h->dinv.alloc(2 * (size_t)dim)) || (e = h->rhs.alloc(dim)) ||
      (e = h->xsol.alloc(dim)) || (e = h->cb.alloc(an.cb_elems)) || (e = h->ytmp.alloc(std::max(dim, 8))) ||
      (e = h->vtmp.alloc(dim)) || (e = h->linv.alloc(an.linv_elems)) || (e = h->ptype.alloc(dim)) ||
      (e = h->lperm.alloc(dim)) || (e = h->flags.alloc(128)) ||
      (e = h->vin.alloc(2 * (size_t)m + n + me + 2 * (size_t)m)) ||
      (e = h->vout.alloc((size_t)n + me + 2 * (size_t)m)) ||
      (e = h->vres.alloc((size_t)n + me + 2 * (size_t)m)) ||
      (e = h->vcor.alloc((size_t)n + me + 2 * (size_t)m)) || (e = h->tz.alloc(m)))
    return e;
  h->bits.p = (unsigned long long *)(h->flags.p + 120);
  HIPCHK(hipMemset(h->flags.p, 0, sizeof(int) * 128));
  h->res_read = 122;
  if ((e = alloc_hpin(h))) return e;
  if (h->hstage) (void)hipHostFree(h->hstage), h->hstage = nullptr, h->hstage_dev = nullptr;
  h->hstage_in = h->hstage_out = 0;
  {
    const size_t nin = 4 * (size_t)m + n + me, nout = (size_t)n + me + 2 * (size_t)m;
    if ((nin + nout) * sizeof(double) <= (size_t)512 * 1024 && nin + nout > 0) {
      HIPCHK(hipHostMalloc((void **)&h->hstage, sizeof(double) * (nin + nout), hipHostMallocMapped | hipHostMallocCoherent));
      h->hstage_in = nin, h->hstage_out = nout;
      h->hstage_dev = nullptr;
      if (hipHostGetDevicePointer((void **)&h->hstage_dev, h->hstage, 0) != hipSuccess) h->hstage_dev = nullptr, (void)hipGetLastError();
    }
  }
  {
    std::vector<double> ones(dim, 1.0);
    HIPCHK(hipMemcpy(h->sc.p, ones.data(), sizeof(double) * dim, hipMemcpyHostToDevice));
    const double one = 1.0;
    HIPCHK(hipMemcpy(h->vals.p + (nv - 1), &one, sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->wt.p + m, &one, sizeof(double), hipMemcpyHostToDevice));
  }
  // dynamic LDS budgets
  const size_t mp = an.max_npiv, ldm = mp | 1;
  h->lds_diag = (std::max<size_t>(ldm * mp, 2 * FD_PLD * FD_PANEL) + 5 * 128 + 2 * mp) * sizeof(double) +
                2 * mp * sizeof(int) + 16;
  h->lds_panel = (PS_LD * mp + 1 + 2 * mp) * sizeof(double) + mp * sizeof(int);
  h->lds_bwdb = ((size_t)an.max_nbor + 2) * sizeof(double);
  h->old_fd = false;  // (k_factor_diag of rounds 1-3 stays reachable through hqpkkt_debug_factor_block: tests/test_gpu_block.py compares the two)
  if (getenv("HQPKKT_SU1_MAX")) h->su1_max = atoi(getenv("HQPKKT_SU1_MAX"));
  if (getenv("HQPKKT_XCD_PS")) h->xcd_ps = atoi(getenv("HQPKKT_XCD_PS"));
  if (getenv("HQPKKT_XCD_SU")) h->xcd_su = atoi(getenv("HQPKKT_XCD_SU"));
  if (!h->old_fd) h->lds_diag = 0;
  const size_t lds_blk = fb_lds_bytes((int)mp);
  if (h->lds_diag > 160 * 1024 || h->lds_bwdb > 160 * 1024 || lds_blk > 160 * 1024) return HQPKKT_E_MEM;
  for (int w = 0; w < 2; w++) {
    const Analysis::Sched &S = an.sched[w];
    h->level_maxp[w].assign(an.nlevels, 0), h->level_maxb[w].assign(an.nlevels, 0);
    for (int l = 0; l < an.nlevels && S.nnodes; l++)
      for (int q = S.level_ptr[l] + S.level_fsmall[l] + S.level_small[l]; q < S.level_ptr[l + 1]; q++) {
        h->level_maxp[w][l] = std::max(h->level_maxp[w][l], an.npiv[S.level_nodes[q]]);
        h->level_maxb[w][l] = std::max(h->level_maxb[w][l], an.nbor[S.level_nodes[q]]);
      }
  }
  {  // the counters of the polled exchanges: [0] solves, [1] factorisations so far (k_rhs_*, the assembly kernels count)
    std::vector<int> two(2, 0);
    if ((e = h->tree_words.upload(two))) return e;
  }
  {  // tries before a poll gives up (HQPKKT_POLL_LIMIT: a test hook that forces the fall-back of poll_fallback)
    const char *pl = getenv("HQPKKT_POLL_LIMIT");
    const int lim = pl ? atoi(pl) : 1 << 20;
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(xw_poll_limit), &lim, sizeof(int)));
    const double spp = 1e-6;  // (kernels.hip.h, soft_pivot_pert)
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(soft_pivot_pert), &spp, sizeof(double)));
  }
  // a tree of small fronts only: whole-tree sweeps
  h->small_tree = false, h->tree_factor = false;
  if (!getenv("HQPKKT_NO_TREE_SWEEPS") && !h->no_polled && an.shard_count == 1 && an.sched[0].nnodes > 1 && an.sched[1].nnodes == 0) {
    const Analysis::Sched &S = an.sched[0];
    bool all = true;
    for (int l = 0; l < an.nlevels && all; l++) all = S.level_fsmall[l] == S.level_ptr[l + 1] - S.level_ptr[l];
    if (all) {
      std::vector<int> down, one(2, 0);
      for (int l = an.nlevels - 1; l >= 0; l--)
        for (int q = S.level_ptr[l]; q < S.level_ptr[l + 1]; q++) down.push_back(S.level_nodes[q]);
      if ((e = h->tree_down.upload(down)) || (e = h->tree_x.alloc(2 * (size_t)(an.cb_elems + an.dim)))) return e;
      h->small_tree = true;
      h->tree_factor = !an.upd_pingpong;
      if (h->tree_factor && (e = h->tree_u.alloc(2 * (size_t)std::max<long long>(an.upd_elems, 1)))) return e;
      if ((e = reset_solve_top(h))) return e;
    }
  }
  // the fused top of the solve sweeps: the highest levels whose fronts all fit one instance of k_solve_top, at most
  // ST_MAXFRONTS fronts (single rank: with a sharded tree the two sweeps of a schedule are not adjacent)
  h->top_n = 0, h->top_lt = 1 << 30, h->top_lds = 0;
  if (!getenv("HQPKKT_NO_SOLVE_TOP") && !h->no_polled && an.shard_count == 1 && an.sched[0].nnodes > 0) {
    const Analysis::Sched &S = an.sched[0];
    int lt = an.nlevels, cnt = 0, maxp = 0;
    bool ok3 = true, ok4 = true;  // the instances <3, 11> and <4, 10>
    for (int l = an.nlevels - 1; l >= 0; l--) {
      const int nn = S.level_ptr[l + 1] - S.level_ptr[l];
      bool f3 = ok3, f4 = ok4;
      int mp2 = maxp;
      for (int q = S.level_ptr[l]; q < S.level_ptr[l + 1]; q++) {
        const int v = S.level_nodes[q];
        f3 = f3 && st_top_fits(an.npiv[v], an.nbor[v], 3, 11), f4 = f4 && st_top_fits(an.npiv[v], an.nbor[v], 4, 10);
        mp2 = std::max(mp2, an.npiv[v]);
      }
      // (levels of small fronts stay with their one-wavefront kernels: a step of k_solve_top costs 16 wavefronts'
      // worth of barriers and reductions whatever the size of the front - measured slower on the DID tree)
      if (cnt + nn > ST_MAXSPLIT || !(f3 || f4) || S.level_fsmall[l] > 0) break;
      cnt += nn, lt = l, maxp = mp2, ok3 = f3, ok4 = f4;
    }
    if (an.nlevels - lt >= 2 && cnt >= 2) {
      std::vector<int> nodes, idx(an.nnodes, -1), owner(an.dim, -1);
      for (int l = an.nlevels - 1; l >= lt; l--)
        for (int q = S.level_ptr[l]; q < S.level_ptr[l + 1]; q++) idx[S.level_nodes[q]] = (int)nodes.size(), nodes.push_back(S.level_nodes[q]);
      for (size_t t = 0; t < nodes.size(); t++)
        for (int k = 0; k < an.npiv[nodes[t]]; k++) owner[an.piv_start[nodes[t]] + k] = (int)t;
      std::vector<int> bpos(nodes.size() * ST_CS, 0);
      for (size_t t = 0; t < nodes.size(); t++)
        for (int i = 0; i < an.nbor[nodes[t]]; i++) {
          const int ei = an.bidx[an.bptr[nodes[t]] + i], o = owner[ei];
          if (o < 0) return HQPKKT_E_INTERN;  // (a border row of a fused front belongs to a fused ancestor)
          bpos[t * ST_CS + i] = o * ST_XS + (ei - an.piv_start[nodes[o]]);
        }
      std::vector<int> up;  // level by level, leaves first; inside a level the largest fronts first, as in `nodes`
      for (int l = lt; l < an.nlevels; l++)
        for (int q = S.level_ptr[l]; q < S.level_ptr[l + 1]; q++) up.push_back(S.level_nodes[q]);
      // One launch for both sweeps needs ALL its fronts resident at once (the forward sweep of a front waits for
      // fronts behind it in the launch): safe only while nothing else competes for the CUs.  Several systems in flight
      // on one GPU (bench.py's concurrent systems, scenario trees) could starve each other, so the form in use is the
      // split one - a front waits only for fronts before it - and the fused launch is an option (HQPKKT_SOLVE_TOP_FUSED,
      // 17 us less per solve: M and L21 are read once).
      h->top_split = (int)nodes.size() > ST_MAXFRONTS || getenv("HQPKKT_SOLVE_TOP_FUSED") == nullptr;
      if ((e = h->top_nodes.upload(nodes)) || (e = h->top_idx.upload(idx)) || (e = h->top_bpos.upload(bpos)) || (e = h->top_up.upload(up)) ||
          (e = h->top_x.alloc(2 * nodes.size() * (size_t)(ST_CS + ST_XS))))
        return e;
      h->top_n = (int)nodes.size(), h->top_lt = lt, h->top_ns = ok3 ? 3 : 4, h->top_lds = st_top_lds_bytes(maxp, h->top_ns);
      if ((e = reset_solve_top(h))) return e;
    }
  }
  {
    // the attribute is state of the PROCESS, not of the handle: a second handle with smaller fronts must
    // not lower the limit under one that still launches with more (several plugins in one host, the
    // bench's concurrent systems): keep the largest value ever asked for, under a mutex
    // ... and hipFuncSetAttribute acts on the CURRENT device: the largest values are kept per device
    static std::mutex attr_mutex;
    struct PerDev { size_t diag = 0, panel = 0, bwdb = 0, blk = 0, top = 0; };
    static PerDev per_dev[64];
    if (h->opts.device < 0 || h->opts.device >= 64) return HQPKKT_E_RANGE;
    std::lock_guard<std::mutex> lk(attr_mutex);
    size_t &a_diag = per_dev[h->opts.device].diag, &a_panel = per_dev[h->opts.device].panel, &a_bwdb = per_dev[h->opts.device].bwdb,
           &a_blk = per_dev[h->opts.device].blk, &a_top = per_dev[h->opts.device].top;
    if (h->top_lds > a_top) {
      const int l3 = (int)std::min(h->top_lds, st_top_lds_bytes(176, 3)), l4 = (int)std::min(h->top_lds, st_top_lds_bytes(160, 4));
      HIPCHK(hipFuncSetAttribute((const void *)k_solve_top<3, 11, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, l3));
      HIPCHK(hipFuncSetAttribute((const void *)k_solve_top<3, 11, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, l3));
      HIPCHK(hipFuncSetAttribute((const void *)k_solve_top<3, 11, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, l3));
      HIPCHK(hipFuncSetAttribute((const void *)k_solve_top<4, 10, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, l4));
      HIPCHK(hipFuncSetAttribute((const void *)k_solve_top<4, 10, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, l4));
      HIPCHK(hipFuncSetAttribute((const void *)k_solve_top<4, 10, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, l4));
      a_top = h->top_lds;
    }
    if (lds_blk > a_blk) {
      HIPCHK(hipFuncSetAttribute((const void *)k_factor_blk<8, 6, 144, 2, FB_OWNSIMD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::min(lds_blk, fb_lds_bytes(128))));
      HIPCHK(hipFuncSetAttribute((const void *)k_factor_blk<12, 8, 208, 3, FB_OWNSIMD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_blk));
      HIPCHK(hipFuncSetAttribute((const void *)k_factor_blk<12, 6, 208, 3, FB_OWNSIMD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::min(lds_blk, fb_lds_bytes(176))));
      HIPCHK(hipFuncSetAttribute((const void *)k_factor_blk<12, FB_NS160, 208, 3, FB_OWNSIMD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::min(lds_blk, fb_lds_bytes(160))));
      a_blk = lds_blk;
    }
    if (h->lds_diag > a_diag) {
      HIPCHK(hipFuncSetAttribute((const void *)k_factor_diag, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_diag));
      a_diag = h->lds_diag;
    }
    if (h->lds_panel > a_panel) {
      HIPCHK(hipFuncSetAttribute((const void *)k_panel_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_panel));
      a_panel = h->lds_panel;
    }
    if (h->lds_bwdb > a_bwdb) {
      HIPCHK(hipFuncSetAttribute((const void *)k_solve_bwd_b, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bwdb));
      a_bwdb = h->lds_bwdb;
    }
  }
  {
    const double rows = 2.0 * n + me + m;  // Q, A', C' per x row; A, C rows
    const double nnz = (double)an.Qfull.col.size() + 2.0 * an.A.col.size() + 2.0 * an.C.col.size();
    h->short_rows = rows > 0 && nnz / rows < 8.0;
  }
  h->st.bytes_panels = (long long)sizeof(double) * (an.panel_elems + an.x_elems);
  h->st.bytes_updates = (long long)sizeof(double) * an.upd_elems;
  h->uploaded = true;
  return 0;
}

// pointers of the six input vectors / four outputs for the current call
struct Vecs {
  const double *z, *w, *r1, *r2, *r3, *r4;
  double *dx, *dy, *dz, *dw;
};

// (HQPKKT_NO_HOST_KERNEL_COPIES=1: the copy engine as before, for same-box comparisons)
static bool host_kernel_copies(const hqpkkt_t *) {
  return getenv("HQPKKT_NO_HOST_KERNEL_COPIES") == nullptr;
}
// A call with host vectors as one graph (hqpkkt::ghost_factor / ghost_step): the tree engine on one GPU, vectors that
// fit the pinned staging buffer.  (HQPKKT_NO_HOST_GRAPHS=1: launch by launch.)
static bool host_graphs_ok(const hqpkkt_t *h) {
  return getenv("HQPKKT_NO_HOST_GRAPHS") == nullptr && host_kernel_copies(h) && !h->lazy && h->opts.loc != HQPKKT_LOC_DEVICE && h->hstage_in && h->hstage_dev && h->use_graphs &&
         !h->prof.on && h->opts.mode != HQPKKT_MODE_STAGED && h->an.shard_count <= 1;
}
// the caller's vectors packed into the pinned buffer by the CPU (stage_in's layout); returns the doubles in use
static size_t stage_pack(hqpkkt_t *h, const double *z, const double *w, const double *r1, const double *r2, const double *r3, const double *r4) {
  const int n = h->an.n, me = h->an.me, m = h->an.m;
  double *q = h->hstage;
  const double *src[6] = {z, w, r1, r2, r3, r4};
  const int len[6] = {m, m, n, me, m, m};
  size_t used = 0, off = 0;
  for (int k = 0; k < 6; k++) {
    if (src[k] && len[k] > 0) std::memcpy(q + off, src[k], sizeof(double) * len[k]), used = off + len[k];
    off += len[k];
  }
  return used;
}
static int stage_in(hqpkkt_t *h, const double *z, const double *w, const double *r1,
                    const double *r2, const double *r3, const double *r4, Vecs &v) {
  const int n = h->an.n, me = h->an.me, m = h->an.m;
  double *b = h->vin.p;
  double *dz_ = b, *dw_ = b + m, *d1 = b + 2 * (size_t)m, *d2 = d1 + n, *d3 = d2 + me, *d4 = d3 + m;
  if (h->opts.loc == HQPKKT_LOC_DEVICE) {
    CopyList L{{z, w, r1, r2, r3, r4}, {dz_, dw_, d1, d2, d3, d4}, {m, m, n, me, m, m}};
    k_copy_vectors<<<copy_blocks(L), 256, 0, h->stream>>>(L, 6);
  } else if (h->hstage_in) {
    // packed by the CPU; the prefix up to the last vector the caller passes
    double *q = h->hstage;
    const double *src[6] = {z, w, r1, r2, r3, r4};
    const int len[6] = {m, m, n, me, m, m};
    size_t used = 0, off = 0;
    for (int k = 0; k < 6; k++) {
      if (src[k] && len[k] > 0) std::memcpy(q + off, src[k], sizeof(double) * len[k]), used = off + len[k];
      off += len[k];
    }
    if (used && h->hstage_dev && host_kernel_copies(h)) {  // read out of the pinned buffer by a kernel: no copy engine in the chain
      CopyList L{{h->hstage_dev, nullptr, nullptr, nullptr, nullptr, nullptr}, {b, nullptr, nullptr, nullptr, nullptr, nullptr}, {(int)used, 0, 0, 0, 0, 0}};
      k_copy_vectors<<<copy_blocks(L), 256, 0, h->stream>>>(L, 1);
    } else if (used)
      HIPCHK(hipMemcpyAsync(b, q, sizeof(double) * used, hipMemcpyHostToDevice, h->stream));
  } else {
#define H2D(dst, src, k) \
  if ((src) && (k) > 0) HIPCHK(hipMemcpyAsync(dst, src, sizeof(double) * (k), hipMemcpyHostToDevice, h->stream))
    H2D(dz_, z, m);
    H2D(dw_, w, m);
    H2D(d1, r1, n);
    H2D(d2, r2, me);
    H2D(d3, r3, m);
    H2D(d4, r4, m);
#undef H2D
  }
  v.z = dz_, v.w = dw_, v.r1 = d1, v.r2 = d2, v.r3 = d3, v.r4 = d4;
  return 0;
}

static void stage_out_ptrs(hqpkkt_t *h, Vecs &v) {
  const int n = h->an.n, me = h->an.me, m = h->an.m;
  v.dx = h->vout.p, v.dy = v.dx + n, v.dz = v.dy + me, v.dw = v.dz + m;
}

static int stage_out(hqpkkt_t *h, const Vecs &v, double *dx, double *dy, double *dz, double *dw) {
  const int n = h->an.n, me = h->an.me, m = h->an.m;
  if (h->opts.loc == HQPKKT_LOC_DEVICE) {
    CopyList L{{v.dx, v.dy, v.dz, v.dw, nullptr, nullptr}, {dx, dy, dz, dw, nullptr, nullptr}, {n, me, m, m, 0, 0}};
    k_copy_vectors<<<copy_blocks(L), 256, 0, h->stream>>>(L, 4);
    return 0;
  }
  if (h->hstage_out) {  // one transfer into pinned memory; unstage() hands it out after the sync
    if (h->hstage_dev && host_kernel_copies(h)) {  // ... written by a kernel (coherent host memory: there when the next kernel of the stream starts)
      CopyList L{{v.dx, nullptr, nullptr, nullptr, nullptr, nullptr}, {h->hstage_dev + h->hstage_in, nullptr, nullptr, nullptr, nullptr, nullptr}, {(int)h->hstage_out, 0, 0, 0, 0, 0}};
      k_copy_vectors<<<copy_blocks(L), 256, 0, h->stream>>>(L, 1);
      h->out_by_kernel = true;
    } else {
      HIPCHK(hipMemcpyAsync(h->hstage + h->hstage_in, v.dx, sizeof(double) * h->hstage_out, hipMemcpyDeviceToHost,
                            h->stream));
      h->out_by_kernel = false;
    }
    h->out_pending = h->hstage + h->hstage_in;
    return 0;
  }
#define D2H(dst, src, k) \
  if ((dst) && (k) > 0) HIPCHK(hipMemcpyAsync(dst, src, sizeof(double) * (k), hipMemcpyDeviceToHost, h->stream))
  D2H(dx, v.dx, n);
  D2H(dy, v.dy, me);
  D2H(dz, v.dz, m);
  D2H(dw, v.dw, m);
#undef D2H
  return 0;
}

// after the stream has been drained: the packed results to the caller's vectors
static void unstage(hqpkkt_t *h, double *dx, double *dy, double *dz, double *dw) {
  if (!h->out_pending) return;
  const int n = h->an.n, me = h->an.me, m = h->an.m;
  const double *q = h->out_pending;
  h->out_pending = nullptr;
  if (dx && n) std::memcpy(dx, q, sizeof(double) * n);
  if (dy && me) std::memcpy(dy, q + n, sizeof(double) * me);
  if (dz && m) std::memcpy(dz, q + n + me, sizeof(double) * m);
  if (dw && m) std::memcpy(dw, q + n + me + m, sizeof(double) * m);
}

static const int FWD_FUSED_MAX_SLABS = 1024;  // above: forward step of a level in two launches
static_assert(FS_MAXP == kktdev::SMALL_PIVOTS && FS_MAXB == kktdev::SMALL_BORDER, "small-supernode kernels and schedule disagree");
// ------------------------------------------------------------ numeric phases
// phases: 1 = assemble + this rank's subtrees, 2 = replicated top of the tree
// (3 = everything, the single-rank case)
// (HQPKKT_NO_FUSED_VECTORS=1: the vector work around the sweeps and the assembly as the separate launches of round 5 - the
// comparison the bit-identity test makes)
static bool no_fused_vectors() { return getenv("HQPKKT_NO_FUSED_VECTORS") != nullptr; }
static int run_factor(hqpkkt_t *h, const double *z, const double *w, int phases) {
  Analysis &an = h->an;
  hipStream_t s = h->stream;
  const int m = an.m, nent = (int)an.ent_a.size();
  DevTree T = h->tree();
  if (phases & 1) {
    if (an.shard_count <= 1) {  // the panel arena, and the status words, counters and the two maxima
      KLAUNCH(h, KC_ASSEMBLE, k_clear<<<(int)std::max<long long>(1, std::min<long long>(2048, (an.panel_elems / 2 + 1023) / 1024)), 256, 0, s>>>(h->panel.p, an.panel_elems, h->flags.p));
    } else {  // only the blocks this rank writes
      const int np = (int)an.zero_panel.size() / 2;
      if (np) k_zero_ranges<<<dim3(512, np), 256, 0, s>>>(h->panel.p, h->zero_panel.p);
      k_clear<<<1, 256, 0, s>>>(nullptr, 0, h->flags.p);
    }
    if (!h->capturing) HIPCHK(hipEventRecord(h->ev0, s));
    if (m > 0 && (h->simple_src.count || no_fused_vectors()))
      KLAUNCH(h, KC_ASSEMBLE, k_weights<<<nblk(m), 256, 0, s>>>(an.mode, m, an.n + an.me, z, w, h->wt.p, h->sc.p, h->flags.p));
    if (h->simple_src.count) {  // FULL: one pass
      KLAUNCH(h, KC_ASSEMBLE, k_assemble_simple<<<std::min(nblk(nent), 2048), 256, 0, s>>>(
                                  nent, h->simple_src.p, h->simple_wi.p, h->ent_a.p, h->ent_b.p, h->ent_dst.p,
                                  h->vals.p, h->wt.p, h->sc.p, h->panel.p, h->bits.p, h->tree_words.p + 1));
    } else {
      // weights + entry values, scales + scatter: one launch each (kernels.hip.h, k_wt_entry / k_scale_scatter)
      if (no_fused_vectors()) {
        KLAUNCH(h, KC_ASSEMBLE, k_entry_values<<<nblk(nent), 256, 0, s>>>(nent, h->term_ptr.p, h->terms.p, h->vals.p, h->wt.p,
                                                  h->ent_val.p, h->tree_words.p + 1));
        if (an.mode == 1 && an.n > 0)
          KLAUNCH(h, KC_ASSEMBLE, k_red_scale<<<nblk(an.n), 256, 0, s>>>(an.n, h->diag_ent.p, h->ent_val.p, h->sc.p));
        KLAUNCH(h, KC_ASSEMBLE, k_scatter<<<std::min(nblk(nent), 2048), 256, 0, s>>>(nent, h->ent_a.p, h->ent_b.p, h->ent_dst.p, h->ent_val.p,
                                             h->sc.p, h->panel.p, h->bits.p));
      } else {
      KLAUNCH(h, KC_ASSEMBLE, k_wt_entry<<<nblk(nent) + (m > 0 ? nblk(m) : 0), 256, 0, s>>>(an.mode, m, an.n + an.me, nent, nblk(nent), z, w, h->wt.p, h->sc.p,
                                                h->flags.p, h->term_ptr.p, h->terms.p, h->vals.p, h->ent_val.p, h->tree_words.p + 1));
      const int nsc = std::min(nblk(nent), 2048);
      if (an.mode == 1 && an.n > 0)
        KLAUNCH(h, KC_ASSEMBLE, k_scale_scatter<<<nsc + nblk(an.n), 256, 0, s>>>(an.n, nent, nsc, h->diag_ent.p, h->ent_a.p, h->ent_b.p, h->ent_dst.p,
                                                 h->ent_val.p, h->sc.p, h->panel.p, h->bits.p));
      else
        KLAUNCH(h, KC_ASSEMBLE, k_scatter<<<nsc, 256, 0, s>>>(nent, h->ent_a.p, h->ent_b.p, h->ent_dst.p, h->ent_val.p,
                                             h->sc.p, h->panel.p, h->bits.p));
      }
    }
    if (!h->capturing) HIPCHK(hipEventRecord(h->ev1, s));
  }
  const double alpha = h->opts.tol * 0.6403882032022076;  // tol (1+sqrt 17)/8, hqp/spBKP.C:392
  for (int which = 0; which < 2; which++) {
    if (!(phases & (1 << which))) continue;
    const Analysis::Sched &S = an.sched[which];
    const hqpkkt::DevSched &D = h->ds[which];
    if (S.nnodes == 0) continue;
    const TreeXchgF txf{h->tree_u.p, an.upd_elems, h->tree_words.p + 1};
    if (which == 0 && h->tree_factor) {  // a tree of small fronts: all levels in one launch
      int ldp = 1, ldb = 1;
      for (int l = 0; l < an.nlevels; l++) ldp = std::max(ldp, S.level_fs_p[l] | 1), ldb = std::max(ldb, S.level_fs_b[l]);
      KLAUNCH(h, KC_FACTOR_DIAG, (k_factor_diag_small<true, true><<<S.nnodes, 64, fs_lds_bytes(true, ldp, ldb), s>>>(T, D.level_nodes.p, h->panel.p,
                                               h->dinv.p, h->ptype.p, h->lperm.p, h->esign.p, h->linv.p, h->linv_off.p, alpha, h->opts.pivot_eps, h->bits.p,
                                               h->flags.p + 1, h->upd.p, h->xar.p, ldp, ldb, txf)));
      continue;
    }
    for (int l = 0; l < an.nlevels; l++) {
      const int nn = S.level_ptr[l + 1] - S.level_ptr[l], nfs = S.level_fsmall[l], nsm = S.level_small[l];
      if (nfs > 0) {  // small fronts: extend-add, pivot block, panel and update in one kernel
        const int ldp = S.level_fs_p[l] | 1, ldb = S.level_fs_b[l];
        KLAUNCH(h, KC_FACTOR_DIAG, k_factor_diag_small<true><<<nfs, 64, fs_lds_bytes(true, ldp, ldb), s>>>(T, D.level_nodes.p + S.level_ptr[l], h->panel.p,
                                                 h->dinv.p, h->ptype.p, h->lperm.p, h->esign.p, h->linv.p, h->linv_off.p, alpha, h->opts.pivot_eps, h->bits.p,
                                                 h->flags.p + 1, h->upd.p, h->xar.p, ldp, ldb, txf));
      }
      if (nsm > 0) {
        const int ldp = S.level_sm_p[l] | 1;
        KLAUNCH(h, KC_FACTOR_DIAG, k_factor_diag_small<false><<<nsm, 64, fs_lds_bytes(false, ldp, 1), s>>>(T, D.level_nodes.p + S.level_ptr[l] + nfs, h->panel.p,
                                                 h->dinv.p, h->ptype.p, h->lperm.p, h->esign.p, h->linv.p, h->linv_off.p, alpha, h->opts.pivot_eps, h->bits.p,
                                                 h->flags.p + 1, h->upd.p, h->xar.p, ldp, 1, txf));
      }
      if (nn > nfs + nsm && h->old_fd)
        KLAUNCH(h, KC_FACTOR_DIAG, k_factor_diag<<<nn - nfs - nsm, FD_THREADS, h->lds_diag, s>>>(T, D.level_nodes.p + S.level_ptr[l] + nfs + nsm, h->panel.p,
                                                 h->dinv.p, h->ptype.p, h->lperm.p, h->esign.p, h->linv.p, h->linv_off.p, alpha, h->opts.pivot_eps, h->bits.p,
                                                 h->flags.p + 1, h->upd.p));
      else if (nn > nfs + nsm) {
        // the pivot blocks on the matrix pipe: 8 wavefronts (two workgroups per CU) for levels of <= 128 pivots, 12
        // wavefronts beyond, each holding as many 16 x 16 blocks of the triangle as the level's largest front needs
        const int lmp = h->level_maxp[which][l];
        if (lmp <= 128)
          KLAUNCH(h, KC_FACTOR_DIAG, (k_factor_blk<8, 6, 144, 2, FB_OWNSIMD><<<nn - nfs - nsm, 512, fb_lds_bytes(lmp), s>>>(T, D.level_nodes.p + S.level_ptr[l] + nfs + nsm, h->panel.p,
                                                 h->dinv.p, h->ptype.p, h->lperm.p, h->esign.p, h->linv.p, h->linv_off.p, alpha, h->opts.pivot_eps, h->bits.p,
                                                 h->flags.p + 1, h->upd.p)));
        else if (lmp <= 160)  // (55 blocks on 11 wavefronts: five per wavefront)
          KLAUNCH(h, KC_FACTOR_DIAG, (k_factor_blk<12, FB_NS160, 208, 3, FB_OWNSIMD><<<nn - nfs - nsm, 768, fb_lds_bytes(lmp), s>>>(T, D.level_nodes.p + S.level_ptr[l] + nfs + nsm, h->panel.p,
                                                 h->dinv.p, h->ptype.p, h->lperm.p, h->esign.p, h->linv.p, h->linv_off.p, alpha, h->opts.pivot_eps, h->bits.p,
                                                 h->flags.p + 1, h->upd.p)));
        else if (lmp <= 176)  // (66 blocks of the triangle on 11 wavefronts: six per wavefront - 16 registers fewer than with eight, no scratch)
          KLAUNCH(h, KC_FACTOR_DIAG, (k_factor_blk<12, 6, 208, 3, FB_OWNSIMD><<<nn - nfs - nsm, 768, fb_lds_bytes(lmp), s>>>(T, D.level_nodes.p + S.level_ptr[l] + nfs + nsm, h->panel.p,
                                                 h->dinv.p, h->ptype.p, h->lperm.p, h->esign.p, h->linv.p, h->linv_off.p, alpha, h->opts.pivot_eps, h->bits.p,
                                                 h->flags.p + 1, h->upd.p)));
        else
          KLAUNCH(h, KC_FACTOR_DIAG, (k_factor_blk<12, 8, 208, 3, FB_OWNSIMD><<<nn - nfs - nsm, 768, fb_lds_bytes(lmp), s>>>(T, D.level_nodes.p + S.level_ptr[l] + nfs + nsm, h->panel.p,
                                                 h->dinv.p, h->ptype.p, h->lperm.p, h->esign.p, h->linv.p, h->linv_off.p, alpha, h->opts.pivot_eps, h->bits.p,
                                                 h->flags.p + 1, h->upd.p)));
      }
      const int ns = S.slab_ptr[l + 1] - S.slab_ptr[l];
      // the work lists go over the XCDs in chunks of about half a front's items (kernels.hip.h, xcd_order; measured on C2:
      // 1.875 -> 1.826 ms per factor + solve; whole fronts per chunk 1.830, contiguous ranges per XCD 1.915)
      const int lmb = h->level_maxb[which][l];
      const int xps = h->xcd_ps >= 0 ? h->xcd_ps : std::max(1, (lmb + 31) / 32);
      const int xtt = (lmb + 63) / 64, xsu = h->xcd_su >= 0 ? h->xcd_su : std::max(1, xtt * (xtt + 1) / 4);
      if (ns > 0)  // (the schedule lists 32-row slabs; the kernel takes 16 rows per workgroup)
        KLAUNCH(h, KC_PANEL_SOLVE, k_panel_solve<<<2 * ns, 256, h->lds_panel, s>>>(T, D.slabs.p + 2 * (size_t)S.slab_ptr[l],
                                                    h->panel.p, h->xar.p, h->dinv.p, h->ptype.p,
                                                    h->lperm.p, h->linv.p, h->linv_off.p, h->upd.p, xps));
      const int nt = S.upd_big_ptr[l] - S.upd_tile_ptr[l], ntb = S.upd_tile_ptr[l + 1] - S.upd_big_ptr[l];
      // a wave holds 32 x 32 of a tile while a level fills the chip; 32 x 16 (two workgroups per tile) on the thin levels above
      if (nt > 0 && nt > h->su1_max)
        KLAUNCH(h, KC_SCHUR_UPDATE, k_schur_update<2><<<nt, 256, 0, s>>>(T, D.upd_tiles.p + 3 * (size_t)S.upd_tile_ptr[l],
                                          h->panel.p, h->xar.p, h->upd.p, xsu));
      else if (nt > 0)
        KLAUNCH(h, KC_SCHUR_UPDATE, k_schur_update<1><<<2 * nt, 256, 0, s>>>(T, D.upd_tiles.p + 3 * (size_t)S.upd_tile_ptr[l],
                                          h->panel.p, h->xar.p, h->upd.p, 2 * xsu));
      if (ntb > 0)
        KLAUNCH(h, KC_SCHUR_UPDATE, (k_schur_update_big<2, 2, 4, 4, 2, 2><<<ntb, 256, 0, s>>>(T, D.upd_tiles.p + 3 * (size_t)S.upd_big_ptr[l],
                                          h->panel.p, h->xar.p, h->upd.p)));
    }
  }
  if (!h->capturing) HIPCHK(hipEventRecord(h->evs1, s));
  HIPCHK(hipGetLastError());
  return 0;
}

// phases: 1 = right-hand side + forward sweep over this rank's subtrees,
// 2 = forward / backward over the replicated top, backward over the subtrees,
// 4 = unscale + scatter of the solution (7 = everything, the single-rank case)
static int run_step(hqpkkt_t *h, const Vecs &v, int phases) {
  Analysis &an = h->an;
  hipStream_t s = h->stream;
  const int n = an.n, me = an.me, m = an.m, dim = an.dim;
  DevTree T = h->tree();
  auto forward = [&](int which) -> int {
    const Analysis::Sched &S = an.sched[which];
    const hqpkkt::DevSched &D = h->ds[which];
    const TreeXchg tx{h->tree_x.p, h->tree_x.p + 2 * an.cb_elems, an.cb_elems, an.dim, h->tree_words.p, h->flags.p};
    if (which == 0 && h->small_tree) {  // all levels in one launch
      KLAUNCH(h, KC_SOLVE_FWD,
              k_solve_fwd_small<true><<<S.nnodes, 64, 0, s>>>(T, D.level_nodes.p, h->panel.p, h->linv.p, h->linv_off.p, h->dinv.p, h->ptype.p,
                                                              h->lperm.p, h->rhs.p, h->xsol.p, h->ytmp.p, h->cb.p, tx));
      return 0;
    }
    const int lend = which == 0 && h->top_n > 0 ? h->top_lt : an.nlevels;  // (the levels above: k_solve_top)
    for (int l = 0; l < lend && S.nnodes; l++) {
      const int nn = S.level_ptr[l + 1] - S.level_ptr[l], nfs = S.level_fsmall[l];
      if (nfs > 0)
        KLAUNCH(h, KC_SOLVE_FWD,
                k_solve_fwd_small<false><<<nfs, 64, 0, s>>>(T, D.level_nodes.p + S.level_ptr[l], h->panel.p, h->linv.p,
                                                     h->linv_off.p, h->dinv.p, h->ptype.p, h->lperm.p, h->rhs.p,
                                                     h->xsol.p, h->ytmp.p, h->cb.p, tx));
      const int ng = S.gslab_ptr[l + 1] - S.gslab_ptr[l];  // (front, 64-row slab), at least one per front
      if (ng > 0 && ng <= FWD_FUSED_MAX_SLABS)  // a handful of fronts: the launch is what costs
        KLAUNCH(h, KC_SOLVE_FWD,
                k_solve_fwd<<<ng, 256, 0, s>>>(T, D.gslabs.p + 2 * (size_t)S.gslab_ptr[l], h->panel.p, h->linv.p,
                                               h->linv_off.p, h->dinv.p, h->ptype.p, h->lperm.p, h->rhs.p,
                                               h->xsol.p, h->ytmp.p, h->cb.p));
      else if (ng > 0) {  // thousands of slabs: M once per front, then the slabs
        KLAUNCH(h, KC_SOLVE_FWD,
                k_solve_fwd_a<<<nn - nfs, 256, 0, s>>>(T, D.level_nodes.p + S.level_ptr[l] + nfs, h->linv.p,
                                                 h->linv_off.p, h->dinv.p, h->ptype.p, h->lperm.p,
                                                 h->rhs.p, h->xsol.p, h->ytmp.p, h->cb.p));
        KLAUNCH(h, KC_SOLVE_FWD,
                k_solve_fwd_b<<<ng, 256, 0, s>>>(T, D.gslabs.p + 2 * (size_t)S.gslab_ptr[l], h->panel.p,
                                                 h->ytmp.p, h->cb.p));
      }
    }
    return 0;
  };
  auto backward = [&](int which) -> int {
    const Analysis::Sched &S = an.sched[which];
    const hqpkkt::DevSched &D = h->ds[which];
    const TreeXchg tx{h->tree_x.p, h->tree_x.p + 2 * an.cb_elems, an.cb_elems, an.dim, h->tree_words.p, h->flags.p};
    if (which == 0 && h->small_tree) {
      KLAUNCH(h, KC_SOLVE_BWD, k_solve_bwd_small<true><<<S.nnodes, 64, 0, s>>>(T, h->tree_down.p, h->panel.p, h->linv.p, h->linv_off.p, h->lperm.p,
                                                                               h->xsol.p, tx));
      return 0;
    }
    const int lbeg = which == 0 && h->top_n > 0 ? h->top_lt - 1 : an.nlevels - 1;
    for (int l = lbeg; l >= 0 && S.nnodes; l--) {
      const int nn = S.level_ptr[l + 1] - S.level_ptr[l], nfs = S.level_fsmall[l];
      const int ncb = S.cblk_ptr[l + 1] - S.cblk_ptr[l];
      if (nn <= 0) continue;
      if (nfs > 0)
        KLAUNCH(h, KC_SOLVE_BWD,
                k_solve_bwd_small<false><<<nfs, 64, 0, s>>>(T, D.level_nodes.p + S.level_ptr[l], h->panel.p, h->linv.p,
                                                     h->linv_off.p, h->lperm.p, h->xsol.p, tx));
      if (nn <= nfs) continue;
      // (one workgroup per front doing both steps was measured slower: L21' x needs the
      // column blocks spread over the chip)
      KLAUNCH(h, KC_SOLVE_BWD,
              k_solve_bwd_b<<<ncb, 256, h->lds_bwdb, s>>>(T, D.cblks.p + 2 * (size_t)S.cblk_ptr[l],
                                                          h->panel.p, h->xsol.p, h->vtmp.p));
      KLAUNCH(h, KC_SOLVE_BWD,
              k_solve_bwd_a<<<nn - nfs, 256, 0, s>>>(T, D.level_nodes.p + S.level_ptr[l] + nfs, h->linv.p,
                                               h->linv_off.p, h->lperm.p, h->vtmp.p, h->xsol.p));
    }
    return 0;
  };
  if (phases & 1) {
    if (an.mode == 0) {
      KLAUNCH(h, KC_VECTOR, k_rhs_full<<<nblk(dim), 256, 0, s>>>(n, me, m, h->q2e.p, h->sc.p, v.z, v.r1, v.r2, v.r3, v.r4,
                                           h->rhs.p, h->tree_words.p));
    } else {
      // (tz and the right-hand side that needs it in one launch: kernels.hip.h, k_rhs_red_t; HQPKKT_NO_FUSED_VECTORS=1: two)
      if (no_fused_vectors()) {
        if (m > 0) KLAUNCH(h, KC_VECTOR, k_red_t<<<nblk(m), 256, 0, s>>>(m, v.w, h->wt.p, v.r3, v.r4, h->tz.p));
        KLAUNCH(h, KC_VECTOR, k_rhs_red<<<nblk(dim), 256, 0, s>>>(n, me, h->q2e.p, h->sc.p, h->CT.ptr.p, h->CT.col.p,
                                            h->CT.src.p, h->vals.p, h->tz.p, v.r1, v.r2, h->rhs.p,
                                            h->tree_words.p));
      } else
        KLAUNCH(h, KC_VECTOR, k_rhs_red_t<<<nblk(dim) + (m > 0 ? nblk(m) : 0), 256, 0, s>>>(n, me, m, nblk(dim), h->q2e.p, h->sc.p, h->CT.ptr.p, h->CT.col.p,
                                            h->CT.src.p, h->vals.p, v.w, h->wt.p, v.r3, v.r4, h->tz.p, v.r1, v.r2, h->rhs.p,
                                            h->tree_words.p));
    }
    forward(0);
  }
  if (phases & 2) {
    forward(1);
    if (h->top_n > 0) {  // the top levels, up and down: one launch, or one per sweep (k_solve_top)
      TopArgs ta{h->top_nodes.p, h->top_idx.p, h->top_bpos.p, h->top_x.p, h->top_x.p + 2 * (size_t)h->top_n * ST_CS, h->tree_words.p, h->top_n, h->top_stamps};
#define TOP_LAUNCH(NS, NU, MODE)                                                                                                          \
  KLAUNCH(h, KC_SOLVE_TOP, (k_solve_top<NS, NU, MODE><<<h->top_n, ST_THREADS, h->top_lds, s>>>(T, ta, h->panel.p, h->linv.p, h->linv_off.p, \
                                                            h->dinv.p, h->ptype.p, h->lperm.p, h->rhs.p, h->xsol.p, h->cb.p, h->flags.p)))
      if (!h->top_split) {
        if (h->top_ns == 3) TOP_LAUNCH(3, 11, 0); else TOP_LAUNCH(4, 10, 0);
      } else {
        ta.nodes = h->top_up.p;
        if (h->top_ns == 3) TOP_LAUNCH(3, 11, 1); else TOP_LAUNCH(4, 10, 1);
        ta.nodes = h->top_nodes.p;
        if (h->top_ns == 3) TOP_LAUNCH(3, 11, 2); else TOP_LAUNCH(4, 10, 2);
      }
#undef TOP_LAUNCH
    }
    backward(1);
    backward(0);
    if (an.shard_count > 1)  // leave only this rank's share for the all-reduce
      KLAUNCH(h, KC_VECTOR, k_mask_vector<<<nblk(dim), 256, 0, s>>>(dim, h->keep_e.p, h->xsol.p));
  }
  if (phases & 4) {
    if (an.mode == 0) {
      KLAUNCH(h, KC_VECTOR, k_unpack_full<<<nblk(dim), 256, 0, s>>>(n, me, m, h->q2e.p, h->sc.p, h->xsol.p, v.dx, v.dy,
                                              v.dz));
      if (m > 0)
        KLAUNCH(h, KC_VECTOR, k_dw<<<nblk(m), 256, 0, s>>>(m, h->C.ptr.p, h->C.col.p, h->C.src.p, h->vals.p, v.dx, v.r3,
                                     v.z, v.w, v.r4, v.dz, v.dw));
    } else {
      // (dx, dy and the dz, dw that need dx in one launch: kernels.hip.h, k_unpack_dzdw)
      const int nb_dzdw = m > 0 ? nblk(m) : 0;
      if (no_fused_vectors()) {
        KLAUNCH(h, KC_VECTOR, k_unpack_red<<<nblk(dim), 256, 0, s>>>(n, me, h->q2e.p, h->sc.p, h->xsol.p, v.dx, v.dy));
        if (m > 0)
          KLAUNCH(h, KC_VECTOR, k_red_dzdw<<<nblk(m), 256, 0, s>>>(m, h->C.ptr.p, h->C.col.p, h->C.src.p, h->vals.p, v.dx,
                                             h->wt.p, h->tz.p, v.r3, v.dz, v.dw));
      } else
      KLAUNCH(h, KC_VECTOR, k_unpack_dzdw<<<nb_dzdw + nblk(dim), 256, 0, s>>>(n, me, m, nb_dzdw, h->q2e.p, h->sc.p, h->xsol.p, v.dx, v.dy, h->C.ptr.p,
                                           h->C.col.p, h->C.src.p, h->vals.p, h->wt.p, h->tz.p, v.r3, v.dz, v.dw));
    }
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// replay (or first capture) of a kernel sequence as a hipGraph; falls back to
// eager launches while per-kernel profiling is on
template <class F>
static int graphed(hqpkkt_t *h, hqpkkt::GraphSlot &slot, F body) {
  if (!h->use_graphs || h->prof.on || h->capturing) return body();  // (capturing: a sequence inside a segment's capture)
  if (!slot.ge) {
    HIPCHK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    h->capturing = true, h->cap_posts = 0;
    int e = body();
    h->capturing = false;
    hipGraph_t g = nullptr;
    hipError_t ce = hipStreamEndCapture(h->stream, &g);
    if (e) {
      if (g) (void)hipGraphDestroy(g);
      h->post_seq -= h->cap_posts;  // (nothing was posted)
      return e;
    }
    if (ce != hipSuccess || !g) {  // capture not possible: run eagerly from now on
      h->use_graphs = false;
      (void)hipGetLastError();
      h->post_seq -= h->cap_posts;
      return body();
    }
    slot.g = g;
    if (hipGraphInstantiate(&slot.ge, g, nullptr, nullptr, 0) != hipSuccess) {
      slot.drop();
      h->use_graphs = false;
      (void)hipGetLastError();
      h->post_seq -= h->cap_posts;
      return body();
    }
    slot.n_posts = h->cap_posts;  // (the host has counted them during the capture)
  } else
    h->post_seq += slot.n_posts;
  HIPCHK(hipGraphLaunch(slot.ge, h->stream));
  return 0;
}

// The exchange steps of a sharded system: the handle's stream is
// drained, the caller's collective runs, and the next phase starts afterwards.
static int exchange(hqpkkt_t *h, int op, double *buf, long long slot, int nslots, hipStream_t on = nullptr) {
  // (profiled as the class "exchange": in the stream-ordered form the time between the collective's place in
  // the stream and its completion - the wait for the slowest rank and the transfer)
  if (h->xchg_sfn) {  // the collective is put into the handle's stream (or `on`) behind the kernels that fill `buf`
    hipStream_t st = on ? on : h->stream;
    h->prof.begin(KC_XCHG, st);
    const int rc = h->xchg_sfn(h->xchg_ctx, op, buf, slot, nslots, (void *)st);
    h->prof.end(st);
    return rc ? HQPKKT_E_DEVICE : 0;
  }
  if (!h->xchg_fn) return HQPKKT_E_INTERN;
  h->prof.begin(KC_XCHG, h->stream);
  HIPCHK(hipStreamSynchronize(h->stream));
  const int rc = h->xchg_fn(h->xchg_ctx, op, buf, slot, nslots);
  h->prof.end(h->stream);
  return rc ? HQPKKT_E_DEVICE : 0;
}

static int staged_run_factor(hqpkkt_t *h, const double *z, const double *w);
static int staged_run_step(hqpkkt_t *h, const Vecs &v);
static bool staged_is_sharded(hqpkkt_t *h);

static int do_factor(hqpkkt_t *h, const Vecs &v) {
  Analysis &an = h->an;
  int e;
  if (h->opts.mode == HQPKKT_MODE_STAGED) {
    if (staged_is_sharded(h)) return staged_run_factor(h, v.z, v.w);  // an exchange per stage: not captured
    return graphed(h, h->gfactor[0], [&]() { return staged_run_factor(h, v.z, v.w); });
  }
  if (an.shard_count <= 1) {
    if (an.m > 0 && v.z != h->vin.p) {  // the caller's device vectors themselves (direct_vectors)
      const void *key[10] = {v.z, v.w};
      return graphed(h, h->direct_slot(h->gdirect_factor, key), [&]() { return run_factor(h, v.z, v.w, 3); });
    }
    return graphed(h, h->gfactor[0], [&]() { return run_factor(h, v.z, v.w, 3); });
  }
  if ((e = graphed(h, h->gfactor[0], [&]() { return run_factor(h, v.z, v.w, 1); }))) return e;
  if (an.upd_x_slot > 0 &&
      (e = exchange(h, HQPKKT_XCHG_ALLGATHER, h->upd.p + an.upd_x_off, an.upd_x_slot, an.shard_count)))
    return e;
  if ((e = graphed(h, h->gfactor[1], [&]() { return run_factor(h, v.z, v.w, 2); }))) return e;
  // A zero pivot inside a subtree is seen by its owner only: agree on the status words (one small
  // all-reduce), so that every rank returns the same code and nobody waits in a collective alone
  k_status_pack<<<1, 64, 0, h->stream>>>(h->flags.p, h->bits.p, h->ytmp.p);
  if ((e = exchange(h, HQPKKT_XCHG_ALLREDUCE_SUM, h->ytmp.p, 4, 1))) return e;
  k_status_unpack<<<1, 64, 0, h->stream>>>(h->ytmp.p, h->flags.p, h->bits.p);
  return 0;
}

static int do_step(hqpkkt_t *h, const Vecs &v, int which) {
  Analysis &an = h->an;
  int e;
  if (h->opts.mode == HQPKKT_MODE_STAGED) {
    if (staged_is_sharded(h)) return staged_run_step(h, v);  // exchanges inside the sweeps: not captured
    return graphed(h, h->gstep[which][0], [&]() { return staged_run_step(h, v); });
  }
  if (an.shard_count <= 1) {
    // the caller's device vectors themselves (direct_vectors): also the refinement's sequence (which == 1: residual and
    // correction vectors are the handle's, z and w the caller's)
    if ((which == 0 && v.dx != h->vout.p) || (an.m > 0 && v.z != h->vin.p)) {
      const void *key[10] = {v.z, v.w, v.r1, v.r2, v.r3, v.r4, v.dx, v.dy, v.dz, v.dw};
      return graphed(h, h->direct_slot(h->gdirect_step, key), [&]() { return run_step(h, v, 7); });
    }
    return graphed(h, h->gstep[which][0], [&]() { return run_step(h, v, 7); });
  }
  if ((e = graphed(h, h->gstep[which][0], [&]() { return run_step(h, v, 1); }))) return e;
  if (an.cb_x_slot > 0 &&
      (e = exchange(h, HQPKKT_XCHG_ALLGATHER, h->cb.p + an.cb_x_off, an.cb_x_slot, an.shard_count)))
    return e;
  if ((e = graphed(h, h->gstep[which][1], [&]() { return run_step(h, v, 2); }))) return e;
  if ((e = exchange(h, HQPKKT_XCHG_ALLREDUCE_SUM, h->xsol.p, an.dim, 1))) return e;
  return graphed(h, h->gstep[which][2], [&]() { return run_step(h, v, 4); });
}

// residual of (d) for rhs (r); leaves the residual vectors in h->vres
// out != nullptr: the caller's copy of (d) is put into the stream before the read-back, so
// that a solve that needs no refinement round is over with this one round trip
// ---- read-backs through mapped host memory (hqpkkt::hpin_dev)
// the status words (and, with `out`, n_out <= 40 of the IP loop's scalars) as they stand at this point of the stream
static int post_words(hqpkkt_t *h, const double *out, int n_out, bool residual = false) {
  h->post_seq++;
  if (h->capturing) h->cap_posts++;
  k_post_words<<<1, 64, 0, h->stream>>>(h->flags.p, out, n_out, h->hpin_dev, h->post_seq_dev.p, residual ? 1 : 0);
  return 0;
}
// waits until the last posted words have arrived (every earlier post of the stream has then arrived as well)
static int post_wait(hqpkkt_t *h) {
  volatile unsigned *seq = (volatile unsigned *)(h->hpin + HPIN_SEQ);
  for (long long spin = 0;; spin++) {
    if (*seq == h->post_seq) break;
    if ((spin & 0xfffff) == 0xfffff) {  // (about every millisecond: has the stream died or drained without the word?)
      const hipError_t q = hipStreamQuery(h->stream);
      if (q == hipSuccess) {
        if (*seq == h->post_seq) break;
        (void)snprintf(g_last_hip_error, sizeof(g_last_hip_error), "posted read-back: the stream is empty and the sequence word is %u, not %u", *seq, h->post_seq);
        return HQPKKT_E_DEVICE;
      }
      if (q != hipErrorNotReady) HIPCHK(q);
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return 0;
}

struct OutPtrs {
  double *dx, *dy, *dz, *dw;
};
static int staged_dense_products(hqpkkt_t *h, const Vecs &v, const double **x1, const double **x2, int *ndyn);

static int collect_residual(hqpkkt_t *h, double *res);
// the residual kernel alone (what run_residual puts into the stream first)
static int residual_launch(hqpkkt_t *h, const Vecs &v) {
  Analysis &an = h->an;
  hipStream_t s = h->stream;
  const int n = an.n, me = an.me, m = an.m;
  double *o1 = h->vres.p, *o2 = o1 + n, *o3 = o2 + me, *o4 = o3 + m;
  // the maximum is accumulated in the ints 122-123 of the flags buffer; the posting kernel behind every residual kernel
  // clears it (and a factorisation clears the whole buffer).  (rb_next: the word the kernel zeroes for its successor - a
  // spare one since the posting kernel does that.)
  unsigned long long *const rb_now = h->bits.p + 1, *const rb_next = h->bits.p - 1;
  const double *x1 = nullptr, *x2 = nullptr;  // STAGED, dense dynamics: their share of A dx and A'dy
  int ndyn = 0;
  if (h->opts.mode == HQPKKT_MODE_STAGED) {
    int e1 = staged_dense_products(h, v, &x1, &x2, &ndyn);
    if (e1) return e1;
  }
  if (h->short_rows)
    KLAUNCH(h, KC_RESIDUAL, k_residual<4><<<std::min(nblk(4LL * ((long long)n + me + m)), 1024), 256, 0, s>>>(
        n, me, m, h->Qf.dev(), h->AT.dev(), h->CT.dev(), h->A.dev(), h->C.dev(), h->vals.p, v.z, v.w,
        v.r1, v.r2, v.r3, v.r4, v.dx, v.dy, v.dz, v.dw, o1, o2, o3, o4, rb_now, rb_next, x1, x2, ndyn));
  else
    KLAUNCH(h, KC_RESIDUAL, k_residual<16><<<std::min(nblk(16LL * ((long long)n + me + m)), 1024), 256, 0, s>>>(
        n, me, m, h->Qf.dev(), h->AT.dev(), h->CT.dev(), h->A.dev(), h->C.dev(), h->vals.p, v.z, v.w,
        v.r1, v.r2, v.r3, v.r4, v.dx, v.dy, v.dz, v.dw, o1, o2, o3, o4, rb_now, rb_next, x1, x2, ndyn));
  return 0;
}
static int run_residual(hqpkkt_t *h, const Vecs &v, double *res, const OutPtrs *out = nullptr) {
  hipStream_t s = h->stream;
  {
    const int e1 = residual_launch(h, v);
    if (e1) return e1;
  }
  if (out) {
    int e2 = stage_out(h, v, out->dx, out->dy, out->dz, out->dw);
    if (e2) return e2;
  }
  // one read-back: the residual maximum and the status of the factorisation this solve belongs to
  int ep;
  if ((ep = post_words(h, nullptr, 0, true))) return ep;
  if (h->defer_residual && !out) {  // the caller queues more work and waits once (collect_residual)
    h->residual_pending = true;
    *res = 0.0;
    return 0;
  }
  if (out && !(h->out_pending && h->out_by_kernel)) HIPCHK(hipStreamSynchronize(s));  // (the caller's vectors: copies into pageable memory have landed)
  if ((ep = post_wait(h))) return ep;
  return collect_residual(h, res);
}

// the words run_residual copied to the pinned buffer, after the stream has been waited for
static int collect_residual(hqpkkt_t *h, double *res) {
  h->residual_pending = false;
  const bool check = h->factor_unchecked;
  int *hs = (int *)h->hpin;
  int flags[4] = {hs[0], hs[1], hs[2], hs[3]};
  if (poll_fallback(h, hs)) {  // a polled launch gave up waiting for a word: no result, and per-level launches from now on
    if (h->factor_unchecked) h->factor_unchecked = false, h->factored = false;  // (the factorisation may be the one that gave up)
    return HQPKKT_E_POLL;
  }
  unsigned long long kb, bits;
  std::memcpy(&kb, hs + 120, sizeof(kb)), std::memcpy(&bits, hs + h->res_read, sizeof(bits));
  double r;
  std::memcpy(&r, &bits, sizeof(r));
  *res = r;
  if (check) {
    h->factor_unchecked = false;
    std::memcpy(&h->st.kmax, &kb, sizeof(kb));
    h->st.n_2x2 = flags[1], h->st.n_perturbed = flags[2], h->st.n_slow_pivots = flags[3];
    h->soft_singular = hs[4] != 0;
    h->soft_tiny = hs[5] != 0;
    if (getenv("HQPKKT_TRACE_SOLVE") && (flags[0] || hs[4] || hs[5]))
      fprintf(stderr, "factor (checked with the solve): status %d, perturbed %d, zero pivot perturbed %d, tiny multiplier pivot %d\n", flags[0],
              flags[2], hs[4], hs[5]);
    if (flags[0] || std::isinf(h->st.kmax)) {
      h->factored = false;
      return flags[0] ? flags[0] : HQPKKT_E_SING;
    }
  }
  return 0;
}

static float elapsed(hipEvent_t a, hipEvent_t b) {
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, a, b) != hipSuccess) ms = -1.f;
  return ms;
}

#include "staged_host.hip.h"
static int staged_dense_products(hqpkkt_t *h, const Vecs &v, const double **x1, const double **x2, int *ndyn) {
  StagedDev &d = *h->sd;
  const kktdev::StagedPlan &P = d.plan;
  if (!P.dense_dyn) return 0;
  int nzmax = 1, npmax = 1;
  for (int k = 0; k < P.K; k++) nzmax = std::max(nzmax, P.nk[k] + P.mk[k]), npmax = std::max(npmax, P.nk[k + 1]);
  nzmax = std::max(nzmax, P.nk[P.K]);
  if (P.sharded) {
    // the rank's share of both products from its local blocks (staged.hip.h, DynLoc), summed over the ranks
    const long long tot = d.dyn_sum_x2 + P.ndyn;
    KLAUNCH(h, KC_RESIDUAL, stg::k_st_zero<<<nblk(tot), 256, 0, h->stream>>>(tot, d.dyn_sum.p));
    KLAUNCH(h, KC_RESIDUAL, stg::k_st_dynloc_ax<<<dim3(std::min((npmax + 3) / 4, 2048), P.K), 256, 0, h->stream>>>(d.dyn_loc.p, d.F.p, v.dx,
                                                                                                           d.dyn_sum.p + d.dyn_sum_x2));
    KLAUNCH(h, KC_RESIDUAL, stg::k_st_dynloc_aty<<<dim3((nzmax + 255) / 256, P.K + 1), 256, 0, h->stream>>>(d.dyn_loc.p, d.F.p, v.dy, d.dyn_sum.p));
    int e = exchange(h, HQPKKT_XCHG_ALLREDUCE_SUM, d.dyn_sum.p, tot, 1);
    if (e) return e;
    *x1 = d.dyn_sum.p, *x2 = d.dyn_sum.p + d.dyn_sum_x2, *ndyn = P.ndyn;
    return 0;
  }
  const bool two_passes = false;  // (one pass over F for both products; the two-pass kernels stay for blocks the fused one does not take)
  const int nbc = (nzmax + 255) / 256;
  if (!two_passes && d.dyn_part.p && d.dyn_part_cols == nbc) {
    // one pass over F for both products (k_st_dyn_both), then the row sums' column blocks
    KLAUNCH(h, KC_RESIDUAL, stg::k_st_dyn_both<<<dim3(nbc, P.K + 1), 256, 0, h->stream>>>(d.dyn_desc.p, d.F.p, v.dx, v.dy, d.dyn_x1.p,
                                                                                        d.dyn_part.p, nbc));
    KLAUNCH(h, KC_RESIDUAL, stg::k_st_dyn_ax_finish<<<dim3((npmax + 255) / 256, P.K), 256, 0, h->stream>>>(d.dyn_desc.p, d.dyn_part.p, nbc,
                                                                                                     v.dx, d.dyn_x2.p));
  } else {
    KLAUNCH(h, KC_RESIDUAL, stg::k_st_dyn_ax<<<dim3(std::min((npmax + 3) / 4, 2048), P.K), 256, 0, h->stream>>>(d.dyn_desc.p, d.F.p, v.dx,
                                                                                                          d.dyn_x2.p));
    KLAUNCH(h, KC_RESIDUAL, stg::k_st_dyn_aty<<<dim3((nzmax + 255) / 256, P.K + 1), 256, 0, h->stream>>>(d.dyn_desc.p, d.F.p, v.dy,
                                                                                                   d.dyn_x1.p));
  }
  *x1 = d.dyn_x1.p, *x2 = d.dyn_x2.p, *ndyn = P.ndyn;
  return 0;
}
static bool staged_is_sharded(hqpkkt_t *h) { return h->sd && h->sd->plan.sharded; }
static void staged_release(StagedDev *sd, bool destroy) {
  if (!sd) return;
  sd->release();
  if (destroy) delete sd;
}

// =========================================================================
// The C ABI promises that nothing is thrown across it (the shim's callers longjmp through Meschach
// frames): every entry point that allocates with the standard library runs inside this guard.
template <class F>
static int guarded(F body) {
  try {
    return body();
  } catch (const std::bad_alloc &) {
    return HQPKKT_E_MEM;
  } catch (...) {
    return HQPKKT_E_INTERN;
  }
}

// The attempts of a device-resident loop (hqpkkt_mehrotra / hqpkkt_franke; `loop` reads the caller's `opts` through the
// reference it captured).  ONE rule ("Pivoting"): the factors are first those the reference's own
// loop gets from this plugin through the shim - a multiplier pivot that cancelled to rounding level is used as it
// stands -, and STATIC PIVOTING is the fall-back: a run that ends "degenerate" (or singular) is made again, from a cold
// start, with such pivots replaced (kernels.hip.h, TINY_REPLACE_WORD; what the reference's own PARDISO plugin is
// configured to do, hqp/Hqp_IpPARDISO.C:138-142).  hqpkkt_ip_result.attempts says which happened; plugin calls and device
// time are the totals over the attempts, `iters` is the count of the run that produced the result.  HQPKKT_TINY_IN_LOOP=1 / =0 (campaign switches): static
// pivoting from the first attempt on / never.
template <class Loop>
static int ip_attempts(hqpkkt_t *h, const hqpkkt_ip_opts *&opts, hqpkkt_ip_result *res, Loop loop) {
  hqpkkt_ip_opts again;
  auto cold = [&]() {  // (a first attempt has used up what a hot start would start from; "2" keeps what the NEXT call needs)
    if (opts && opts->hot_start == 1) {
      again = *opts;
      again.hot_start = 2;
      opts = &again;
    }
  };
  // (a polled launch that gave up has switched the handle to the per-level launches: the loop runs once more - from a cold
  // start: the aborted pass has left its own iterates in the loop's vectors and may have overwritten the hot-start candidates)
  auto attempt = [&]() {
    int rc = guarded(loop);
    if (rc == HQPKKT_E_POLL) {
      cold();
      if (h) h->ip_hot_valid = false, h->fr_hot_valid = false;
      rc = guarded(loop);
    }
    return rc == HQPKKT_E_POLL ? HQPKKT_E_DEVICE : rc;
  };
  static const char *const pol = getenv("HQPKKT_TINY_IN_LOOP");
  if (h) h->tiny_replace_in_loop = pol && atoi(pol) == 1;
  int rc = attempt();
  if (res && rc == 0) res->attempts = 1;
  if (h && res && !h->tiny_replace_in_loop && !(pol && atoi(pol) == 0) && (rc == HQPKKT_E_SING || (rc == 0 && res->result == 4))) {
    const hqpkkt_ip_result first = *res;  // (all zero but `result` when the first attempt ended with E_SING before its finish())
    const bool counted = rc == 0;
    h->tiny_replace_in_loop = true;
    cold();
    rc = attempt();
    h->tiny_replace_in_loop = false;
    if (rc == 0) {
      res->attempts = 2;
      if (counted)  // the work of both runs; `iters` stays the count of the run that gave the result (what the reference's count is compared with)
        res->n_factor += first.n_factor, res->n_solve += first.n_solve, res->ms_total += first.ms_total;
    }
  }
  return rc;
}

extern "C" {

int hqpkkt_default_opts(hqpkkt_opts *o) {
  if (!o) return HQPKKT_E_NULL;
  std::memset(o, 0, sizeof(*o));
  o->mode = HQPKKT_MODE_FULL;
  o->device = 0;
  o->loc = HQPKKT_LOC_HOST;
  o->tol = 1.0;    // hqp/Hqp_IpSpBKP.C:46
  o->eps = 1e-10;  // hqp/Hqp_IpMatrix.C:45
  o->pivot_eps = 1e-20;  // only (near-)exact zeros are replaced: the reference accepts any non-zero pivot
  o->leaf_size = 0;
  o->max_pivots = 0;
  o->zd_policy = -1;  // by the values (hqpkkt_set_values)
  o->slack_policy = 2;
  return 0;
}

int hqpkkt_create(const hqpkkt_opts *opts, hqpkkt_t **out) {
  if (!out) return HQPKKT_E_NULL;
  hqpkkt_opts o;
  if (opts)
    o = *opts;
  else
    hqpkkt_default_opts(&o);
  if (o.mode != HQPKKT_MODE_FULL && o.mode != HQPKKT_MODE_REDUCED && o.mode != HQPKKT_MODE_STAGED) return HQPKKT_E_RANGE;
  if (!(o.tol > 0.0 && o.tol <= 1.0)) return HQPKKT_E_RANGE;  // hqp/spBKP.C:389-390
  if (o.loc != HQPKKT_LOC_HOST && o.loc != HQPKKT_LOC_DEVICE) return HQPKKT_E_RANGE;
  hqpkkt_t *h = new (std::nothrow) hqpkkt;
  if (!h) return HQPKKT_E_MEM;
  h->opts = o;
  std::memset(&h->st, 0, sizeof(h->st));
  h->st.sbw = -1;
  *out = h;
  return 0;
}

int hqpkkt_destroy(hqpkkt_t *h) {
  if (!h) return 0;
  if (h->own_stream) {
    (void)hipSetDevice(h->opts.device);
    (void)hipStreamSynchronize(h->own_stream);
    h->release_device();
    (void)hipEventDestroy(h->ev0);
    (void)hipEventDestroy(h->ev1);
    (void)hipEventDestroy(h->evs0);
    (void)hipEventDestroy(h->evs1);
    (void)hipEventDestroy(h->evt0);
    (void)hipEventDestroy(h->evt1);
    h->prof.destroy();
    (void)hipStreamDestroy(h->own_stream);
  }
  staged_release(h->sd, true);
  delete h;
  return 0;
}

static int run_analysis(hqpkkt_t *h, int n, int me, int m, int zd);

int hqpkkt_analyze(hqpkkt_t *h, int n, int me, int m, const int *Qp, const int *Qi,
                   const int *Ap, const int *Ai, const int *Cp, const int *Ci, int *sbw) {
  return guarded([&]() -> int {
    if (!h) return HQPKKT_E_NULL;
    if ((n > 0 && (!Qp || (Qp[n] > 0 && !Qi))) || (me > 0 && (!Ap || (Ap[me] > 0 && !Ai))) ||
        (m > 0 && (!Cp || (Cp[m] > 0 && !Ci))))
      return HQPKKT_E_NULL;
    if (h->uploaded) {
      (void)hipSetDevice(h->opts.device);
      (void)hipStreamSynchronize(h->stream);
      h->release_device();
    }
    h->analyzed = false;
    h->ip_hot_valid = h->fr_hot_valid = false;
    auto keep = [](std::vector<int> &dst, const int *src, size_t k) {
      dst.clear();
      if (src && k) dst.assign(src, src + k);
    };
    keep(h->pQp, Qp, n ? (size_t)n + 1 : 0), keep(h->pQi, Qi, n ? (size_t)Qp[n] : 0);
    keep(h->pAp, Ap, me ? (size_t)me + 1 : 0), keep(h->pAi, Ai, me ? (size_t)Ap[me] : 0);
    keep(h->pCp, Cp, m ? (size_t)m + 1 : 0), keep(h->pCi, Ci, m ? (size_t)Cp[m] : 0);
    h->zd_decided = h->opts.zd_policy >= 0;
    h->zd_weak = false;
    if (h->opts.mode == HQPKKT_MODE_STAGED) {
      h->zd_decided = true;
      int es = staged_analyze(h, n, me, m);
      if (es) return es;
      if (sbw) *sbw = -1;
      return 0;
    }
    int e = run_analysis(h, n, me, m, h->zd_decided ? h->opts.zd_policy : 2);
    if (e) return e;
    if (sbw) *sbw = h->an.sbw;
    return 0;
  });
}

// the symbolic phase with the zero-diagonal policy zd (hqpkkt_analyze; once more from
// hqpkkt_set_values when zd_policy -1 sees weak Hessian diagonals)
static int run_analysis(hqpkkt_t *h, int n, int me, int m, int zd) {
  h->an = Analysis();
  h->an.shard_rank = h->shard_rank, h->an.shard_count = h->shard_count;
  h->an.slack_policy = h->opts.slack_policy;
  h->an.small_fronts = !h->opts.no_small_fronts;
  h->an.amalgamation = h->opts.amalgamation != 0;
  h->an.ordering = (h->opts.ordering == 1 || h->opts.ordering == 2) ? h->opts.ordering : 0;
  if (h->opts.upd_pingpong_mb > 0) h->an.upd_pingpong_bytes = (long long)h->opts.upd_pingpong_mb << 20;
  if (h->opts.upd_pingpong_mb < 0) h->an.upd_pingpong_bytes = 0;
  // pivots per supernode: the caller's number (k_factor_blk takes up to 192), or 160 - ten blocks of 16, the widest
  // front whose M = L11^-1 stays in LDS next to L21 in registers for both sweeps of k_solve_top, so that ALL levels of
  // a banded system's tree go through its two launches (C2: 2.27 ms per factor + solve against 2.33 with 192, where
  // the leaf level stays outside).  HQPKKT_MAX_PIVOTS for same-box comparisons; k_factor_diag of rounds 1-3
  // (HQPKKT_OLD_FD) takes 128.
  // Separators of >= LONG_CHAIN_VERTS vertices (irregular graphs) are cut into pieces of 192 all the same: their fronts go
  // through the per-level kernels anyway, and the update blocks are rewritten 17 % less often (1e6-cell mesh with far
  // couplings: 154 ms per factor + solve against 162).
  int maxp = h->opts.max_pivots;
  h->an.long_chain_pivots = (maxp <= 0 && !getenv("HQPKKT_MAX_PIVOTS")) ? 192 : 0;
  if (maxp <= 0) maxp = getenv("HQPKKT_MAX_PIVOTS") ? std::atoi(getenv("HQPKKT_MAX_PIVOTS")) : 160;
  int e = h->an.run(h->opts.mode, n, me, m, h->pQp.data(), h->pQi.data(), h->pAp.data(), h->pAi.data(),
                    h->pCp.data(), h->pCi.data(), h->opts.leaf_size, maxp, zd);
  if (e) return e;
  h->zd_used = zd;
  h->analyzed = true;
  Analysis &an = h->an;
  h->st.dim = an.dim, h->st.sbw = an.sbw, h->st.n_supernodes = an.nnodes;
  h->st.n_levels = an.nlevels, h->st.max_front = an.max_front;
  h->st.nnz_kkt = (long long)an.ent_a.size();
  h->st.nnz_factor = an.nnz_factor, h->st.flops_factor = an.flops_factor;
  h->st.bytes_panels = (long long)sizeof(double) * (an.panel_elems + an.x_elems);
  h->st.bytes_updates = (long long)sizeof(double) * an.upd_elems;
  h->st.shard_rank = an.shard_rank, h->st.shard_count = an.shard_count;
  h->st.n_top = an.sched[1].nnodes, h->st.n_exchange_blocks = (int)an.xroots.size();
  h->st.flops_local = an.sched[0].flops, h->st.flops_top = an.sched[1].flops;
  h->st.bytes_exchange_factor = (long long)sizeof(double) * an.upd_x_slot * (an.xroots.empty() ? 0 : an.shard_count);
  h->st.bytes_exchange_step =
      an.shard_count > 1 ? (long long)sizeof(double) * (an.cb_x_slot * an.shard_count + an.dim) : 0;
  return 0;
}

int hqpkkt_set_values(hqpkkt_t *h, const double *Qx, const double *Ax, const double *Cx) {
  return guarded([&]() -> int {
    if (!h) return HQPKKT_E_NULL;
    if (!h->analyzed) return HQPKKT_E_INTERN;
    Analysis &an = h->an;
    if ((an.nq && !Qx) || (an.na && !Ax) || (an.nc && !Cx)) return HQPKKT_E_NULL;
    int e;
    if (h->opts.mode == HQPKKT_MODE_STAGED) return staged_set_values(h, Qx, Ax, Cx);
    if (!h->uploaded && (e = upload(h))) return e;
    HIPCHK(hipSetDevice(h->opts.device));
    hipMemcpyKind kind =
        h->opts.loc == HQPKKT_LOC_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    if (an.nq) HIPCHK(hipMemcpyAsync(h->vals.p, Qx, sizeof(double) * an.nq, kind, h->stream));
    if (an.na) HIPCHK(hipMemcpyAsync(h->vals.p + an.nq, Ax, sizeof(double) * an.na, kind, h->stream));
    if (an.nc)
      HIPCHK(hipMemcpyAsync(h->vals.p + an.nq + an.na, Cx, sizeof(double) * an.nc, kind, h->stream));
    for (CsrBuf *c : {&h->Qf, &h->A, &h->AT, &h->C, &h->CT})
      if (c->src.count)
        k_gather_values<<<nblk((long long)c->src.count), 256, 0, h->stream>>>((int)c->src.count, c->src.p, h->vals.p, c->val.p);
    if (h->opts.zd_policy < 0 && h->zd_used != 0) {
      // zd_policy -1: an x whose Hessian diagonal is weak against its coupling to an equality needs the
      // 2x2 pivot with that equality's multiplier inside its own pivot block.  Tested on EVERY update (the
      // values of an SQP run change: an identity Hessian may turn weak later), on the device: one pass
      // over Q's diagonal and the columns of A, one word read back.
      h->zd_decided = true;
      int *flag = h->flags.p + 100;
      HIPCHK(hipMemsetAsync(flag, 0, sizeof(int), h->stream));
      if (an.n > 0 && an.me > 0)
        k_zd_weak<<<nblk(an.n), 256, 0, h->stream>>>(an.n, h->Qf.dev(), h->AT.dev(), flag);
      int *hs = (int *)h->hpin;
      HIPCHK(hipMemcpyAsync(hs, flag, sizeof(int), hipMemcpyDeviceToHost, h->stream));
      HIPCHK(hipStreamSynchronize(h->stream));
      h->zd_weak = hs[0] != 0;
      if (h->zd_weak) {  // what the switch to policy 0 will need: the values on the host
        auto keep = [&](std::vector<double> &dst, const double *src, size_t k) -> int {
          dst.resize(k);
          if (!k) return 0;
          if (h->opts.loc == HQPKKT_LOC_DEVICE)
            HIPCHK(hipMemcpy(dst.data(), src, sizeof(double) * k, hipMemcpyDeviceToHost));
          else
            std::memcpy(dst.data(), src, sizeof(double) * k);
          return 0;
        };
        if ((e = keep(h->hQ, Qx, an.nq)) || (e = keep(h->hA, Ax, an.na)) || (e = keep(h->hC, Cx, an.nc))) return e;
      }
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    h->have_values = true;
    h->factored = false;
    return 0;
  });
}

// zd_policy -1, weak Hessian diagonals, a solve whose refinement did not reach mat_eps: from now
// on every multiplier right behind a matched neighbour (policy 0).  Symbolic phase, upload and
// values again (hqpkkt_mehrotra's vectors stay); the caller factorises and solves once more.
static int switch_to_policy0(hqpkkt_t *h) {
  if (getenv("HQPKKT_TRACE_SOLVE")) fprintf(stderr, "solve: refinement failed with weak Hessian diagonals: zero-diagonal placement 2 -> 0, factorising again\n");
  HIPCHK(hipSetDevice(h->opts.device));
  HIPCHK(hipStreamSynchronize(h->stream));
  const bool lazy = h->lazy, hot = h->ip_hot_valid;
  const int loc = h->opts.loc;
  h->release_device(true);
  const int n = h->an.n, me = h->an.me, m = h->an.m;
  int e = run_analysis(h, n, me, m, 0);
  if (e) return e;
  h->opts.loc = HQPKKT_LOC_HOST;  // the kept values are host copies
  h->lazy = false;
  e = hqpkkt_set_values(h, h->hQ.data(), h->hA.data(), h->hC.data());
  h->opts.loc = loc, h->lazy = lazy, h->ip_hot_valid = hot;
  return e;
}

// inside the device-resident loops (lazy) of the tree engine on one GPU: no staging copies, the sequences are captured
// on the caller's device vectors (hqpkkt_t::DirectGraph)
static bool direct_vectors(const hqpkkt_t *h) {
  return getenv("HQPKKT_NO_DIRECT_VECTORS") == nullptr && (h->lazy || h->direct_now) && h->opts.loc == HQPKKT_LOC_DEVICE &&
         h->opts.mode != HQPKKT_MODE_STAGED && h->an.shard_count <= 1 && h->use_graphs;
}
// A caller's own factor / solve calls (not the device-resident loops): direct from the SECOND call in a row with the same
// set of pointers (a caller that passes fresh vectors every time would pay a graph capture per call), and only if no
// two of the vectors overlap (the staging copies read every input before any output is written; the sequences do not).
struct DirectCall {
  hqpkkt_t *h;
  DirectCall(hqpkkt_t *h_, const void *const *ptr, const int *len, int count, const void **last) : h(h_) {
    bool same = !h->lazy && h->an.m > 0;
    for (int i = 0; i < count; i++) same = same && ptr[i] == last[i] && (ptr[i] != nullptr || len[i] == 0);
    for (int i = 0; i < count; i++) last[i] = ptr[i];
    for (int i = 0; same && i < count; i++)
      for (int j = i + 1; j < count; j++) {
        const char *a = (const char *)ptr[i], *b = (const char *)ptr[j];
        if (len[i] > 0 && len[j] > 0 && a < b + sizeof(double) * (size_t)len[j] && b < a + sizeof(double) * (size_t)len[i]) same = false;
      }
    h->direct_now = same;
  }
  ~DirectCall() { h->direct_now = false; }
};
// the vectors a solve works on: the caller's (direct_vectors) or the staging buffers, filled
static int solve_vecs(hqpkkt_t *h, const double *z, const double *w, const double *r1, const double *r2, const double *r3,
                      const double *r4, double *dx, double *dy, double *dz, double *dw, Vecs &v) {
  if (direct_vectors(h)) {
    v.z = z, v.w = w, v.r1 = r1, v.r2 = r2, v.r3 = r3, v.r4 = r4, v.dx = dx, v.dy = dy, v.dz = dz, v.dw = dw;
    return 0;
  }
  int e = stage_in(h, z, w, r1, r2, r3, r4, v);
  if (e) return e;
  stage_out_ptrs(h, v);
  return 0;
}

static int factor_once(hqpkkt_t *h, const double *z, const double *w);
// (a polled launch that gave up has switched the handle to the per-level launches: once more, then)
int hqpkkt_factor(hqpkkt_t *h, const double *z, const double *w) {
  int e = factor_once(h, z, w);
  if (e == HQPKKT_E_POLL) e = factor_once(h, z, w);
  return e == HQPKKT_E_POLL ? HQPKKT_E_DEVICE : e;
}
static int factor_once(hqpkkt_t *h, const double *z, const double *w) {
  if (!h) return HQPKKT_E_NULL;
  if (!h->analyzed || !h->have_values) return HQPKKT_E_INTERN;
  if (h->an.m > 0 && (!z || !w)) return HQPKKT_E_NULL;
  HIPCHK(hipSetDevice(h->opts.device));
  const void *const fp[2] = {z, w};
  const int fl[2] = {h->an.m, h->an.m};
  DirectCall direct_call(h, fp, fl, 2, h->last_f);
  Vecs v{};
  int e = 0;
  // a caller's own call on its DEVICE vectors (from the second one in a row with the same pointers: DirectCall): the
  // sequence and the posted status words as one graph as well, no timing events in the chain
  const bool devg = !h->lazy && direct_vectors(h) && h->use_graphs && !h->prof.on && getenv("HQPKKT_NO_HOST_GRAPHS") == nullptr;
  const bool hostg = devg || (host_graphs_ok(h) && !direct_vectors(h));
  const auto wall0 = std::chrono::steady_clock::now();
  if (devg) {
    v.z = z, v.w = w;
    h->factored = false;
    const void *key[10] = {z, w, (const void *)(intptr_t)1};
    if ((e = graphed(h, h->direct_slot(h->gdirect_call, key), [&]() {
           const int e2 = do_factor(h, v);
           return e2 ? e2 : post_words(h, nullptr, 0);
         })))
      return e;
  } else if (hostg) {
    // packed by the CPU, then ONE graph: the copy out of the pinned buffer, the factorisation, the posted status words
    const size_t used = stage_pack(h, z, w, nullptr, nullptr, nullptr, nullptr);
    const int m = h->an.m;
    v.z = h->vin.p, v.w = h->vin.p + m;
    h->factored = false;
    if ((e = graphed(h, h->ghost_factor, [&]() {
           if (used) {
             CopyList L{{h->hstage_dev, nullptr, nullptr, nullptr, nullptr, nullptr}, {h->vin.p, nullptr, nullptr, nullptr, nullptr, nullptr}, {(int)used, 0, 0, 0, 0, 0}};
             k_copy_vectors<<<copy_blocks(L), 256, 0, h->stream>>>(L, 1);
           }
           const int e2 = do_factor(h, v);
           return e2 ? e2 : post_words(h, nullptr, 0);
         })))
      return e;
  } else {
  if (direct_vectors(h))
    v.z = z, v.w = w;
  else if ((e = stage_in(h, z, w, nullptr, nullptr, nullptr, nullptr, v)))
    return e;
  h->factored = false;
  // (inside the device-resident loops - lazy - nobody reads these times, and an event record between two graph launches
  // is a packet the queue stops at: 22 us between the factorisation and the first solve of an iteration with three of
  // them, profiles/r06_ip_did_timeline.txt)
  if (!h->lazy) HIPCHK(hipEventRecord(h->ev0, h->stream));
  if ((e = do_factor(h, v))) return e;
  if (h->use_graphs && !h->prof.on && !h->lazy) {  // phases are not timed separately inside a graph replay
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    HIPCHK(hipEventRecord(h->evs1, h->stream));
  }
  if (h->lazy) {
    h->factor_unchecked = true, h->factored = true;
    return 0;
  }
  }
  int *hs = (int *)h->hpin;
  {  // the status words, posted (k_post_words) and waited for
    int ep = hostg ? 0 : post_words(h, nullptr, 0);
    if (ep || (ep = post_wait(h))) return ep;
  }
  if (poll_fallback(h, hs)) return HQPKKT_E_POLL;  // (factored stays false)
  const int flags[4] = {hs[0], hs[1], hs[2], hs[3]};
  std::memcpy(&h->st.kmax, hs + 120, sizeof(double));
  h->prof.collect();
  if (hostg) {  // (no events in the chain: the call's wall time, copies and read-back included)
    h->st.ms_assemble = 0.f;
    h->st.ms_factor = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - wall0).count();
  } else if (h->use_graphs && !h->prof.on) {
    h->st.ms_assemble = 0.f;  // inside the replayed graph
    h->st.ms_factor = elapsed(h->ev0, h->evs1);
  } else {
    h->st.ms_assemble = elapsed(h->ev0, h->ev1);
    h->st.ms_factor = elapsed(h->ev1, h->evs1);
  }
  h->st.n_2x2 = flags[1], h->st.n_perturbed = flags[2], h->st.n_slow_pivots = flags[3];
  h->soft_singular = hs[4] != 0;
  h->soft_tiny = hs[5] != 0;
  if (getenv("HQPKKT_TRACE_SOLVE") && (flags[0] || hs[4] || hs[5]))
    fprintf(stderr, "factor: status %d, 2x2 %d, perturbed %d, zero pivot perturbed %d, tiny multiplier pivot %d, kmax %.3e\n", flags[0],
            flags[1], flags[2], hs[4], hs[5], h->st.kmax);
  if (flags[0]) return flags[0];
  if (!(h->st.kmax == h->st.kmax) || std::isinf(h->st.kmax)) return HQPKKT_E_SING;
  h->factored = true;
  return 0;
}

static int step_once(hqpkkt_t *h, const double *z, const double *w, const double *r1, const double *r2, const double *r3,
                     const double *r4, double *dx, double *dy, double *dz, double *dw);
int hqpkkt_step(hqpkkt_t *h, const double *z, const double *w, const double *r1,
                const double *r2, const double *r3, const double *r4, double *dx, double *dy,
                double *dz, double *dw) {
  int e = step_once(h, z, w, r1, r2, r3, r4, dx, dy, dz, dw);
  if (e == HQPKKT_E_POLL) e = step_once(h, z, w, r1, r2, r3, r4, dx, dy, dz, dw);
  return e == HQPKKT_E_POLL ? HQPKKT_E_DEVICE : e;
}
static int step_once(hqpkkt_t *h, const double *z, const double *w, const double *r1, const double *r2, const double *r3,
                     const double *r4, double *dx, double *dy, double *dz, double *dw) {
  if (!h) return HQPKKT_E_NULL;
  if (!h->factored) return HQPKKT_E_INTERN;
  HIPCHK(hipSetDevice(h->opts.device));
  Vecs v{};
  int e = stage_in(h, z, w, r1, r2, r3, r4, v);
  if (e) return e;
  stage_out_ptrs(h, v);
  HIPCHK(hipEventRecord(h->evs0, h->stream));
  if ((e = do_step(h, v, 0))) return e;
  HIPCHK(hipEventRecord(h->evs1, h->stream));
  if ((e = stage_out(h, v, dx, dy, dz, dw))) return e;
  int *hs = (int *)h->hpin;  // the give-up word of the polled sweeps comes back with the result
  HIPCHK(hipMemcpyAsync(hs + XW_GAVE_UP, h->flags.p + XW_GAVE_UP, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (poll_fallback(h, hs)) return HQPKKT_E_POLL;
  unstage(h, dx, dy, dz, dw);
  h->prof.collect();
  h->st.ms_step = elapsed(h->evs0, h->evs1);
  return 0;
}

int hqpkkt_residual(hqpkkt_t *h, const double *z, const double *w, const double *r1,
                    const double *r2, const double *r3, const double *r4, const double *dx,
                    const double *dy, const double *dz, const double *dw, double *res) {
  if (!h || !res) return HQPKKT_E_NULL;
  if (!h->have_values) return HQPKKT_E_INTERN;
  HIPCHK(hipSetDevice(h->opts.device));
  const int n = h->an.n, me = h->an.me, m = h->an.m;
  Vecs v{};
  int e = stage_in(h, z, w, r1, r2, r3, r4, v);
  if (e) return e;
  stage_out_ptrs(h, v);
  if (h->opts.loc == HQPKKT_LOC_DEVICE) {
    CopyList L{{dx, dy, dz, dw, nullptr, nullptr}, {v.dx, v.dy, v.dz, v.dw, nullptr, nullptr}, {n, me, m, m, 0, 0}};
    k_copy_vectors<<<copy_blocks(L), 256, 0, h->stream>>>(L, 4);
  } else {
    if (n) HIPCHK(hipMemcpyAsync(v.dx, dx, sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
    if (me) HIPCHK(hipMemcpyAsync(v.dy, dy, sizeof(double) * me, hipMemcpyHostToDevice, h->stream));
    if (m) HIPCHK(hipMemcpyAsync(v.dz, dz, sizeof(double) * m, hipMemcpyHostToDevice, h->stream));
    if (m) HIPCHK(hipMemcpyAsync(v.dw, dw, sizeof(double) * m, hipMemcpyHostToDevice, h->stream));
  }
  HIPCHK(hipEventRecord(h->evs0, h->stream));
  e = run_residual(h, v, res);
  if (e == HQPKKT_E_POLL) e = HQPKKT_E_DEVICE;  // (a stale word: the residual kernels poll nothing)
  HIPCHK(hipEventRecord(h->evs1, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->prof.collect();
  h->st.ms_residual = elapsed(h->evs0, h->evs1);
  return e;
}

static int solve_tail(hqpkkt_t *h, Vecs &v, const double *z, const double *w, const double *r1, const double *r2, const double *r3,
                      const double *r4, double *dx, double *dy, double *dz, double *dw, double res, double *res_out);
// Hqp_IpMatrix::solve (hqp/Hqp_IpMatrix.C:65-128)
static int solve_once(hqpkkt_t *h, const double *z, const double *w, const double *r1, const double *r2, const double *r3,
                      const double *r4, double *dx, double *dy, double *dz, double *dw, double *res_out);
int hqpkkt_solve(hqpkkt_t *h, const double *z, const double *w, const double *r1,
                 const double *r2, const double *r3, const double *r4, double *dx, double *dy,
                 double *dz, double *dw, double *res_out) {
  int e = solve_once(h, z, w, r1, r2, r3, r4, dx, dy, dz, dw, res_out);
  // a polled sweep gave up: the factors are those of the per-level kernels as well (bit-identical), the solve runs again
  // on the per-level launches.  (Inside the device-resident loops - lazy - the loop itself is run again.)
  if (e == HQPKKT_E_POLL && h->factored && !h->lazy) e = solve_once(h, z, w, r1, r2, r3, r4, dx, dy, dz, dw, res_out);
  return e == HQPKKT_E_POLL && !h->lazy ? HQPKKT_E_DEVICE : e;
}
static int solve_once(hqpkkt_t *h, const double *z, const double *w, const double *r1, const double *r2, const double *r3,
                      const double *r4, double *dx, double *dy, double *dz, double *dw, double *res_out) {
  if (!h) return HQPKKT_E_NULL;
  if (!h->factored) return HQPKKT_E_INTERN;
  HIPCHK(hipSetDevice(h->opts.device));
  hipStream_t s = h->stream;
  const void *const sp[10] = {z, w, r1, r2, r3, r4, dx, dy, dz, dw};
  const int sl[10] = {h->an.m, h->an.m, h->an.n, h->an.me, h->an.m, h->an.m, h->an.n, h->an.me, h->an.m, h->an.m};
  DirectCall direct_call(h, sp, sl, 10, h->last_s);
  Vecs v{};
  int e = 0;
  if (!h->lazy && direct_vectors(h) && h->use_graphs && !h->prof.on && getenv("HQPKKT_NO_HOST_GRAPHS") == nullptr) {
    // a caller's own solve on its DEVICE vectors: sweeps, residual and the posted words as one graph
    const auto wall0 = std::chrono::steady_clock::now();
    if ((e = solve_vecs(h, z, w, r1, r2, r3, r4, dx, dy, dz, dw, v))) return e;
    const void *key[10] = {z, w, r1, r2, r3, r4, dx, dy, dz, (const void *)((const char *)dw + 1)};  // (+ 1: not the key of the sequence's own graph)
    if ((e = graphed(h, h->direct_slot(h->gdirect_call, key), [&]() {
           int e2 = do_step(h, v, 0);
           if (e2 || (e2 = residual_launch(h, v))) return e2;
           return post_words(h, nullptr, 0, true);
         })))
      return e;
    double res = 0.0;
    if ((e = post_wait(h)) || (e = collect_residual(h, &res))) return e;
    h->host_graph_call = true;
    e = solve_tail(h, v, z, w, r1, r2, r3, r4, dx, dy, dz, dw, res, res_out);
    h->host_graph_call = false;
    h->st.ms_solve = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    return e;
  }
  {
    bool all = host_graphs_ok(h) && !direct_vectors(h) && h->hstage_out;
    for (int k = 0; k < 10; k++) all = all && (sp[k] != nullptr || sl[k] == 0);
    if (all) {
      // packed by the CPU, then ONE graph: the copy out of the pinned buffer, the sweeps, the residual, the result into
      // the pinned buffer, the posted words (residual and status); the CPU hands the result out when they have arrived
      const auto wall0 = std::chrono::steady_clock::now();
      const size_t used = stage_pack(h, z, w, r1, r2, r3, r4);
      const int n = h->an.n, me = h->an.me, m = h->an.m;
      double *b = h->vin.p;
      v.z = b, v.w = b + m, v.r1 = b + 2 * (size_t)m, v.r2 = v.r1 + n, v.r3 = v.r2 + me, v.r4 = v.r3 + m;
      stage_out_ptrs(h, v);
      if ((e = graphed(h, h->ghost_step, [&]() {
             if (used) {
               CopyList L{{h->hstage_dev, nullptr, nullptr, nullptr, nullptr, nullptr}, {b, nullptr, nullptr, nullptr, nullptr, nullptr}, {(int)used, 0, 0, 0, 0, 0}};
               k_copy_vectors<<<copy_blocks(L), 256, 0, s>>>(L, 1);
             }
             int e2 = do_step(h, v, 0);
             if (e2 || (e2 = residual_launch(h, v))) return e2;
             CopyList O{{v.dx, nullptr, nullptr, nullptr, nullptr, nullptr}, {h->hstage_dev + h->hstage_in, nullptr, nullptr, nullptr, nullptr, nullptr}, {(int)h->hstage_out, 0, 0, 0, 0, 0}};
             k_copy_vectors<<<copy_blocks(O), 256, 0, s>>>(O, 1);
             return post_words(h, nullptr, 0, true);
           })))
        return e;
      h->out_pending = h->hstage + h->hstage_in, h->out_by_kernel = true;
      double res = 0.0;
      if ((e = post_wait(h)) || (e = collect_residual(h, &res))) return e;
      h->host_graph_call = true;
      e = solve_tail(h, v, z, w, r1, r2, r3, r4, dx, dy, dz, dw, res, res_out);
      h->host_graph_call = false;
      h->st.ms_solve = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - wall0).count();
      return e;
    }
  }
  if ((e = solve_vecs(h, z, w, r1, r2, r3, r4, dx, dy, dz, dw, v))) return e;
  if (!h->lazy) HIPCHK(hipEventRecord(h->ev0, s));
  if ((e = do_step(h, v, 0))) return e;
  double res = 0.0;
  const OutPtrs outp{dx, dy, dz, dw};
  if ((e = run_residual(h, v, &res, (h->lazy || v.dx == dx) ? nullptr : &outp))) return e;
  if (h->residual_pending) {  // (hqpkkt_franke: solve_tail follows if the residual, once read, asks for it)
    if (res_out) *res_out = 0.0;
    return v.dx == dx ? 0 : stage_out(h, v, dx, dy, dz, dw);
  }
  return solve_tail(h, v, z, w, r1, r2, r3, r4, dx, dy, dz, dw, res, res_out);
}

// refinement and the checks of hqpkkt_solve, from the residual `res` of the first solution on
static int solve_tail(hqpkkt_t *h, Vecs &v, const double *z, const double *w, const double *r1, const double *r2, const double *r3,
                      const double *r4, double *dx, double *dy, double *dz, double *dw, double res, double *res_out) {
  const int n = h->an.n, me = h->an.me, m = h->an.m;
  hipStream_t s = h->stream;
  int e;
  double res_last;
  const double target = h->refine_target > 0.0 ? std::fmin(h->opts.eps, h->refine_target) : h->opts.eps;
  const bool refined = res > target;  // otherwise the caller's copy is already complete
  const double res_first = res;
  double res_acc = res;  // residual of the solution as it stands (res is the last TRIAL's when that one was rejected, as in the reference)
  double res_acc_prev = HUGE_VAL;  // ... and before the last round
  // correction solve: rhs = residual vectors, result = vcor
  Vecs c = v;
  c.r1 = h->vres.p, c.r2 = c.r1 + n, c.r3 = c.r2 + me, c.r4 = c.r3 + m;
  c.dx = h->vcor.p, c.dy = c.dx + n, c.dz = c.dy + me, c.dw = c.dz + m;
  const int ntot = n + me + 2 * m;
  int rounds = 0;
  // (the reference makes at most five rounds, hqp/Hqp_IpMatrix.C:84: its factors come from a pivot search over the whole
  // remaining column.  Where THIS factorisation had to perturb a pivot, or met a multiplier pivot at rounding level -
  // the pivot search is confined to the supernode's block - the factors are those of a nearby matrix and the rounds
  // contract more slowly: up to fifteen then, as long as they still gain, so that solve() returns what the caller's
  // optimality test expects of an accurate factorisation - Hqp_IpsFranke compares the returned residual with its
  // eps, hqp/Hqp_IpsFranke.C:372)
  // (not behind a cancelled pivot that was USED as it was - soft_tiny alone: five rounds leave such factors at a residual
  // that says "singular", fifteen can drag a consistent singular system below mat_eps, and the reference reports it)
  // Only inside the device-resident loops (h->lazy): a caller's own hqpkkt_solve - the reference's solvers through the
  // shim - gets the reference's five rounds and with them the residual the reference's plugin contract describes
  // (ADVICE r5).  Rounds beyond five show in hqpkkt_stats.refine_rounds.
  const int max_rounds = (h->st.n_perturbed > 0 && h->lazy) ? 15 : 5;
  for (int it = 0; it < max_rounds && res > target; it++) {
    if (it >= 5 && !(res < 0.5 * res_acc_prev)) break;  // beyond the reference's five: only while a round still halves the residual
    res_last = res;
    if ((e = do_step(h, c, 1))) return e;
    rounds++;
    double alpha = 1.0;
    do {
      KLAUNCH(h, KC_VECTOR, k_axpy4<<<nblk(ntot), 256, 0, s>>>(n, me, m, alpha, c.dx, c.dy, c.dz, c.dw, v.dx, v.dy, v.dz,
                                         v.dw));
      if ((e = run_residual(h, v, &res))) return e;
      if (res > res_last) {
        KLAUNCH(h, KC_VECTOR, k_axpy4<<<nblk(ntot), 256, 0, s>>>(n, me, m, -alpha, c.dx, c.dy, c.dz, c.dw, v.dx, v.dy,
                                           v.dz, v.dw));
        alpha -= 0.3;
      }
    } while (res > res_last && alpha > 0.0);
    if (alpha <= 0.0) break;
    res_acc_prev = res_acc, res_acc = res;
  }
  if (!(res <= h->opts.eps) && h->zd_weak && h->zd_used == 2) {  // (sharded: every rank sees the same residual and switches)
    // the refinement did not reach mat_eps: weak Hessian diagonals and every multiplier behind
    // ALL its neighbours is the placement that loses accuracy when z/w spreads (hqpkkt_opts.
    // zd_policy); switch the handle to the matching rule and do this factor + solve again
    if ((e = switch_to_policy0(h)) || (e = hqpkkt_factor(h, z, w))) return e;
    return hqpkkt_solve(h, z, w, r1, r2, r3, r4, dx, dy, dz, dw, res_out);
  }
  if (!h->lazy && !h->host_graph_call) HIPCHK(hipEventRecord(h->ev1, s));
  if (h->lazy) {
    if (v.dx != dx && (e = stage_out(h, v, dx, dy, dz, dw))) return e;
  } else {
    if (refined && v.dx != dx) {
      if ((e = stage_out(h, v, dx, dy, dz, dw))) return e;
    }
    // (a call whose result a kernel has put into the pinned buffer in front of the posted words, and no refinement: it is there)
    if (!h->host_graph_call || rounds > 0 || refined) HIPCHK(hipStreamSynchronize(s));  // (returns at once when nothing was queued after the read-back)
    unstage(h, dx, dy, dz, dw);
    h->prof.collect();
    if (!h->host_graph_call) h->st.ms_solve = elapsed(h->ev0, h->ev1);
  }
  h->st.refine_rounds = rounds;
  if (getenv("HQPKKT_TRACE_SOLVE")) fprintf(stderr, "solve: first residual %.3e, %d rounds, final %.3e\n", res_first, rounds, res);
  if (res_out) *res_out = res;
  if (res != res) return HQPKKT_E_SING;
  // an exactly zero pivot outside a root front was perturbed: singular if the refinement failed
  if (h->soft_singular && !(res <= h->opts.eps)) return HQPKKT_E_SING;
  // a multiplier-type pivot below 1e-13 max|K| and a solve that ends no nearer to a solution than the zero
  // vector is: the rank-deficient equality block the reference reports as E_SING (hqp/spBKP.C:699-700).
  // (Not: a late interior-point system whose refinement stalls at 1e-4 - the reference goes on there.)
  if (h->soft_tiny && !(res_acc <= 1e-4)) {
    double rnorm = 0.0;
    Vecs vz = v;
    vz.dx = h->vcor.p, vz.dy = vz.dx + n, vz.dz = vz.dy + me, vz.dw = vz.dz + m;
    HIPCHK(hipMemsetAsync(h->vcor.p, 0, sizeof(double) * (size_t)ntot, s));
    if ((e = run_residual(h, vz, &rnorm))) return e;
    if (!(res_acc < rnorm)) return HQPKKT_E_SING;
  }
  return 0;
}

// ---- device-resident Mehrotra predictor-corrector loop ----------------------
// Restatement of hqp/Hqp_IpsMehrotra.C: cold_start (:209-327), step (:355-693),
// solve (:696-735, cold start only).  Scalars are reduced on the device in a fixed
// order and read back; vectors stay on the device.
namespace {
struct IpCtx {
  hqpkkt_t *h;
  int n, me, m;
  double *x, *y, *z, *w, *r1, *r2, *r3, *r4, *dxa, *dya, *dza, *dwa, *dx, *dy, *dz, *dw, *c, *b, *d, *part, *out;
  double *zh, *wh;  // hot-start candidates (hqp/Hqp_IpsMehrotra.C:475-478)
  double *hout;  // pinned (h->hpin + 64)
  int reduce(const int (&ops)[IP_SLOTS], int nout) {
    IpOps o;
    for (int k = 0; k < IP_SLOTS; k++) o.op[k] = ops[k];
    k_ip_final<<<1, 256, 0, h->stream>>>(part, o, out, IpEpi{0, 0, 0.0, 0.0, 0.0, nullptr, nullptr});
    int e = post_words(h, out, nout);  // (hout = hpin + 64: where the posting kernel puts them)
    return e ? e : post_wait(h);
  }
};
}  // namespace

int hqpkkt_default_ip_opts(hqpkkt_ip_opts *o) {
  if (!o) return HQPKKT_E_NULL;
  std::memset(o, 0, sizeof(*o));
  o->eps = 1e-10;       // hqp/Hqp_Solver.C:53
  o->max_iters = 200;   // hqp/Hqp_Solver.C:52
  o->gammaf = 0.01;     // hqp/Hqp_IpsMehrotra.C:95
  return 0;
}

int hqpkkt_mehrotra(hqpkkt_t *h, const hqpkkt_ip_opts *opts, const double *c, const double *b,
                    const double *d, double *x, double *y, double *z, double *w, hqpkkt_ip_result *res) {
  auto loop = [&]() -> int {
    if (!h || !res) return HQPKKT_E_NULL;
    if (!h->analyzed || !h->have_values) return HQPKKT_E_INTERN;
    if (opts && opts->max_iters < 0) return HQPKKT_E_RANGE;
      if (h->an.shard_count > 1) return HQPKKT_E_INTERN;
    hqpkkt_ip_opts o;
    if (opts)
      o = *opts;
    else
      hqpkkt_default_ip_opts(&o);
    Analysis &an = h->an;
    const int n = an.n, me = an.me, m = an.m;
    if ((n && !c) || (me && !b) || (m && !d) || (n && !x) || (me && !y) || (m && (!z || !w))) return HQPKKT_E_NULL;
    HIPCHK(hipSetDevice(h->opts.device));
    hipStream_t s = h->stream;
    const size_t nv = (size_t)n + me + 2 * (size_t)m;
    const size_t need = 4 * nv + (size_t)n + me + m + (size_t)IP_BLOCKS * IP_SLOTS + 64 + 2 * (size_t)m;
    int e;
    if (h->ipv.count < need) {
      if ((e = h->ipv.alloc(need))) return e;
      h->ip_hot_valid = false;
    }
    h->fr_hot_valid = false;  // the arena is shared with hqpkkt_franke
    IpCtx C;
    C.h = h, C.n = n, C.me = me, C.m = m, C.hout = h->hpin + 64;
    double *q = h->ipv.p;
    auto take = [&](size_t k) { double *r = q; q += k; return r; };
    C.x = take(n), C.y = take(me), C.z = take(m), C.w = take(m);
    C.r1 = take(n), C.r2 = take(me), C.r3 = take(m), C.r4 = take(m);
    C.dxa = take(n), C.dya = take(me), C.dza = take(m), C.dwa = take(m);
    C.dx = take(n), C.dy = take(me), C.dz = take(m), C.dw = take(m);
    C.c = take(n), C.b = take(me), C.d = take(m);
    C.part = take((size_t)IP_BLOCKS * IP_SLOTS), C.out = take(64);
    C.zh = take(m), C.wh = take(m);
    // out: 0..7 reductions (k_ip_final), 16..27 the blocking components (k_ip_minratio_final),
    // 32..39 the step's scalars (IPS_*)
    double *const Bk = C.out + 16, *const S = C.out + 32;
    const hipMemcpyKind in_kind = h->opts.loc == HQPKKT_LOC_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    const hipMemcpyKind out_kind = h->opts.loc == HQPKKT_LOC_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (n) HIPCHK(hipMemcpyAsync(C.c, c, sizeof(double) * n, in_kind, s));
    if (me) HIPCHK(hipMemcpyAsync(C.b, b, sizeof(double) * me, in_kind, s));
    if (m) HIPCHK(hipMemcpyAsync(C.d, d, sizeof(double) * m, in_kind, s));
    // the plugin entry points below take the driver's DEVICE vectors
    const int saved_loc = h->opts.loc;
    struct Restore {
      hqpkkt_t *h;
      int loc;
      ~Restore() {
        h->opts.loc = loc, h->lazy = false, h->factor_unchecked = false;
        (void)hipMemsetAsync(h->flags.p + TINY_REPLACE_WORD, 0, sizeof(int), h->stream);  // (kernels.hip.h: cancelled pivots are replaced inside the loop only)
      }
    } restore{h, saved_loc};
    h->opts.loc = HQPKKT_LOC_DEVICE;
    h->lazy = true;  // no host round trip where the loop does not need the answer at once
    // STAGED with dense dynamics: their share of A x and A'y for the right-hand sides (k_ip_rhs)
    const double *dx1 = nullptr, *dx2 = nullptr;
    int dndyn = 0;
    auto dyn_products = [&]() -> int {
      if (h->opts.mode != HQPKKT_MODE_STAGED) return 0;
      Vecs vv{};
      vv.dx = C.x, vv.dy = C.y;
      return staged_dense_products(h, vv, &dx1, &dx2, &dndyn);
    };
    hipEvent_t t0 = h->ev0;  // total time: own pair of events (the plugin calls reuse the handle's)
    const hipEvent_t tb = h->evt0, te = h->evt1;  // owned by the handle: no early return can leak them
    (void)t0;
    HIPCHK(hipEventRecord(tb, s));
    std::memset(res, 0, sizeof(*res));
    res->result = 2;  // Hqp_Infeasible until decided (hqp/Hqp_IpsMehrotra.C:219)
    const int total = n + me + m;
    double resid = 0.0;
    int iter = 0, n_factor = 0, n_solve = 0;
    auto finish = [&](int result) -> int {
      res->result = result, res->iters = iter, res->n_factor = n_factor, res->n_solve = n_solve;
      if (n) HIPCHK(hipMemcpyAsync(x, C.x, sizeof(double) * n, out_kind, s));
      if (me) HIPCHK(hipMemcpyAsync(y, C.y, sizeof(double) * me, out_kind, s));
      if (m) HIPCHK(hipMemcpyAsync(z, C.z, sizeof(double) * m, out_kind, s));
      if (m) HIPCHK(hipMemcpyAsync(w, C.w, sizeof(double) * m, out_kind, s));
      HIPCHK(hipEventRecord(te, s));
      HIPCHK(hipStreamSynchronize(s));
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, tb, te);
      res->ms_total = ms;
      return 0;
    };
    auto factor = [&]() -> int { n_factor++; return hqpkkt_factor(h, C.z, C.w); };
    auto solve = [&](double *ox, double *oy, double *oz, double *ow) -> int {
      n_solve++;
      return hqpkkt_solve(h, C.z, C.w, C.r1, C.r2, C.r3, C.r4, ox, oy, oz, ow, &resid);
    };
    const int OPS_NONE[IP_SLOTS] = {IP_SUM, IP_SUM, IP_SUM, IP_SUM, IP_SUM, IP_SUM, IP_SUM, IP_SUM};
    // small QPs: an iteration's vector work between its solves in one workgroup each (ipdriver.hip.h, k_ip_pred_small)
    const bool ip_small = !getenv("HQPKKT_NO_IP_SMALL") && m > 0 && m <= IP_SMALL_M && (long long)n + me + m <= 4 * IP_SMALL_M;
  
    // ------------------------------------------------------------ iterations
    std::vector<double> phimin((size_t)o.max_iters + 2, 0.0);
    double mu0 = 0.0, norm_r0 = 0.0, norm_data = 1.0;
    // hot start (hqp/Hqp_IpsMehrotra.C:330-352, 475-478, 696-733): x, y of the last solve and
    // the (z, w) kept from its last iteration far enough from the solution; a hot start that
    // does not reduce phi by 1.2 per iteration, takes a step below 1e-5, runs max_warm_iters or
    // does not end optimal is thrown away and the QP solved again from a cold start
    const bool keep_hot = o.hot_start != 0 && m > 0;  // 1: hot start if possible, 2: cold, but prepare the next
    bool hot = o.hot_start == 1 && m > 0 && h->ip_hot_valid;
    const int max_warm = o.max_warm_iters > 0 ? o.max_warm_iters : 25;
    const double hot_thresh = std::pow(o.eps, 0.3333);
    int fail_iters = 0;
    double test1 = 0.0;
    const double gamma = std::pow(1.0e-4, 0.25);
    int result = 2;
    bool sing_hot = false;  // E_SING inside a hot-started run: restart cold like any failed hot start
    bool stepped = false, pending = false;  // pending: a step is in the stream whose scalars were not read yet
    double mu_pending = 0.0;
    // The rare second corrector (hqp/Hqp_IpsMehrotra.C:612-624: the first corrector's own
    // step is tiny): safe sigma, then Mehrotra's step rule with the host in the loop.
    auto second_corrector = [&](double mu) -> int {
      int e2;
      const double smm = gamma / (1.0 - gamma) * mu;
      k_ip_corr_rhs<<<nblk(m), 256, 0, s>>>(m, C.z, C.w, C.dza, C.dwa, smm, nullptr, C.r4);
      if ((e2 = solve(C.dx, C.dy, C.dz, C.dw))) return e2;
      k_ip_minratio_part<<<IP_BLOCKS, 256, 0, s>>>(m, C.z, C.w, C.dz, C.dw, C.part);
      k_ip_minratio_final<<<1, 256, 0, s>>>(C.part, C.z, C.w, C.dz, C.dw, Bk, m, gamma, nullptr);
      HIPCHK(hipMemcpyAsync(C.hout, Bk, sizeof(double) * 12, hipMemcpyDeviceToHost, s));
      HIPCHK(hipStreamSynchronize(s));
      const double zmin = C.hout[0], wmin = C.hout[6];
      const int izmin = (int)C.hout[1], iwmin = (int)C.hout[7];
      const double z_iz = C.hout[2], dz_iz = C.hout[3], w_iz = C.hout[4], dw_iz = C.hout[5];
      const double z_iw = C.hout[8], dz_iw = C.hout[9], w_iw = C.hout[10], dw_iw = C.hout[11];
      double alpha;
      if (izmin < 0 && iwmin < 0)
        alpha = 1.0;
      else {
        alpha = izmin < 0 ? wmin : iwmin < 0 ? zmin : std::fmin(zmin, wmin);
        k_ip_mupl<<<IP_BLOCKS, 256, 0, s>>>(m, alpha, nullptr, C.z, C.w, C.dz, C.dw, C.part);
        if ((e2 = C.reduce(OPS_NONE, 1))) return e2;
        const double mu_pl = C.hout[0] / m;
        double fpd;
        if (iwmin >= 0 && alpha == wmin && z_iw > -alpha * dz_iw)
          fpd = (o.gammaf * mu_pl / (z_iw + alpha * dz_iw) - w_iw) / (alpha * dw_iw);
        else if (izmin >= 0 && alpha == zmin && w_iz > -alpha * dw_iz)
          fpd = (o.gammaf * mu_pl / (w_iz + alpha * dw_iz) - z_iz) / (alpha * dz_iz);
        else
          fpd = 0.0;
        alpha = std::fmax(0.0, std::fmin(std::fmax(1.0 - o.gammaf, fpd) * alpha, 1.0));
      }
      res->alpha = alpha;
      k_ip_update<<<IP_BLOCKS, 256, 0, s>>>(n, me, m, alpha, nullptr, C.x, C.y, C.z, C.w, C.dx, C.dy, C.dz, C.dw, C.part);
      return 0;
    };
    // before leaving the loop with a step still in the stream: was it taken?
    auto settle = [&]() -> int {
      if (!pending) return 0;
      pending = false;
      HIPCHK(hipMemcpyAsync(C.hout + 32, S, sizeof(double) * 8, hipMemcpyDeviceToHost, s));
      HIPCHK(hipStreamSynchronize(s));
      res->alpha = C.hout[32 + IPS_ALPHA];
      if (C.hout[32 + IPS_NEED2] != 0.0) return second_corrector(mu_pending);
      return 0;
    };
    // The iteration's launches between two read-backs as ONE captured graph each (small QPs on the tree engine: an
    // iteration of the double-integrator QP is 27 launches and 0.3 ms, and every boundary between a graph and the next
    // launch costs the queue 5 - 20 us, profiles/r06_ip_did_timeline.txt):
    //   A: factorisation + predictor solve + its residual + the posting kernel
    //   B: predictor statistics + corrector solve + residual + post
    //   C: the step + the next iterate's right-hand sides and reductions + post
    // Whatever the read-back then asks for - refinement rounds, the second corrector - runs as before, launch by launch.
    const bool seg_ok = ip_small && h->use_graphs && !h->prof.on && h->opts.mode != HQPKKT_MODE_STAGED && h->an.shard_count <= 1 &&
                        !getenv("HQPKKT_NO_IP_SEGMENTS");
    auto seg_slot = [&](int tag) -> hqpkkt::GraphSlot & {
      unsigned long long gf;
      std::memcpy(&gf, &o.gammaf, sizeof(gf));
      const void *key[10] = {(const void *)(intptr_t)tag, (const void *)(uintptr_t)gf, C.x, C.z, C.r1, C.dx, C.dxa, C.out, nullptr, nullptr};
      return h->direct_slot(h->gdirect_seg, key);
    };
    // a solve whose first residual is in the stream (run_residual with defer_residual): wait, read, and finish it as
    // hqpkkt_solve does (refinement, the checks behind a perturbed pivot)
    auto finish_solve = [&](double *ox, double *oy, double *oz, double *ow) -> int {
      int e2 = post_wait(h);
      if (e2) return e2;
      if ((e2 = collect_residual(h, &resid))) return e2;
      Vecs v{};
      if ((e2 = solve_vecs(h, C.z, C.w, C.r1, C.r2, C.r3, C.r4, ox, oy, oz, ow, v))) return e2;
      return solve_tail(h, v, C.z, C.w, C.r1, C.r2, C.r3, C.r4, ox, oy, oz, ow, resid, &resid);
    };
    auto enqueue_head = [&]() -> int {
      if (h->short_rows)
        k_ip_rhs<4><<<IP_BLOCKS, 256, 0, s>>>(n, me, m, h->Qf.dev(), h->AT.dev(), h->CT.dev(), h->A.dev(), h->C.dev(),
                                              h->vals.p, C.c, C.b, C.d, C.x, C.y, C.z, C.w, C.r1, C.r2, C.r3, C.r4,
                                              C.part, dx1, dx2, dndyn);
      else
        k_ip_rhs<16><<<IP_BLOCKS, 256, 0, s>>>(n, me, m, h->Qf.dev(), h->AT.dev(), h->CT.dev(), h->A.dev(), h->C.dev(),
                                               h->vals.p, C.c, C.b, C.d, C.x, C.y, C.z, C.w, C.r1, C.r2, C.r3, C.r4,
                                               C.part, dx1, dx2, dndyn);
      if (m == 0) return 0;
      // the reductions of this iterate and what the step before left behind, one round trip
      const int ops2[IP_SLOTS] = {IP_SUM, IP_SUM, IP_SUM, IP_MAX, IP_MIN, IP_MIN, IP_SUM, IP_SUM};
      IpOps o2;
      for (int k = 0; k < IP_SLOTS; k++) o2.op[k] = ops2[k];
      k_ip_final<<<1, 256, 0, s>>>(C.part, o2, C.out, IpEpi{0, 0, 0.0, 0.0, 0.0, nullptr, nullptr});
      return post_words(h, C.out, 40);  // (C.hout = hpin + 64: where the posting kernel puts them)
    };
    bool head_in_stream = false;  // segment C of the iteration before has queued this iterate's head already
    for (;;) {  // hot first (if asked for and possible), cold after a failed hot start
    iter = 0, result = 2, stepped = false, pending = false, sing_hot = false, head_in_stream = false;
    std::fill(phimin.begin(), phimin.end(), 0.0);
    res->alpha = 1.0;
    if (hot) {
      CopyList L{{C.zh, C.wh, nullptr, nullptr, nullptr, nullptr}, {C.z, C.w, nullptr, nullptr, nullptr, nullptr}, {m, m, 0, 0, 0, 0}};
      k_copy_vectors<<<copy_blocks(L), 256, 0, s>>>(L, 2);
    } else {
        // (x = y = 0 until the cold start's solve has succeeded: what the caller gets back when the
        // very first factorisation is singular, as from the reference)
        if (n) HIPCHK(hipMemsetAsync(C.x, 0, sizeof(double) * n, s));
        if (me) HIPCHK(hipMemsetAsync(C.y, 0, sizeof(double) * me, s));
        if (m > 0) {
      // qp_init_method (:226-250, 294-297): 0 z = w = 1, r4 = 0; 1, 2 w = a ratio of the data's norms;
      // 3 as 0 with r4 = -z.*w and the solve's dz, dw added to z, w
      double w0 = 1.0;
      if (o.init_method == 1) w0 = std::fmax(o.norm_d, 1e-10) * o.norm_Q / o.norm_C;
      if (o.init_method == 2) w0 = o.norm_C / std::fmax(o.norm_d, 1e-10) / o.norm_Q;
      k_ip_cold_rhs<<<nblk(total), 256, 0, s>>>(n, me, m, C.c, C.b, C.d, C.z, C.w, C.r1, C.r2, C.r3, C.r4, w0,
                                                o.init_method ? -w0 : 0.0);
          if ((e = factor()) || (e = solve(C.dx, C.dy, C.dz, C.dw))) {
            if (e == HQPKKT_E_SING) return finish(4);  // Hqp_Degenerate (:262-269)
            return e;
          }
          HIPCHK(hipMemcpyAsync(C.x, C.dx, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
          if (me) HIPCHK(hipMemcpyAsync(C.y, C.dy, sizeof(double) * me, hipMemcpyDeviceToDevice, s));
      if (o.init_method == 3) k_ip_shift<<<nblk(m), 256, 0, s>>>(m, C.dz, C.dw, 1.0, 1.0, C.dz, C.dw);  // :294-297
      k_ip_cold_stats<<<IP_BLOCKS, 256, 0, s>>>(m, C.dz, C.dw, C.part);
          const int ops1[IP_SLOTS] = {IP_MIN, IP_MIN, IP_MAX, IP_MAX, IP_SUM, IP_SUM, IP_SUM, IP_SUM};
          if ((e = C.reduce(ops1, 6))) return e;
          double mindz = C.hout[0], mindw = C.hout[1], sumdz = C.hout[4], sumdw = C.hout[5];
          if (C.hout[2] == 0.0) {  // :301-304
            k_ip_fill<<<nblk(m), 256, 0, s>>>(m, 1.0e-10, C.dz);
            mindz = 1.0e-10, sumdz = 1.0e-10 * m;
          }
          if (C.hout[3] == 0.0) {
            k_ip_fill<<<nblk(m), 256, 0, s>>>(m, 1.0e-10, C.dw);
            mindw = 1.0e-10, sumdw = 1.0e-10 * m;
          }
          double delz = std::fmax(-1.5 * mindz, 0.0), delw = std::fmax(-1.5 * mindw, 0.0);
          // gap = (dz + delz)'(dw + delw): k_ip_mupl with alpha = 1 on (delz, dz), (delw, dw) shifted vectors
          k_ip_shift<<<nblk(m), 256, 0, s>>>(m, C.dz, C.dw, delz, delw, C.z, C.w);
          k_ip_mupl<<<IP_BLOCKS, 256, 0, s>>>(m, 0.0, nullptr, C.z, C.w, C.dz, C.dw, C.part);
          if ((e = C.reduce(OPS_NONE, 1))) return e;
          const double gap0 = C.hout[0];
          delz += 0.5 * gap0 / (sumdw + m * delw);
          delw += 0.5 * gap0 / (sumdz + m * delz);
          k_ip_shift<<<nblk(m), 256, 0, s>>>(m, C.dz, C.dw, delz, delw, C.z, C.w);
        }
  
      if (keep_hot) {  // :318-319
        k_ip_fill<<<nblk(m), 256, 0, s>>>(m, 1.0, C.zh);
        k_ip_fill<<<nblk(m), 256, 0, s>>>(m, 1.0, C.wh);
      }
    }
    // the cold start's factorisation has succeeded (or a hot start carries on): the matrix is regular, cancelled multiplier
    // pivots are replaced from here on (kernels.hip.h, TINY_REPLACE_WORD)
    // (2: exactly zero pivots as well - only where the factorisation just checked met no cancelled multiplier pivot: kernels.hip.h)
    if (h->tiny_replace_in_loop) HIPCHK(hipMemsetAsync(h->flags.p + TINY_REPLACE_WORD, (!hot && !h->soft_tiny) ? 2 : 1, sizeof(int), s));
    bool restart_cold = false;
    while (true) {
      double phi = 0.0;
      bool redo = false;  // the second corrector replaced the step: same step() call, new right-hand sides
      do {
      // ---- one step (hqp/Hqp_IpsMehrotra.C:355-693)
      if (!head_in_stream) {
        if ((e = dyn_products()) || (e = enqueue_head())) return e;
      }
      head_in_stream = false;
      if (m == 0) {  // equality-constrained QP: one Newton step (:364-413)
        if ((e = factor()) || (e = solve(C.dx, C.dy, C.dz, C.dw))) {
          if (e == HQPKKT_E_SING) return finish(4);
          return e;
        }
        k_ip_update<<<IP_BLOCKS, 256, 0, s>>>(n, me, m, 1.0, nullptr, C.x, C.y, C.z, C.w, C.dx, C.dy, C.dz, C.dw, C.part);
        iter++;
        return finish(0);
      }
      if ((e = post_wait(h))) return e;
      if (pending) {
        pending = false;
        res->alpha = C.hout[32 + IPS_ALPHA];
        if (C.hout[32 + IPS_NEED2] != 0.0) {  // that step was not taken (alpha 0): second corrector first
          iter--;
          if ((e = second_corrector(mu_pending))) {
              if (e == HQPKKT_E_SING && hot) {
              sing_hot = true;
              break;
            }
            if (e == HQPKKT_E_SING) return finish(4);
            return e;
          }
          iter++;
          redo = true;  // right-hand sides and reductions of the new iterate
          break;
        }
      }
      const double gap = C.hout[0], mu = C.hout[2] / m, norm_r = C.hout[3];
      if (stepped && (!std::isfinite(mu) || !std::isfinite(norm_r) || !std::isfinite(gap))) {
        iter--;  // the reference leaves the failed step uncounted
        result = 4;
        break;
      }
      res->gap = gap, res->mu = mu, res->pcost = C.hout[1];
      if (iter == 0) {
        mu0 = mu, norm_r0 = norm_r;
        norm_data = o.norm_data > 0.0 ? o.norm_data : 1.0;
      }
      phi = (norm_r + std::fabs(gap)) / norm_data;
      phimin[iter] = phi;
      res->phi = phi;
      if (keep_hot && phi > hot_thresh) {  // prepare the next hot start (:475-478)
        CopyList L{{C.z, C.w, nullptr, nullptr, nullptr, nullptr}, {C.zh, C.wh, nullptr, nullptr, nullptr, nullptr}, {m, m, 0, 0, 0, 0}};
        k_copy_vectors<<<copy_blocks(L), 256, 0, s>>>(L, 2);
      }
      if (mu <= o.eps && norm_r <= o.eps * norm_data) {  // :487-490
        result = 0;
        break;
      }
      double pm = phimin[0];
      for (int i = 1; i <= iter; i++) pm = std::fmin(pm, phimin[i]);
      if (phi > o.eps && phi >= 1.0e4 * pm) {  // :494-502
        result = 3;
        break;
      }
      if (iter >= 30) {  // slow convergence (:506-516)
        double pm30 = phimin[1];
        for (int i = 2; i <= iter - 30; i++) pm30 = std::fmin(pm30, phimin[i]);
        if (pm >= 0.5 * pm30) {
          result = 3;
          break;
        }
      }
      if (norm_r > o.eps * norm_data && norm_r / mu >= 1.0e8 * norm_r0 / mu0) result = 3;  // :520-524 (no return)
      // factorise; predictor (affine) step
      if (seg_ok) {
        h->defer_residual = true;
        e = graphed(h, seg_slot(1), [&]() {
          const int e2 = hqpkkt_factor(h, C.z, C.w);
          return e2 ? e2 : hqpkkt_solve(h, C.z, C.w, C.r1, C.r2, C.r3, C.r4, C.dxa, C.dya, C.dza, C.dwa, &resid);
        });
        h->defer_residual = false;
        n_factor++, n_solve++;
        if (!e) {
          h->factor_unchecked = true, h->factored = true, h->residual_pending = true;  // (what the two calls leave, replayed or not)
          e = finish_solve(C.dxa, C.dya, C.dza, C.dwa);
        }
      } else if (!(e = factor()))
        e = solve(C.dxa, C.dya, C.dza, C.dwa);
      if (e) {
        if (e == HQPKKT_E_SING && hot) {  // a hot start that ends degenerate is thrown away (:723-727)
          sing_hot = true;
          break;
        }
        if (e == HQPKKT_E_SING) return finish(4);
        return e;
      }
      // From here to the step itself nothing is read back: sigma (Terlaky's modification,
      // :583-590; the safe value when the predictor step is short and the reference skips the
      // first corrector, :612-616), the corrector's blocking components, the damped step length
      // (:629-672) are computed by thread 0 of the reduction kernels and consumed through device pointers.
      if (seg_ok) {
        h->defer_residual = true;
        e = graphed(h, seg_slot(2), [&]() {
          k_ip_pred_small<<<1, 1024, 0, s>>>(m, C.z, C.w, C.dza, C.dwa, C.out + 2, gamma, S, C.r4);
          return hqpkkt_solve(h, C.z, C.w, C.r1, C.r2, C.r3, C.r4, C.dx, C.dy, C.dz, C.dw, &resid);
        });
        h->defer_residual = false;
        n_solve++;
        if (!e) {
          h->residual_pending = true;
          e = finish_solve(C.dx, C.dy, C.dz, C.dw);
        }
      } else {
      if (ip_small) {  // one workgroup: the three launches below, same arithmetic (ipdriver.hip.h)
        k_ip_pred_small<<<1, 1024, 0, s>>>(m, C.z, C.w, C.dza, C.dwa, C.out + 2, gamma, S, C.r4);
      } else {
      k_ip_ratio<<<IP_BLOCKS, 256, 0, s>>>(m, C.z, C.w, C.dza, C.dwa, C.part);
      {
        const int ops3[IP_SLOTS] = {IP_MIN, IP_MAX, IP_SUM, IP_SUM, IP_SUM, IP_SUM, IP_SUM, IP_SUM};
        IpOps o3;
        for (int k = 0; k < IP_SLOTS; k++) o3.op[k] = ops3[k];
        k_ip_final<<<1, 256, 0, s>>>(C.part, o3, C.out, IpEpi{1, m, mu, gamma, 0.0, nullptr, S});
      }
      k_ip_corr_rhs<<<nblk(m), 256, 0, s>>>(m, C.z, C.w, C.dza, C.dwa, 0.0, S + IPS_SMM, C.r4);
      }
      e = solve(C.dx, C.dy, C.dz, C.dw);
      }
      if (e) {
        if (e == HQPKKT_E_SING && hot) {
          sing_hot = true;
          break;
        }
        if (e == HQPKKT_E_SING) return finish(4);
        return e;
      }
      if (seg_ok) {  // the step and the head of the next pass through this loop
        if ((e = graphed(h, seg_slot(3), [&]() {
               k_ip_step_small<<<1, 1024, 0, s>>>(n, me, m, C.x, C.y, C.z, C.w, C.dx, C.dy, C.dz, C.dw, Bk, gamma, o.gammaf, S);
               return enqueue_head();
             })))
          return e;
        head_in_stream = true;
      } else if (ip_small) {  // one workgroup: the five launches below, same arithmetic
        k_ip_step_small<<<1, 1024, 0, s>>>(n, me, m, C.x, C.y, C.z, C.w, C.dx, C.dy, C.dz, C.dw, Bk, gamma, o.gammaf, S);
      } else {
      k_ip_minratio_part<<<IP_BLOCKS, 256, 0, s>>>(m, C.z, C.w, C.dz, C.dw, C.part);
      k_ip_minratio_final<<<1, 256, 0, s>>>(C.part, C.z, C.w, C.dz, C.dw, Bk, m, gamma, S);
      k_ip_mupl<<<IP_BLOCKS, 256, 0, s>>>(m, 0.0, S + IPS_ALPHA_PRE, C.z, C.w, C.dz, C.dw, C.part);
      {
        IpOps on;
        for (int k = 0; k < IP_SLOTS; k++) on.op[k] = IP_SUM;
        k_ip_final<<<1, 256, 0, s>>>(C.part, on, C.out, IpEpi{2, m, 0.0, 0.0, o.gammaf, Bk, S});
      }
      k_ip_update<<<IP_BLOCKS, 256, 0, s>>>(n, me, m, 0.0, S + IPS_ALPHA, C.x, C.y, C.z, C.w, C.dx, C.dy, C.dz, C.dw,
                                            C.part);
      }
      // (:684-690: a non-finite mu or x ends the solve as degenerate; seen here by the next
      // pass through k_ip_rhs, whose sums and maximum carry the NaN / inf)
      iter++;
      stepped = true, pending = true, mu_pending = mu;
      } while (0);
      if (sing_hot) {
        result = 4;
        break;
      }
      if (redo) continue;
      // ---- what solve() does after every step() call (:703-718)
      const bool leave = result == 0 || result == 3 || result == 4 || iter + fail_iters >= o.max_iters ||
                         (hot && iter >= max_warm);
      if (hot || leave) {  // the step's own scalars are needed now: was it taken, how long was it
        if ((e = settle())) {
          if (e == HQPKKT_E_SING && hot) {
            result = 4;
            break;
          }
          if (e == HQPKKT_E_SING) return finish(4);
          return e;
        }
      }
      if (hot) {
        if (iter == 1)
          test1 = phi;
        else if (phi > test1 / std::pow(1.2, iter - 1.0) || res->alpha < 1.0e-5) {
          fail_iters += iter;
          restart_cold = true;
          break;
        }
      }
      if (leave) break;
    }
    if (restart_cold || (hot && result != 0)) {  // bad hot start: its iterations are lost (:723-727)
      if (!restart_cold) fail_iters += iter;
      hot = false;
      continue;
    }
    break;
    }
    iter += fail_iters;
    if (m > 0) h->ip_hot_valid = keep_hot;
    return finish(result);
  };
  return ip_attempts(h, opts, res, loop);
}

// ---- device-resident Franke loop ----------------------------------------------
// Restatement of hqp/Hqp_IpsFranke.C: cold_start (:156-216), step (:271-378), solve
// (:381-416, cold start only).  One factor + one solve per iteration; the scalars (mu from
// the gap and rhomin, the step length, zeta) live on the host as in the reference.
int hqpkkt_franke(hqpkkt_t *h, const hqpkkt_ip_opts *opts, const double *c, const double *b,
                  const double *d, double *x, double *y, double *z, double *w, hqpkkt_ip_result *res) {
  auto loop = [&]() -> int {
    if (!h || !res) return HQPKKT_E_NULL;
    if (!h->analyzed || !h->have_values) return HQPKKT_E_INTERN;
    if (opts && opts->max_iters < 0) return HQPKKT_E_RANGE;
      if (h->an.shard_count > 1) return HQPKKT_E_INTERN;
    hqpkkt_ip_opts o;
    if (opts)
      o = *opts;
    else
      hqpkkt_default_ip_opts(&o);
    Analysis &an = h->an;
    const int n = an.n, me = an.me, m = an.m;
    if ((n && !c) || (me && !b) || (m && !d) || (n && !x) || (me && !y) || (m && (!z || !w))) return HQPKKT_E_NULL;
    HIPCHK(hipSetDevice(h->opts.device));
    hipStream_t s = h->stream;
    const size_t nv = (size_t)n + me + 2 * (size_t)m;
    // same arena as hqpkkt_mehrotra (its hot-start data does not survive this call)
    const size_t need = 5 * nv + (size_t)n + me + m + (size_t)IP_BLOCKS * IP_SLOTS + 64 + 2 * (size_t)m;  // (+ nv: the iterate before a step)
    int e;
    if (h->ipv.count < need) {
      if ((e = h->ipv.alloc(need))) return e;
      h->fr_hot_valid = false;
    }
    h->ip_hot_valid = false;
    IpCtx C;
    C.h = h, C.n = n, C.me = me, C.m = m, C.hout = h->hpin + 64;
    double *q = h->ipv.p;
    auto take = [&](size_t k) { double *r = q; q += k; return r; };
    C.x = take(n), C.y = take(me), C.z = take(m), C.w = take(m);
    C.r1 = take(n), C.r2 = take(me), C.r3 = take(m), C.r4 = take(m);
    double *a1 = take(n), *a2 = take(me), *a3 = take(m);
    (void)take(m);
    C.dx = take(n), C.dy = take(me), C.dz = take(m), C.dw = take(m);
    C.c = take(n), C.b = take(me), C.d = take(m);
    C.part = take((size_t)IP_BLOCKS * IP_SLOTS), C.out = take(64);
    (void)take(2 * (size_t)m);
    double *const keep = take(nv);  // x, y, z, w before the step of an iteration (see below)
    const hipMemcpyKind in_kind = h->opts.loc == HQPKKT_LOC_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    const hipMemcpyKind out_kind = h->opts.loc == HQPKKT_LOC_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (n) HIPCHK(hipMemcpyAsync(C.c, c, sizeof(double) * n, in_kind, s));
    if (me) HIPCHK(hipMemcpyAsync(C.b, b, sizeof(double) * me, in_kind, s));
    if (m) HIPCHK(hipMemcpyAsync(C.d, d, sizeof(double) * m, in_kind, s));
    const int saved_loc = h->opts.loc;
    struct Restore {
      hqpkkt_t *h;
      int loc;
      ~Restore() {
        h->opts.loc = loc, h->lazy = false, h->factor_unchecked = false, h->defer_residual = false, h->residual_pending = false;
        (void)hipMemsetAsync(h->flags.p + TINY_REPLACE_WORD, 0, sizeof(int), h->stream);
      }
    } restore{h, saved_loc};
    h->opts.loc = HQPKKT_LOC_DEVICE;
    h->lazy = true;
    // STAGED with dense dynamics: their share of A x and A'y for the right-hand sides (k_ip_rhs)
    const double *dx1 = nullptr, *dx2 = nullptr;
    int dndyn = 0;
    auto dyn_products = [&]() -> int {
      if (h->opts.mode != HQPKKT_MODE_STAGED) return 0;
      Vecs vv{};
      vv.dx = C.x, vv.dy = C.y;
      return staged_dense_products(h, vv, &dx1, &dx2, &dndyn);
    };
    const hipEvent_t tb = h->evt0, te = h->evt1;  // owned by the handle: no early return can leak them
    HIPCHK(hipEventRecord(tb, s));
    std::memset(res, 0, sizeof(*res));
    res->result = 2;
    int iter = 0, n_factor = 0, n_solve = 0;
    auto finish = [&](int result) -> int {
      res->result = result, res->iters = iter, res->n_factor = n_factor, res->n_solve = n_solve;
      if (n) HIPCHK(hipMemcpyAsync(x, C.x, sizeof(double) * n, out_kind, s));
      if (me) HIPCHK(hipMemcpyAsync(y, C.y, sizeof(double) * me, out_kind, s));
      if (m) HIPCHK(hipMemcpyAsync(z, C.z, sizeof(double) * m, out_kind, s));
      if (m) HIPCHK(hipMemcpyAsync(w, C.w, sizeof(double) * m, out_kind, s));
      HIPCHK(hipEventRecord(te, s));
      HIPCHK(hipStreamSynchronize(s));
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, tb, te);
      res->ms_total = ms;
      return 0;
    };
    const int total = n + me + m;
    const double beta = 0.995;  // qp_beta (:77)
    const int max_warm = o.max_warm_iters > 0 ? o.max_warm_iters : 15;  // qp_max_warm_iters (:81)
    bool hot = o.hot_start == 1 && m > 0 && h->fr_hot_valid;
    int fail_iters = 0, result = 2;
    double rhomin = 0.0, Ltilde = 0.0, zeta = 1.0, gap = 0.0, alpha = 1.0, alphabar = 1.0, gap1 = 0.0;
    const int OPS_SUM[IP_SLOTS] = {IP_SUM, IP_SUM, IP_SUM, IP_SUM, IP_SUM, IP_SUM, IP_SUM, IP_SUM};
    for (;;) {  // hot first (if asked for and possible), cold after a failed hot start (:381-416)
    iter = 0, alpha = 1.0, zeta = 1.0, result = 2;
    if (hot) {
      // hot_start (:222-266): x, y, z, w of the last solve, w += 1e-10, the slack vectors a1..a3
      // of that point - which are the right-hand sides r1..r3 of Mehrotra's loop
      k_ip_shift<<<nblk(m), 256, 0, s>>>(m, C.z, C.w, 0.0, 1e-10, C.z, C.w);
      if ((e = dyn_products())) return e;
      if (h->short_rows)
        k_ip_rhs<4><<<IP_BLOCKS, 256, 0, s>>>(n, me, m, h->Qf.dev(), h->AT.dev(), h->CT.dev(), h->A.dev(), h->C.dev(),
                                              h->vals.p, C.c, C.b, C.d, C.x, C.y, C.z, C.w, a1, a2, a3, C.r4, C.part, dx1, dx2, dndyn);
      else
        k_ip_rhs<16><<<IP_BLOCKS, 256, 0, s>>>(n, me, m, h->Qf.dev(), h->AT.dev(), h->CT.dev(), h->A.dev(), h->C.dev(),
                                               h->vals.p, C.c, C.b, C.d, C.x, C.y, C.z, C.w, a1, a2, a3, C.r4, C.part, dx1, dx2, dndyn);
      if ((e = C.reduce(OPS_SUM, 3))) return e;
      gap = C.hout[2] + 1.0;  // in_prod(z, w) + 1 (:248)
      if (rhomin == 0.0) rhomin = h->fr_rhomin;
    } else {
    // ---- cold start (:156-216)
    if (m > 0) {
      rhomin = 1000.0 * m;
      k_fr_dstats<<<IP_BLOCKS, 256, 0, s>>>(m, C.d, C.part);
      const int opsd[IP_SLOTS] = {IP_MIN, IP_MAX, IP_SUM, IP_SUM, IP_SUM, IP_SUM, IP_SUM, IP_SUM};
      if ((e = C.reduce(opsd, 3))) return e;
      const double min_d = C.hout[0], norm_d = C.hout[1];
      if (o.qp_mu0 > 0.0) {  // "choose Ltilde according _mu0" (:167-173)
        const double mean_d_h = 0.5 * C.hout[2] / (double)m;
        Ltilde = -mean_d_h + std::sqrt(mean_d_h * mean_d_h + (double)m * rhomin * o.qp_mu0);
        Ltilde = std::fmax(Ltilde, -min_d);
      } else {  // "according Wright" (:175-182)
        Ltilde = std::fmax(norm_d, -min_d);
        Ltilde = std::fmax(Ltilde, 1e2 * m);
      }
    }
    if (h->short_rows)
      k_fr_cold<4><<<IP_BLOCKS, 256, 0, s>>>(n, me, m, h->CT.dev(), Ltilde, C.c, C.b, C.d, C.x, C.y, C.z, C.w, a1, a2, a3, C.part);
    else
      k_fr_cold<16><<<IP_BLOCKS, 256, 0, s>>>(n, me, m, h->CT.dev(), Ltilde, C.c, C.b, C.d, C.x, C.y, C.z, C.w, a1, a2, a3, C.part);
    gap = 0.0;
    if (m > 0) {
      if ((e = C.reduce(OPS_SUM, 1))) return e;
      gap = C.hout[0];
    }
    }
    bool restart_cold = false;
    // ---- iterations (:381-416 around :271-378)
    while (true) {
      if (iter == 0) alphabar = 1.0;
      if (iter == 1 && h->tiny_replace_in_loop)
        HIPCHK(hipMemsetAsync(h->flags.p + TINY_REPLACE_WORD, h->soft_tiny ? 1 : 2, sizeof(int), s));  // (the first factorisation + solve has succeeded; 2: exact zeros too, kernels.hip.h)
      double mu;
      if (1.0 / gap < rhomin || alpha < 1.0) {
        mu = alphabar * gap / rhomin;             // potential reduction
        mu += (1.0 - alphabar) * gap / (double)m;  // centering
      } else
        mu = gap * gap;  // quadratic convergence
      if (m == 0) mu = 0.0;
      h->hpin[HPIN_ZM] = zeta, h->hpin[HPIN_ZM + 1] = mu;
      std::atomic_thread_fence(std::memory_order_release);
      // The whole step - right-hand sides, factorisation, solve, its residual, the step length, the update, the new gap,
      // both posts - as ONE captured graph (the launches take nothing from the host that changes from step to step)
      const bool seg_ok = h->use_graphs && !h->prof.on && h->opts.mode != HQPKKT_MODE_STAGED && h->an.shard_count <= 1 &&
                          !getenv("HQPKKT_NO_IP_SEGMENTS") && !getenv("HQPKKT_FRANKE_TWO_READS");
      if (!seg_ok) k_fr_rhs<<<nblk(total), 256, 0, s>>>(n, me, m, h->hpin_dev + HPIN_ZM, a1, a2, a3, C.z, C.w, C.r1, C.r2, C.r3, C.r4);
      double resid = 0.0;
      n_factor++, n_solve++;
      // The step length below compares dw = C dx - r3 with w, whose active components are of the
      // order gap / m: a residual of mat_eps = 1e-10, which the reference's global pivoting stays
      // far below without refinement, lets that noise block the step near the solution (the loop
      // then creeps on with alpha -> 0).  Ask the solve for a residual below the slacks.
      h->refine_target = m > 0 ? std::fmax(0.05 * gap / (double)m, 2e-12) : 0.0;
      // One read-back per iteration: the solve leaves its first residual (and the status of the factorisation) in
      // the stream, the step length is computed and consumed on the device, and residual, status, step length and
      // the new gap come back together.  When the words then say that the solve was not finished (refinement wanted,
      // a perturbed pivot to judge, an error), the iterate of before the step is put back, the solve is finished as
      // hqpkkt_solve would have, and the step is taken again.
      double *const Sfr = C.out + 32;
      auto take_step_enqueue = [&]() -> int {
        if (m > 0) {
          k_fr_ratio<<<IP_BLOCKS, 256, 0, s>>>(m, C.z, C.w, C.dz, C.dw, C.part);
          IpOps orat;
          orat.op[0] = IP_MIN;
          for (int k = 1; k < IP_SLOTS; k++) orat.op[k] = IP_SUM;
          k_ip_final<<<1, 256, 0, s>>>(C.part, orat, C.out, IpEpi{3, m, 0.0, 0.0, beta, nullptr, Sfr});
        }
        k_fr_update<<<IP_BLOCKS, 256, 0, s>>>(n, me, m, 1.0, m > 0 ? Sfr + IPS_ALPHA : nullptr, C.x, C.y, C.z, C.w, C.dx,
                                              C.dy, C.dz, C.dw, C.part);
        IpOps ou;
        for (int k = 0; k < IP_SLOTS; k++) ou.op[k] = IP_SUM;
        ou.op[1] = IP_MAX;
        k_ip_final<<<1, 256, 0, s>>>(C.part, ou, C.out, IpEpi{0, 0, 0.0, 0.0, 0.0, nullptr, nullptr});
        return post_words(h, C.out, 40);  // (the residual of the solve has gone to the host with the post behind its kernel)
      };
      auto take_step = [&]() -> int {
        const int ep = take_step_enqueue();
        return ep ? ep : post_wait(h);
      };
      const double target = h->refine_target > 0.0 ? std::fmin(h->opts.eps, h->refine_target) : h->opts.eps;  // (as solve_tail)
      const CopyList Lkeep{{C.x, C.y, C.z, C.w, nullptr, nullptr}, {keep, keep + n, keep + n + me, keep + n + me + m, nullptr, nullptr}, {n, me, m, m, 0, 0}};
      if (seg_ok) {
        h->defer_residual = true;
        {
          unsigned long long kb;
          std::memcpy(&kb, &beta, sizeof(kb));
          const void *key[10] = {(const void *)(intptr_t)4, (const void *)(uintptr_t)kb, C.x, C.z, C.r1, C.dx, keep, C.out, a1, nullptr};
          e = graphed(h, h->direct_slot(h->gdirect_seg, key), [&]() {
            k_fr_rhs<<<nblk(total), 256, 0, s>>>(n, me, m, h->hpin_dev + HPIN_ZM, a1, a2, a3, C.z, C.w, C.r1, C.r2, C.r3, C.r4);
            int e2 = hqpkkt_factor(h, C.z, C.w);
            if (!e2) e2 = hqpkkt_solve(h, C.z, C.w, C.r1, C.r2, C.r3, C.r4, C.dx, C.dy, C.dz, C.dw, &resid);
            if (e2) return e2;
            k_copy_vectors<<<copy_blocks(Lkeep), 256, 0, s>>>(Lkeep, 4);
            return take_step_enqueue();
          });
        }
        h->defer_residual = false;
        if (!e) {
          h->factor_unchecked = true, h->factored = true, h->residual_pending = true;  // (what the calls leave, replayed or not)
          e = post_wait(h);
        }
      } else {
        h->defer_residual = !getenv("HQPKKT_FRANKE_TWO_READS");
        e = hqpkkt_factor(h, C.z, C.w);
        if (!e) e = hqpkkt_solve(h, C.z, C.w, C.r1, C.r2, C.r3, C.r4, C.dx, C.dy, C.dz, C.dw, &resid);
        h->defer_residual = false;
        if (!e && h->residual_pending) {
          k_copy_vectors<<<copy_blocks(Lkeep), 256, 0, s>>>(Lkeep, 4);
          if ((e = take_step())) return e;
        }
      }
      if (!e && h->residual_pending) {
        e = collect_residual(h, &resid);
        const bool unfinished = e || !(resid <= target) || h->soft_singular || h->soft_tiny;
        if (unfinished) {
          CopyList B{{keep, keep + n, keep + n + me, keep + n + me + m, nullptr, nullptr}, {C.x, C.y, C.z, C.w, nullptr, nullptr}, {n, me, m, m, 0, 0}};
          k_copy_vectors<<<copy_blocks(B), 256, 0, s>>>(B, 4);
          if (!e) {
            Vecs v{};
            if ((e = solve_vecs(h, C.z, C.w, C.r1, C.r2, C.r3, C.r4, C.dx, C.dy, C.dz, C.dw, v))) return e;
            e = solve_tail(h, v, C.z, C.w, C.r1, C.r2, C.r3, C.r4, C.dx, C.dy, C.dz, C.dw, resid, &resid);
          }
          if (!e && (e = take_step())) return e;
        }
      } else if (!e) {
        if ((e = take_step())) return e;
      }
      h->refine_target = 0.0;
      if (e == HQPKKT_E_SING && hot) {  // Hqp_Degenerate inside a hot start: thrown away (:405-411)
        result = 4;
        break;
      }
      if (e) {
        if (e == HQPKKT_E_SING) return finish(4);  // Hqp_Degenerate (:308-310)
        return e;
      }
      alpha = m > 0 ? C.hout[32 + IPS_ALPHA] : std::fmin(1.0, 2.0 * beta);
      alphabar = 0.5 * alphabar + 0.5 * alpha;
      if (alphabar == 1.0)
        rhomin *= 2.0;
      else if (alphabar < 0.5 && rhomin > 100.0 * m)
        rhomin /= 2.0;
      zeta *= (1.0 - alpha);
      gap = m > 0 ? C.hout[0] : 0.0;
      res->gap = gap, res->alpha = alpha, res->mu = mu, res->phi = zeta;
      {
        static const bool trace_ip = getenv("HQPKKT_TRACE_IP") != nullptr;  // (diagnosis: the loop's scalars after every step)
        if (trace_ip)
          fprintf(stderr, "franke: step %d gap %.17g alpha %.17g alphabar %.17g zeta %.17g rhomin %.17g resid %.3e mu %.6e hot %d\n", iter + 1, gap, alpha,
                  alphabar, zeta, rhomin, resid, mu, hot ? 1 : 0);
      }
      if (!std::isfinite(gap) || !std::isfinite(C.hout[1])) {  // :351-354
        result = 4;
      } else {
        iter++;
        if (!(zeta < o.eps))  // (:361-374, comparisons written to filter out NaN)
          result = alpha < o.eps ? 3 : 2;
        else if (!(gap < o.eps) || !(resid < o.eps))
          result = 1;  // Hqp_Feasible
        else
          result = 0;
      }
      // ---- what solve() does after every step() (:388-403)
      if (hot) {
        if (iter == 1)
          gap1 = gap;
        else if (gap > gap1) {
          fail_iters += iter;
          restart_cold = true;
          break;
        }
      }
      if (iter + fail_iters >= o.max_iters) break;
      if (hot && iter >= max_warm) break;
      if (result == 0 || result == 3 || result == 4) break;
    }
    if (restart_cold || (hot && result != 0)) {  // bad hot start (:405-411)
      if (!restart_cold) fail_iters += iter;
      hot = false;
      continue;
    }
    break;
    }
    iter += fail_iters;
    h->fr_hot_valid = m > 0 && result != 4;
    h->fr_rhomin = rhomin;
    return finish(result);
  };
  return ip_attempts(h, opts, res, loop);
}

int hqpkkt_get_sbw(const hqpkkt_t *h, int *sbw) {
  if (!h || !sbw) return HQPKKT_E_NULL;
  *sbw = h->analyzed ? h->an.sbw : -1;
  return 0;
}

int hqpkkt_get_perm(const hqpkkt_t *h, int *perm) {
  if (!h || !perm) return HQPKKT_E_NULL;
  if (!h->analyzed) return HQPKKT_E_INTERN;
  if ((int)h->an.qp2j.size() != h->an.dim) return HQPKKT_E_INTERN;  // STAGED: stage order, no RCM
  std::memcpy(perm, h->an.qp2j.data(), sizeof(int) * h->an.dim);
  return 0;
}

int hqpkkt_set_tol(hqpkkt_t *h, double tol) {
  if (!h) return HQPKKT_E_NULL;
  if (!(tol > 0.0 && tol <= 1.0)) return HQPKKT_E_RANGE;
  if (tol != h->opts.tol) h->drop_graphs();  // alpha is baked into the captured launches
  h->opts.tol = tol;
  return 0;
}

int hqpkkt_set_eps(hqpkkt_t *h, double eps) {
  if (!h) return HQPKKT_E_NULL;
  h->opts.eps = eps;
  return 0;
}

int hqpkkt_set_stream(hqpkkt_t *h, void *hip_stream) {
  if (!h) return HQPKKT_E_NULL;
  int e = ensure_device(h);
  if (e) return e;
  h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
  h->drop_graphs();
  return 0;
}

int hqpkkt_set_shard(hqpkkt_t *h, int rank, int count, hqpkkt_exchange_fn fn, void *ctx) {
  if (!h) return HQPKKT_E_NULL;
  if (count < 1 || rank < 0 || rank >= count) return HQPKKT_E_RANGE;
  if (count > 1 && !fn) return HQPKKT_E_NULL;
  if (h->analyzed && (rank != h->shard_rank || count != h->shard_count)) return HQPKKT_E_INTERN;
  h->shard_rank = rank, h->shard_count = count;
  h->xchg_fn = fn, h->xchg_sfn = nullptr, h->xchg_ctx = ctx;
  return 0;
}
int hqpkkt_set_shard_stream(hqpkkt_t *h, int rank, int count, hqpkkt_exchange_stream_fn fn, void *ctx) {
  if (!h) return HQPKKT_E_NULL;
  if (count < 1 || rank < 0 || rank >= count) return HQPKKT_E_RANGE;
  if (count > 1 && !fn) return HQPKKT_E_NULL;
  if (h->analyzed && (rank != h->shard_rank || count != h->shard_count)) return HQPKKT_E_INTERN;
  h->shard_rank = rank, h->shard_count = count;
  h->xchg_fn = nullptr, h->xchg_sfn = fn, h->xchg_ctx = ctx;
  return 0;
}

int hqpkkt_values_staging(hqpkkt_t *h, double **Qx, double **Ax, double **Cx) {
  if (!h || !Qx || !Ax || !Cx) return HQPKKT_E_NULL;
  if (!h->analyzed) return HQPKKT_E_INTERN;
  int e = ensure_device(h);
  if (e) return e;
  const Analysis &an = h->an;
  const size_t need = (size_t)an.nq + an.na + an.nc + 1;
  if (h->hvals && h->hvals_elems != need) (void)hipHostFree(h->hvals), h->hvals = nullptr;  // (analysed again for another pattern)
  if (!h->hvals) {
    HIPCHK(hipHostMalloc((void **)&h->hvals, sizeof(double) * need, hipHostMallocDefault));
    h->hvals_elems = need;
  }
  *Qx = h->hvals, *Ax = h->hvals + an.nq, *Cx = h->hvals + an.nq + an.na;
  return 0;
}

int hqpkkt_set_stages(hqpkkt_t *h, int K, const int *nx, const int *nu) {
  return guarded([&]() -> int {
    if (!h) return HQPKKT_E_NULL;
    if (h->opts.mode != HQPKKT_MODE_STAGED) return HQPKKT_E_INTERN;
    if (!h->sd) h->sd = new (std::nothrow) StagedDev;
    if (!h->sd) return HQPKKT_E_MEM;
    kktdev::StagedPlan &P = h->sd->plan;
    P.given_nx.clear(), P.given_nu.clear();
    if (K <= 0) return 0;  // back to detection from the staircase of A
    if (!nx || !nu) return HQPKKT_E_NULL;
    for (int k = 0; k <= K; k++)
      if (nx[k] < 1) return HQPKKT_E_RANGE;
    for (int k = 0; k < K; k++)
      if (nu[k] < 0) return HQPKKT_E_RANGE;
    P.given_nx.assign(nx, nx + K + 1), P.given_nu.assign(nu, nu + K);
    return 0;
  });
}

int hqpkkt_analyze_staged(hqpkkt_t *h, int K, const int *nx, const int *nu, int n_total, int me_rest, int m, const int *Qp,
                          const int *Qi, const int *Ep, const int *Ei, const int *Cp, const int *Ci) {
  return guarded([&]() -> int {
    if (!h) return HQPKKT_E_NULL;
    if (h->opts.mode != HQPKKT_MODE_STAGED) return HQPKKT_E_INTERN;
    int e = hqpkkt_set_stages(h, K, nx, nu);
    if (e) return e;
    if (K < 1) return HQPKKT_E_RANGE;
    long long n = nx[K], ndyn = 0;
    for (int k = 0; k < K; k++) n += (long long)nx[k] + nu[k], ndyn += nx[k + 1];
    if (n > 0x7fffffffLL || ndyn + me_rest > 0x7fffffffLL || me_rest < 0 || m < 0) return HQPKKT_E_RANGE;
    if (n != n_total) return HQPKKT_E_SIZES;  // Q, E, C were built for another number of variables
    if ((n > 0 && (!Qp || (Qp[n] > 0 && !Qi))) || (me_rest > 0 && (!Ep || (Ep[me_rest] > 0 && !Ei))) ||
        (m > 0 && (!Cp || (Cp[m] > 0 && !Ci))))
      return HQPKKT_E_NULL;
    if (h->uploaded) {
      (void)hipSetDevice(h->opts.device);
      (void)hipStreamSynchronize(h->stream);
      h->release_device();
    }
    h->analyzed = false;
    h->ip_hot_valid = h->fr_hot_valid = false;
    const int me = (int)ndyn + me_rest;
    h->pQp.assign(Qp, Qp + n + 1), h->pQi.assign(Qi, Qi + Qp[n]);
    h->pAp.assign((size_t)me + 1, 0);  // the dynamics rows are empty: they come as dense blocks
    for (int i = 0; i <= me_rest; i++) h->pAp[ndyn + i] = me_rest ? Ep[i] : 0;
    h->pAi.clear();
    if (me_rest && Ep[me_rest]) h->pAi.assign(Ei, Ei + Ep[me_rest]);
    h->pCp.clear(), h->pCi.clear();
    if (m) h->pCp.assign(Cp, Cp + m + 1), h->pCi.assign(Ci, Ci + Cp[m]);
    h->zd_decided = true, h->zd_weak = false;
    return staged_analyze(h, (int)n, me, m, true);
  });
}

int hqpkkt_set_values_staged(hqpkkt_t *h, const double *Qx, const double *const *F, const long long *ldF,
                             const double *Ex, const double *Cx) {
  return guarded([&]() -> int {
    if (!h) return HQPKKT_E_NULL;
    if (!h->analyzed || h->opts.mode != HQPKKT_MODE_STAGED || !h->sd) return HQPKKT_E_INTERN;
    Analysis &an = h->an;
    if ((an.nq && !Qx) || (an.na && !Ex) || (an.nc && !Cx) || (F && !ldF)) return HQPKKT_E_NULL;
    if (!h->sd->plan.dense_dyn) return HQPKKT_E_INTERN;  // analysed for the CSR hand-over
    if (!F) {  // the blocks came one by one (hqpkkt_set_stage_block): every one of them, since the analysis
      const std::vector<char> &bs = h->sd->blocks_set;
      if ((int)bs.size() != h->sd->plan.K || std::find(bs.begin(), bs.end(), 0) != bs.end()) return HQPKKT_E_INTERN;
    }
    return staged_set_values(h, Qx, Ex, Cx, F, ldF, true);
  });
}

int hqpkkt_detect_stages(int n, int rows, const int *row_len, const int *last_col, const int *prev_col, int cap, int *K,
                         int *nx, int *nu, int *dyn_rows) {
  return guarded([&]() -> int {
    if (!row_len || !last_col || !prev_col || !K || !nx || !nu || !dyn_rows) return HQPKKT_E_NULL;
    if (n < 1 || rows < 1) return HQPKKT_E_FORMAT;
    std::vector<int> st, ct, fc;
    int nd = 0;
    if (kktdev::stages_from_staircase(n, rows, row_len, last_col, prev_col, st, ct, fc, nd)) return HQPKKT_E_FORMAT;
    const int k = (int)ct.size();
    if (k > cap) return HQPKKT_E_SIZES;
    *K = k, *dyn_rows = nd;
    for (int i = 0; i <= k; i++) nx[i] = st[i];
    for (int i = 0; i < k; i++) nu[i] = ct[i];
    return 0;
  });
}

int hqpkkt_stage_staging(hqpkkt_t *h, int which, double **buf, long long *elems) {
  return guarded([&]() -> int {
    if (!h || !buf || !elems || which < 0 || which > 1) return HQPKKT_E_NULL;
    if (!h->analyzed || h->opts.mode != HQPKKT_MODE_STAGED || !h->sd || !h->sd->plan.dense_dyn) return HQPKKT_E_INTERN;
    int e = ensure_device(h);
    if (e) return e;
    StagedDev &d = *h->sd;
    const kktdev::StagedPlan &P = d.plan;
    long long mx = 1;
    for (int k = 0; k < P.K; k++) mx = std::max(mx, (long long)P.nk[k + 1] * (P.nk[k] + P.mk[k]));
    for (int b = 0; b < 2; b++)
      if (!d.hblk[b] || d.hblk_elems < mx) {
        if (d.hblk[b]) (void)hipHostFree(d.hblk[b]), d.hblk[b] = nullptr;
        HIPCHK(hipHostMalloc((void **)&d.hblk[b], sizeof(double) * (size_t)mx, hipHostMallocDefault));
      }
    d.hblk_elems = mx;
    // the copy that last read this buffer must be over before the caller refills it
    if (d.hblk_ev[which]) HIPCHK(hipEventSynchronize(d.hblk_ev[which]));
    *buf = d.hblk[which], *elems = mx;
    return 0;
  });
}

int hqpkkt_set_stage_block(hqpkkt_t *h, int k, const double *F, long long ldF) {
  return guarded([&]() -> int {
    if (!h || !F) return HQPKKT_E_NULL;
    if (!h->analyzed || h->opts.mode != HQPKKT_MODE_STAGED || !h->sd || !h->sd->plan.dense_dyn) return HQPKKT_E_INTERN;
    int e;
    if (!h->uploaded && (e = staged_upload(h))) return e;
    StagedDev &d = *h->sd;
    const kktdev::StagedPlan &P = d.plan;
    if (k < 0 || k >= P.K) return HQPKKT_E_RANGE;
    const int nz = P.nk[k] + P.mk[k];
    if (ldF < nz) return HQPKKT_E_SIZES;
    HIPCHK(hipSetDevice(h->opts.device));
    const hipMemcpyKind kind = h->opts.loc == HQPKKT_LOC_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    if ((e = staged_copy_block(h, k, F, ldF, kind))) return e;
    for (int b = 0; b < 2; b++)
      if (F == d.hblk[b]) {  // the library's own staging buffer: remember when it is free again
        if (!d.hblk_ev[b]) HIPCHK(hipEventCreateWithFlags(&d.hblk_ev[b], hipEventDisableTiming));
        HIPCHK(hipEventRecord(d.hblk_ev[b], h->stream));
      }
    if ((int)d.blocks_set.size() != P.K) d.blocks_set.assign(P.K, 0);
    d.blocks_set[k] = 1;
    h->factored = false;
    return 0;
  });
}

// STAGED: rank and number of carried rows of every stage in the last factorisation
// (2 ints per stage, K+1 stages); tests only
int hqpkkt_debug_stage_ranks(hqpkkt_t *h, int *out, int cap) {
  if (!h || !out) return HQPKKT_E_NULL;
  if (!h->sd || !h->uploaded) return HQPKKT_E_INTERN;
  HIPCHK(hipSetDevice(h->opts.device));
  HIPCHK(hipStreamSynchronize(h->stream));
  const kktdev::StagedPlan &P = h->sd->plan;
  for (int k = 0; k <= P.K && 2 * k + 1 < cap; k++)
    HIPCHK(hipMemcpy(out + 2 * k, h->sd->dyn.p + P.dyn_off[k], 2 * sizeof(int), hipMemcpyDeviceToHost));
  return 0;
}

// Micro-benchmark and self-check of the dense fp64 product the STAGED engine is made of
// (k_dgemm_tn): C = A'B (+ lower / mirror) on pseudo-random operands, `reps` timed launches;
// *ms = average device time per launch, *max_err = max |C - exact| over 4096 sampled entries
// relative to sum |a||b|.  Used by tests/ and bench.py (roofline of the kernel on its own).
namespace {
__global__ void k_fill_rand(double *p, long long n, unsigned long long seed) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long x = (unsigned long long)i * 0x9E3779B97F4A7C15ULL + seed;
  x ^= x >> 30, x *= 0xBF58476D1CE4E5B9ULL, x ^= x >> 27, x *= 0x94D049BB133111EBULL, x ^= x >> 31;
  p[i] = (double)(x >> 11) * (1.0 / 9007199254740992.0) - 0.5;
}
__global__ void k_gemm_check(stg::GemmArgs g, int nsample, double *err) {
  const int sidx = blockIdx.x * blockDim.x + threadIdx.x;
  if (sidx >= nsample) return;
  unsigned long long x = (unsigned long long)sidx * 0x9E3779B97F4A7C15ULL + 12345;
  x ^= x >> 29, x *= 0xBF58476D1CE4E5B9ULL, x ^= x >> 32;
  int i = (int)(x % (unsigned long long)g.M), j = (int)((x >> 20) % (unsigned long long)g.N);
  // lower: only i >= j is computed; mirror: C[j][i] is a copy of C[i][j] (the product is
  // symmetric in the engine; here the operands are not, so the copy is what gets checked)
  int ci = i, cj = j;
  if (g.lower && i < j) {
    const int t = i;
    i = j, j = t;
    if (!g.mirror) ci = i, cj = j;
  }
  double s = 0.0, sa = 0.0;
  for (int k = 0; k < g.K; k++) {
    const double a = g.A[(long long)k * g.lda + i], b = g.B[(long long)k * g.ldb + j];
    s += a * b, sa += fabs(a * b);
  }
  const double e = fabs(g.C[(long long)ci * g.ldc + cj] - g.alpha * s) / (sa + 1e-300);
  atomic_max_pos((unsigned long long *)err, e);
}
}  // namespace
int hqpkkt_debug_dgemm(int device, int M, int N, int K, int lower, int mirror, int reps, double *ms, double *max_err) {
  if (M <= 0 || N <= 0 || K < 0 || reps <= 0) return HQPKKT_E_RANGE;
  if (lower && M < N) return HQPKKT_E_RANGE;  // (M > N: the column strip of a lower triangle)
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device) return HQPKKT_E_DEVICE;
  HIPCHK(hipSetDevice(device));
  const long long lda = (M + 7) / 8 * 8, ldb = (N + 7) / 8 * 8, ldc = ldb;
  double *A = nullptr, *B = nullptr, *Cm = nullptr, *err = nullptr, *zr = nullptr;
  stg::SkUnit *sk_table_dev = nullptr;
  auto fin = [&](int rc) {
    (void)hipFree(A), (void)hipFree(B), (void)hipFree(Cm), (void)hipFree(err), (void)hipFree(zr), (void)hipFree(sk_table_dev);
    return rc;
  };
  const size_t kk = K > 0 ? K : 1;
  if (hipMalloc((void **)&A, sizeof(double) * kk * lda) != hipSuccess || hipMalloc((void **)&B, sizeof(double) * kk * ldb) != hipSuccess ||
      hipMalloc((void **)&Cm, sizeof(double) * (size_t)std::max(M, N) * ldc) != hipSuccess || hipMalloc((void **)&err, 8) != hipSuccess)
    return fin(HQPKKT_E_MEM);
  k_fill_rand<<<nblk((long long)kk * lda), 256>>>(A, (long long)kk * lda, 1);
  k_fill_rand<<<nblk((long long)kk * ldb), 256>>>(B, (long long)kk * ldb, 2);
  (void)hipMemset(err, 0, 8);
  (void)hipMemset(Cm, 0, sizeof(double) * (size_t)std::max(M, N) * ldc);
  stg::GemmArgs g{A, lda, B, ldb, nullptr, 0, Cm, ldc, M, N, K, 1.0, 0.0, lower, mirror, nullptr, nullptr};
  const int variant = stg::gemm_variant_from_env();
  if (variant != stg::GEMM_REG4) {
    if (hipMalloc((void **)&zr, sizeof(double) * 256) != hipSuccess) return fin(HQPKKT_E_MEM);
    (void)hipMemset(zr, 0, sizeof(double) * 256);
    g.zeros = zr;
  }
  int cus = 0;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
  const int skg = stg::gemm_wgs_per_cu(variant) * cus;
  // (HQPKKT_DGEMM_FORCE_SPLIT: the cut form whatever the launch rules say - same-box comparisons of the two forms)
  const bool frac = !getenv("HQPKKT_DGEMM_FORCE_SPLIT") && stg::gemm_use_frac(M, N, K, lower, skg);
  const bool use_sk = frac || stg::gemm_use_split(M, N, K, lower, skg) || getenv("HQPKKT_DGEMM_FORCE_SPLIT");
  const bool big = use_sk || stg::gemm_big_tiles(M, N, lower, K);
  const int b = big ? 128 : 64;
  const long long tiles = stg::gemm_tiles(M, N, b, lower);
  (void)stg::gemm_set_attributes();
  // stream-K form where the engine would use it (staged_host.hip.h, st_gemm)
  double *skws = nullptr;
  unsigned *skcnt = nullptr;
  if (use_sk) {
    if (hipMalloc((void **)&skws, sizeof(double) * (size_t)std::max<long long>(16 * tiles + 8, 2LL * skg + 2) * 128 * 128) != hipSuccess ||
        hipMalloc((void **)&skcnt, sizeof(unsigned) * (tiles + 4)) != hipSuccess) {
      (void)hipFree(skws), (void)hipFree(skcnt);
      return fin(HQPKKT_E_MEM);
    }
  }
  // (the cut form by a table with unequal shares for the two workgroups of a CU: gemm_split_table; HQPKKT_SK_TABLE=0: equal shares)
  stg::SplitTable sk_tab;
  if (use_sk && !frac && stg::gemm_sk_table_from_env() && stg::gemm_split_table(tiles, (K + stg::GEMM_BK - 1) / stg::GEMM_BK, skg, sk_tab) &&
      sk_tab.pieces <= 16 * tiles + 8) {
    const size_t nu = sk_tab.units.size();
    if (hipMalloc((void **)&sk_table_dev, sizeof(stg::SkUnit) * nu) != hipSuccess ||
        hipMemcpy(sk_table_dev, sk_tab.units.data(), sizeof(stg::SkUnit) * nu, hipMemcpyHostToDevice) != hipSuccess) {
      (void)hipFree(skws), (void)hipFree(skcnt);
      return fin(HQPKKT_E_MEM);
    }
  }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  for (int r = -1; r < reps; r++) {
    if (r == 0) (void)hipEventRecord(e0, 0);
    if (use_sk) {
      (void)hipMemsetAsync(skcnt, 0, sizeof(unsigned) * (tiles + 4), 0);
      stg::SplitPlan skk = frac ? stg::gemm_split_plan_frac(tiles, (K + stg::GEMM_BK - 1) / stg::GEMM_BK, skg)
                                : stg::gemm_split_plan(tiles, (K + stg::GEMM_BK - 1) / stg::GEMM_BK, skg);
      skk.ws = skws, skk.cnt = skcnt;
      if (sk_table_dev) skk.table = sk_table_dev, skk.stride = sk_tab.stride;
      stg::gemm_launch_split(variant, skg, 0, g, skk);
    } else if (big)
      stg::gemm_launch_plain(variant, (unsigned)tiles, 0, g, cus);
    else if (stg::gemm_tiles_6432(M, N, K, lower, mirror, cus))  // (as st_gemm chooses)
      stg::k_dgemm_tn<64, 32><<<(unsigned)(((M + 63) / 64) * (long long)((N + 31) / 32)), 256, stg::gemm_lds_bytes(64, 32)>>>(g);
    else
      stg::k_dgemm_tn<64, 64><<<(unsigned)tiles, 256, stg::gemm_lds_bytes(64, 64)>>>(g);
  }
  (void)hipEventRecord(e1, 0);
  hipError_t se = hipDeviceSynchronize();
  (void)hipFree(skws), (void)hipFree(skcnt);
  float t = 0.f;
  (void)hipEventElapsedTime(&t, e0, e1);
  (void)hipEventDestroy(e0), (void)hipEventDestroy(e1);
  if (se != hipSuccess) return fin(HQPKKT_E_DEVICE);
  if (getenv("HQPKKT_DGEMM_STAMPS") && use_sk && !frac) {
    // the split form with time stamps: per workgroup its start and, per unit, the end of the k loop, of the
    // parking / summing of partial tiles and of the epilogue (us after the first start)
    unsigned long long *st = nullptr;
    double *ws2 = nullptr;
    unsigned *cnt2 = nullptr;
    if (hipMalloc((void **)&st, sizeof(unsigned long long) * 32 * skg) == hipSuccess &&
        hipMalloc((void **)&ws2, sizeof(double) * (size_t)(16 * tiles + 8) * 128 * 128) == hipSuccess &&
        hipMalloc((void **)&cnt2, sizeof(unsigned) * (tiles + 4)) == hipSuccess) {
      (void)hipMemset(st, 0, sizeof(unsigned long long) * 32 * skg);
      (void)hipMemset(cnt2, 0, sizeof(unsigned) * (tiles + 4));
      stg::GemmArgs gs = g;
      gs.stamps = st;
      stg::SplitPlan skk = stg::gemm_split_plan(tiles, (K + stg::GEMM_BK - 1) / stg::GEMM_BK, skg);
      skk.ws = ws2, skk.cnt = cnt2;
      if (sk_table_dev) skk.table = sk_table_dev, skk.stride = sk_tab.stride;
      stg::gemm_launch_split(variant, skg, 0, gs, skk);
      std::vector<unsigned long long> hs(32 * (size_t)skg);
      if (hipMemcpy(hs.data(), st, sizeof(unsigned long long) * 32 * skg, hipMemcpyDeviceToHost) == hipSuccess) {
        unsigned long long tmin = ~0ULL;
        for (int w = 0; w < skg; w++) tmin = std::min(tmin, hs[32 * (size_t)w]);
        if (sk_table_dev)
          fprintf(stderr, "table plan: %d / %d whole tiles per first / second workgroup of a CU, %lld parked pieces", sk_tab.nA, sk_tab.nB, sk_tab.pieces);
        else {
          fprintf(stderr, "split plan: %d whole tiles", skk.whole);
          for (int q = 0; q < skk.nphase; q++) fprintf(stderr, ", %d tiles x %d pieces", skk.count[q], skk.split[q]);
        }
        fprintf(stderr, "; stamps of every %dth workgroup (us): start | per unit: k loop end, parked / summed, epilogue end\n", std::max(1, skg / 32));
        const int nr = sk_table_dev ? std::min(10, sk_tab.stride - 1) : std::min(5, skk.dp_rounds + skk.nphase);
        for (int w = 0; w < skg; w += std::max(1, skg / 32)) {
          fprintf(stderr, "  wg %4d: %7.2f |", w, (hs[32 * (size_t)w] - tmin) * 0.01);
          for (int r = 0; r < nr; r++) {
            for (int c = 1; c <= 3; c++) {
              const unsigned long long x = hs[32 * (size_t)w + 3 * r + c];
              if (x) fprintf(stderr, " %8.2f", (x - tmin) * 0.01); else fprintf(stderr, "        -");
            }
            fprintf(stderr, " |");
          }
          fprintf(stderr, "\n");
        }
        // the end of every workgroup's last unit, per class (first / second half of the launch)
        for (int c = 0; c < 2; c++) {
          double lo = 1e30, hi = 0.0, sum = 0.0;
          int n = 0;
          for (int w = c * skg / 2; w < (c + 1) * skg / 2; w++) {
            unsigned long long last = 0;
            for (int r = 0; r < 10; r++) last = std::max(last, hs[32 * (size_t)w + 3 * r + 3]);
            if (!last) continue;
            const double e = (last - tmin) * 0.01;
            lo = std::min(lo, e), hi = std::max(hi, e), sum += e, n++;
          }
          if (n) fprintf(stderr, "  class %c (blockIdx %s grid / 2): last epilogue ends at %.1f ... %.1f us, mean %.1f\n", c ? 'B' : 'A', c ? ">=" : "<", lo, hi, sum / n);
        }
      }
    }
    (void)hipFree(st), (void)hipFree(ws2), (void)hipFree(cnt2);
  }
  if (getenv("HQPKKT_DGEMM_STAMPS") && !use_sk && big) {
    // one more launch with time stamps per workgroup (100 MHz constant clock): when it started, when its k loop
    // ended, when its epilogue ended - relative to the first start; printed as a histogram over the workgroups
    unsigned long long *st = nullptr;
    if (hipMalloc((void **)&st, sizeof(unsigned long long) * 4 * tiles) == hipSuccess) {
      stg::GemmArgs gs = g;
      gs.stamps = st;
      stg::gemm_launch_plain(variant, (unsigned)tiles, 0, gs);
      std::vector<unsigned long long> hs(4 * tiles);
      if (hipMemcpy(hs.data(), st, sizeof(unsigned long long) * 4 * tiles, hipMemcpyDeviceToHost) == hipSuccess) {
        unsigned long long tmin = ~0ULL;
        for (long long t = 0; t < tiles; t++) tmin = std::min(tmin, hs[4 * t]);
        // workgroups in the order of their start
        std::vector<long long> ord(tiles);
        for (long long t = 0; t < tiles; t++) ord[t] = t;
        std::sort(ord.begin(), ord.end(), [&](long long a, long long b) { return hs[4 * a] < hs[4 * b]; });
        fprintf(stderr, "stamps (us after the first start; %lld workgroups, every %lldth in start order): start, k loop end, epilogue end, xcc, blockIdx\n", tiles,
                std::max<long long>(1, tiles / 64));
        for (long long q = 0; q < tiles; q += std::max<long long>(1, tiles / 64)) {
          const long long t = ord[q];
          fprintf(stderr, "  %8.2f %8.2f %8.2f  xcc %llu  wg %lld\n", (hs[4 * t] - tmin) * 0.01, (hs[4 * t + 2] - tmin) * 0.01, (hs[4 * t + 3] - tmin) * 0.01,
                  hs[4 * t + 1], t);
        }
      }
      (void)hipFree(st);
    }
  }
  k_gemm_check<<<16, 256>>>(g, 4096, err);
  double he = 0.0;
  if (hipMemcpy(&he, err, 8, hipMemcpyDeviceToHost) != hipSuccess) return fin(HQPKKT_E_DEVICE);
  if (ms) *ms = t / reps;
  if (max_err) *max_err = he;
  return fin(0);
}

int hqpkkt_debug_sk_table(long long tiles, int nslab, int grid, int *units, long long cap_ints, long long *pieces, int *whole_a, int *whole_b) {
  stg::SplitTable t;
  if (!stg::gemm_split_table(tiles, nslab, grid, t)) return 0;
  if (pieces) *pieces = t.pieces;
  if (whole_a) *whole_a = t.nA;
  if (whole_b) *whole_b = t.nB;
  if (units) {
    if ((long long)t.units.size() * 6 > cap_ints) return 0;
    for (size_t i = 0; i < t.units.size(); i++) {
      const stg::SkUnit &u = t.units[i];
      int *o = units + 6 * i;
      o[0] = u.tile, o[1] = u.s0, o[2] = u.s1, o[3] = u.slot0, o[4] = u.pieces, o[5] = u.j;
    }
  }
  return t.stride;
}

int hqpkkt_set_profile(hqpkkt_t *h, int on) {
  if (!h) return HQPKKT_E_NULL;
  h->prof.on = on != 0;
  h->prof.reset();
  return 0;
}

int hqpkkt_get_profile(const hqpkkt_t *h, int n_classes, double *ms, long long *launches) {
  if (!h || !ms || !launches) return HQPKKT_E_NULL;
  for (int c = 0; c < n_classes && c < KC_COUNT; c++) ms[c] = h->prof.ms[c], launches[c] = h->prof.launches[c];
  return KC_COUNT;
}

const char *hqpkkt_profile_class_name(int c) { return (c >= 0 && c < KC_COUNT) ? kc_names[c] : ""; }

int hqpkkt_get_stats(const hqpkkt_t *h, hqpkkt_stats *out) {
  if (!h || !out) return HQPKKT_E_NULL;
  *out = h->st;
  return 0;
}

const char *hqpkkt_strerror(int status) {
  switch (status) {
    case HQPKKT_OK: return "ok";
    case HQPKKT_E_SIZES: return "sizes of arguments mismatch";
    case HQPKKT_E_MEM: return "out of memory";
    case HQPKKT_E_SING: return "matrix is singular";
    case HQPKKT_E_FORMAT: return "CSR input not sorted / out of range";
    case HQPKKT_E_NULL: return "NULL objects passed";
    case HQPKKT_E_RANGE: return "parameter out of range";
    case HQPKKT_E_INTERN: return "call order violated";
    case HQPKKT_E_DEVICE: return g_last_hip_error[0] ? g_last_hip_error : "HIP device error";
    default: return "unknown status";
  }
}

// diagnostics: run the solve `reps` times with time stamps inside k_solve_top (eager launches) and return, per fused
// front, level and six times in microseconds after the launch's first stamp: start, static data in, children arrived,
// forward done, border solution arrived, backward done (out: top_n x 8 doubles, [0] = tree level, [1..6] the times)
int hqpkkt_debug_solve_top_stamps(hqpkkt_t *h, double *out, int cap) {
  if (!h || !out) return HQPKKT_E_NULL;
  if (!h->factored || h->top_n <= 0) return HQPKKT_E_INTERN;
  if (cap < h->top_n * 8) return HQPKKT_E_SIZES;
  HIPCHK(hipSetDevice(h->opts.device));
  unsigned long long *st = nullptr;
  HIPCHK(hipMalloc((void **)&st, sizeof(unsigned long long) * 8 * h->top_n));
  (void)hipMemset(st, 0, sizeof(unsigned long long) * 8 * h->top_n);
  const bool graphs = h->use_graphs;
  h->use_graphs = false, h->top_stamps = st;
  Vecs v{};
  {  // the staged vectors of the last solve (the layout of stage_in)
    const int n = h->an.n, me = h->an.me, m = h->an.m;
    double *b = h->vin.p;
    v.z = b, v.w = b + m, v.r1 = b + 2 * (size_t)m, v.r2 = v.r1 + n, v.r3 = v.r2 + me, v.r4 = v.r3 + m;
  }
  stage_out_ptrs(h, v);
  int e = do_step(h, v, 0);
  if (!e && hipStreamSynchronize(h->stream) != hipSuccess) e = HQPKKT_E_DEVICE;
  h->use_graphs = graphs, h->top_stamps = nullptr;
  if (!e) {  // stamps of a sweep that gave up on a poll mean nothing
    int gave_up[XW_GAVE_UP + 1] = {};
    if (hipMemcpy(gave_up + XW_GAVE_UP, h->flags.p + XW_GAVE_UP, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess || poll_fallback(h, gave_up)) e = HQPKKT_E_DEVICE;
  }
  std::vector<unsigned long long> hs(8 * (size_t)h->top_n);
  if (!e && hipMemcpy(hs.data(), st, sizeof(unsigned long long) * hs.size(), hipMemcpyDeviceToHost) != hipSuccess) e = HQPKKT_E_DEVICE;
  (void)hipFree(st);
  if (e) return e;
  std::vector<int> nodes(h->top_n);
  HIPCHK(hipMemcpy(nodes.data(), h->top_nodes.p, sizeof(int) * h->top_n, hipMemcpyDeviceToHost));
  unsigned long long t0 = ~0ULL;
  for (int t = 0; t < h->top_n; t++) t0 = std::min(t0, hs[8 * (size_t)t]);
  for (int t = 0; t < h->top_n; t++) {
    out[8 * t] = h->an.level[nodes[t]];
    // split form: the backward launch has its own start (slot 6) and static-data (slot 7) stamps; they are returned in
    // place of nothing - out[7] = start of the backward launch of this front
    for (int k = 0; k < 6; k++) out[8 * t + 1 + k] = (double)(hs[8 * (size_t)t + k] - t0) * 0.01;  // 100 MHz
    out[8 * t + 7] = hs[8 * (size_t)t + 6] ? (double)(hs[8 * (size_t)t + 6] - t0) * 0.01 : 0.0;
  }
  return 0;
}

int hqpkkt_debug_get(const hqpkkt_t *h, int what, int *out, long long *len) {
  if (!h || !len) return HQPKKT_E_NULL;
  if (!h->analyzed) return HQPKKT_E_INTERN;
  const Analysis &an = h->an;
  const std::vector<int> *v = nullptr;
  std::vector<int> tmp;
  switch (what) {
    case 0: v = &an.q2e; break;
    case 1: v = &an.piv_start; break;
    case 2: v = &an.npiv; break;
    case 3: v = &an.nbor; break;
    case 4: v = &an.parent; break;
    case 5: v = &an.level; break;
    case 6:
      tmp.assign(an.bptr.begin(), an.bptr.end());
      v = &tmp;
      break;
    case 7: v = &an.bidx; break;
    case 8: v = &an.ent_er; break;
    case 9: v = &an.ent_ec; break;
    case 10: v = &an.node_owner; break;
    case 11: v = &an.xroots; break;
    case 20: case 21: case 22: case 23: case 24: case 25: case 26: {  // STAGED: the plan
      if (!h->sd) return HQPKKT_E_INTERN;
      const kktdev::StagedPlan &P = h->sd->plan;
      if (what == 20) v = &P.nk;
      if (what == 21) v = &P.mk;
      if (what == 22) v = &P.nmk;
      if (what == 23) v = &P.eq_ptr;
      if (what == 24) v = &P.eq_rows;
      if (what == 25) v = &P.fix_rows;
      if (what == 26) v = &P.cap;
      break;
    }
    case 30:  // zero-diagonal placement in use, and whether the last values have weak Hessian diagonals
      tmp = {h->zd_used, h->zd_weak ? 1 : 0};
      v = &tmp;
      break;
    case 31:  // the solve's fused top (k_solve_top): number of fronts, first fused level, LDS bytes
      tmp = {h->top_n, h->top_n ? h->top_lt : h->an.nlevels, (int)h->top_lds, h->top_ns, h->small_tree ? 1 : 0, h->tree_factor ? 1 : 0, h->top_split ? 1 : 0};
      v = &tmp;
      break;
    case 27: {  // STAGED over several ranks: column cuts, (K+1) x (ranks+1)
      if (!h->sd) return HQPKKT_E_INTERN;
      v = &h->sd->plan.xcut;
      break;
    }
    case 33: {  // STAGED over several ranks: the blocks of G_xx, 10 ints each: stage, block row, block column, r0, r1, c0,
                // c1, owner, computed in the owner's own rows (1) or transposed (0), offset inside the owner's slot
      if (!h->sd) return HQPKKT_E_INTERN;
      const kktdev::StagedPlan &P = h->sd->plan;
      for (int k = 0; k < P.K && !P.xrect_ptr.empty(); k++)
        for (int q = P.xrect_ptr[k]; q < P.xrect_ptr[k + 1]; q++) {
          const kktdev::StagedPlan::XRect &x = P.xrects[q];
          for (int val : {k, x.a, x.b, x.r0, x.r1, x.c0, x.c1, x.owner, x.mine_rows ? 1 : 0, (int)x.off}) tmp.push_back(val);
        }
      v = &tmp;
      break;
    }
    case 34: {  // ... and this rank's tiles of its blocks' products: per stage a count, then the tiles (tile row in the strip << 16 | tile column)
      if (!h->sd) return HQPKKT_E_INTERN;
      const kktdev::StagedPlan &P = h->sd->plan;
      for (int k = 0; k < P.K && !P.gtile_ptr.empty(); k++) {
        tmp.push_back(P.gtile_ptr[k + 1] - P.gtile_ptr[k]);
        for (int q = P.gtile_ptr[k]; q < P.gtile_ptr[k + 1]; q++) tmp.push_back(P.gtile[q]);
      }
      v = &tmp;
      break;
    }
    case 28: {  // STAGED: [0] stages whose blocked elimination ran, [1] those of them that fell back to the one-workgroup form
      if (!h->sd) return HQPKKT_E_INTERN;
      tmp.assign(2, 0);
      if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(tmp.data(), h->flags.p + 6, sizeof(int) * 2, hipMemcpyDeviceToHost) != hipSuccess)
        return HQPKKT_E_DEVICE;
      v = &tmp;
      break;
    }
    case 32: {  // STAGED, free initial state of many components: [0] blocked inverse ran, [1] fell back to the LU factors
      if (!h->sd) return HQPKKT_E_INTERN;
      // ... [2] the pivot block that gave up (1-based, 0: none), [3], [4] |K_jj|, |K_jj^-1| of that block, [5] max |K0 K0^-1 - I|
      // (floats as their bit patterns; the words of the blocked sweep's scratch area as the LAST factorisation left them)
      tmp.assign(6, 0);
      if (hipDeviceSynchronize() != hipSuccess ||
          hipMemcpy(tmp.data(), h->flags.p + stg::X0_BLOCKED, sizeof(int) * 2, hipMemcpyDeviceToHost) != hipSuccess)
        return HQPKKT_E_DEVICE;
      if (h->sd->plan.big0 &&
          hipMemcpy(tmp.data() + 2, stg::big_scratch(h->sd->misc.p + h->sd->plan.oScr, h->sd->plan.q0max).flags + 1, sizeof(int) * 4,
                    hipMemcpyDeviceToHost) != hipSuccess)
        return HQPKKT_E_DEVICE;
      v = &tmp;
      break;
    }
    default: return HQPKKT_E_RANGE;
  }
  *len = (long long)v->size();
  if (out && !v->empty()) std::memcpy(out, v->data(), sizeof(int) * v->size());
  return 0;
}

#ifdef HQPKKT_STAMPS
int hqpkkt_debug_stamps(hqpkkt_t *h, int *out) {
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out, h->flags.p, sizeof(int) * 64, hipMemcpyDeviceToHost));
  return 0;
}
int hqpkkt_debug_fb_stamps(int *out) {
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(kktdev::g_fb_stamps), sizeof(int) * 256));
  return 0;
}
int hqpkkt_debug_ps_stamps(int *out) {
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(kktdev::g_ps_stamps), sizeof(int) * 64));
  return 0;
}
int hqpkkt_debug_gj_stamps(int *out) {
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(stg::g_gj_stamps), sizeof(int) * 32));
  return 0;
}
#endif

int hqpkkt_debug_read(hqpkkt_t *h, int what, int node, double *out, long long cap, long long *len) {
  if (!h || !len) return HQPKKT_E_NULL;
  if (!h->uploaded) return HQPKKT_E_INTERN;
  const Analysis &an = h->an;
  if (node < 0 || node >= an.nnodes) return HQPKKT_E_RANGE;
  const long long p = an.npiv[node], b = an.nbor[node];
  const double *src = nullptr;
  long long n = 0;
  switch (what) {
    case 0: src = h->panel.p + an.panel_off[node], n = (p + b) * p; break;
    case 1: src = h->linv.p + an.linv_off[node], n = p * p; break;
    case 2: src = h->xar.p + an.x_off[node], n = b * p; break;
    case 3: src = h->upd.p + an.upd_off[node], n = b * b; break;
    default: return HQPKKT_E_RANGE;
  }
  *len = n;
  if (!out) return 0;
  if (cap < n) return HQPKKT_E_SIZES;
  HIPCHK(hipSetDevice(h->opts.device));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (what == 3 && h->tree_factor) {  // the block is in the exchange copy of the last factorisation (lower triangle; the rest idle)
    int ep = 0;
    HIPCHK(hipMemcpy(&ep, h->tree_words.p + 1, sizeof(int), hipMemcpyDeviceToHost));
    src = h->tree_u.p + (long long)(ep & 1) * an.upd_elems + an.upd_off[node];
  }
  if (n) HIPCHK(hipMemcpy(out, src, sizeof(double) * n, hipMemcpyDeviceToHost));
  if (what == 3 && h->tree_factor)
    for (long long t = 0; t < n; t++)
      if (std::memcmp(out + t, &XW_SENTINEL, sizeof(double)) == 0) out[t] = 0.0;
  return 0;
}

// One dense symmetric p x p block through the pivot-block kernel on its own (tests, tools): A row-major;
// variant 0 = k_factor_blk as run_factor launches it (8 wavefronts for p <= 128, 12 beyond: five blocks per wavefront up to
// 160 pivots, six up to 176, eight up to 192), 1 = k_factor_diag (p <= 128), 2 = the 12-wavefront instance with eight blocks whatever p,
// 3 = the 12-wavefront instance with six blocks (p <= 176).  Out: the block's panel (p x p column-major: unit lower L11
// below the diagonal), D^-1 (2 p), pivot types, pivot order, M = L11^-1 (p x p column-major), the counters
// (2x2 pivots, perturbed, slow pivots, ...), and the average time of `reps` launches of one workgroup.
int hqpkkt_debug_factor_block(int device, int p, const double *A, double tol, double pivot_eps, int variant,
                              int reps, double *Lout, double *dinv_out, int *ptype_out, int *lperm_out,
                              double *Wout, int *counters_out, double *ms_out) {
  if (!A || p < 1 || p > 192 || (variant == 1 && p > 128)) return HQPKKT_E_RANGE;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device) return HQPKKT_E_DEVICE;
  HIPCHK(hipSetDevice(device));
  if (reps < 1) reps = 1;
  const size_t pp2 = (size_t)p * p;
  std::vector<double> P(pp2 * reps);
  double kmax = 0.0;
  for (int j = 0; j < p; j++)
    for (int i = 0; i < p; i++) {
      P[(size_t)j * p + i] = i >= j ? A[(size_t)i * p + j] : 0.0;
      kmax = std::fmax(kmax, std::fabs(A[(size_t)i * p + j]));
    }
  for (int rp = 1; rp < reps; rp++) std::memcpy(P.data() + pp2 * rp, P.data(), sizeof(double) * pp2);
  std::vector<int> piv_start(reps), npiv(reps, p), nbor(reps, 0), parent(reps, -1), child_ptr(reps + 1, 0), nodes(reps);
  std::vector<long long> zeros(reps + 1, 0), poff(reps), loff(reps);
  std::vector<signed char> sg((size_t)p * reps);
  for (int rp = 0; rp < reps; rp++) {
    piv_start[rp] = rp * p, nodes[rp] = rp, poff[rp] = (long long)pp2 * rp, loff[rp] = (long long)pp2 * rp;
    for (int i = 0; i < p; i++) sg[(size_t)rp * p + i] = A[(size_t)i * p + i] < 0.0 ? -1 : 1;
  }
  DBuf<int> d_ps, d_np, d_nb, d_par, d_cp, d_nodes, d_pt, d_lp, d_flags, d_one;
  DBuf<long long> d_zero, d_poff, d_loff;
  DBuf<double> d_P, d_dinv, d_W, d_upd;
  DBuf<signed char> d_sg;
  std::vector<int> fl(128, 0), onei(4, 0);
  std::memcpy(fl.data() + 120, &kmax, sizeof(double));
  int e;
  if ((e = d_ps.upload(piv_start)) || (e = d_np.upload(npiv)) || (e = d_nb.upload(nbor)) || (e = d_par.upload(parent)) ||
      (e = d_cp.upload(child_ptr)) || (e = d_nodes.upload(nodes)) || (e = d_zero.upload(zeros)) || (e = d_poff.upload(poff)) ||
      (e = d_loff.upload(loff)) || (e = d_P.upload(P)) || (e = d_sg.upload(sg)) || (e = d_flags.upload(fl)) ||
      (e = d_one.upload(onei)) || (e = d_dinv.alloc(2 * (size_t)p * reps)) || (e = d_W.alloc(pp2 * reps)) ||
      (e = d_upd.alloc(8)) || (e = d_pt.alloc((size_t)p * reps)) || (e = d_lp.alloc((size_t)p * reps)))
    return e;
  HIPCHK(hipMemset(d_W.p, 0, sizeof(double) * pp2 * reps));
  DevTree T{d_ps.p, d_np.p, d_nb.p, d_par.p, d_zero.p, d_one.p, d_one.p, d_poff.p, d_zero.p, d_zero.p, d_zero.p,
            d_cp.p, d_one.p, d_one.p, d_zero.p};
  const double alpha = tol * 0.6403882032022076;
  const unsigned long long *kb = (const unsigned long long *)(d_flags.p + 120);
  const size_t mpd = p, ldm = mpd | 1;
  const size_t lds_old = (std::max<size_t>(ldm * mpd, 2 * FD_PLD * FD_PANEL) + 5 * 128 + 2 * mpd) * sizeof(double) + 2 * mpd * sizeof(int) + 16;
  if (variant == 1)
    HIPCHK(hipFuncSetAttribute((const void *)k_factor_diag, hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max<size_t>(lds_old, 64 * 1024)));
  HIPCHK(hipFuncSetAttribute((const void *)k_factor_blk<8, 6, 144, 2, FB_OWNSIMD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fb_lds_bytes(128)));
  HIPCHK(hipFuncSetAttribute((const void *)k_factor_blk<12, 8, 208, 3, FB_OWNSIMD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fb_lds_bytes(192)));
  HIPCHK(hipFuncSetAttribute((const void *)k_factor_blk<12, 6, 208, 3, FB_OWNSIMD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fb_lds_bytes(192)));
  HIPCHK(hipFuncSetAttribute((const void *)k_factor_blk<12, FB_NS160, 208, 3, FB_OWNSIMD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fb_lds_bytes(192)));
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipEventRecord(e0, 0));
  for (int rp = 0; rp < reps; rp++) {
    if (variant == 1)
      k_factor_diag<<<1, FD_THREADS, lds_old, 0>>>(T, d_nodes.p + rp, d_P.p, d_dinv.p, d_pt.p, d_lp.p, d_sg.p, d_W.p, d_loff.p,
                                                  alpha, pivot_eps, kb, d_flags.p + 1, d_upd.p);
    else if (variant == 0 && p <= 128)
      k_factor_blk<8, 6, 144, 2, FB_OWNSIMD><<<1, 512, fb_lds_bytes(p), 0>>>(T, d_nodes.p + rp, d_P.p, d_dinv.p, d_pt.p, d_lp.p, d_sg.p, d_W.p,
                                                        d_loff.p, alpha, pivot_eps, kb, d_flags.p + 1, d_upd.p);
    else if (variant == 0 && p <= 160)  // (as run_factor chooses: five blocks per wavefront up to 160 pivots, six up to 176)
      k_factor_blk<12, FB_NS160, 208, 3, FB_OWNSIMD><<<1, 768, fb_lds_bytes(std::max(p, 129)), 0>>>(T, d_nodes.p + rp, d_P.p, d_dinv.p, d_pt.p, d_lp.p, d_sg.p, d_W.p,
                                                          d_loff.p, alpha, pivot_eps, kb, d_flags.p + 1, d_upd.p);
    else if ((variant == 0 || variant == 3) && p <= 176)
      k_factor_blk<12, 6, 208, 3, FB_OWNSIMD><<<1, 768, fb_lds_bytes(std::max(p, 129)), 0>>>(T, d_nodes.p + rp, d_P.p, d_dinv.p, d_pt.p, d_lp.p, d_sg.p, d_W.p,
                                                          d_loff.p, alpha, pivot_eps, kb, d_flags.p + 1, d_upd.p);
    else
      k_factor_blk<12, 8, 208, 3, FB_OWNSIMD><<<1, 768, fb_lds_bytes(std::max(p, 129)), 0>>>(T, d_nodes.p + rp, d_P.p, d_dinv.p, d_pt.p, d_lp.p, d_sg.p, d_W.p,
                                                          d_loff.p, alpha, pivot_eps, kb, d_flags.p + 1, d_upd.p);
  }
  HIPCHK(hipEventRecord(e1, 0));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipGetLastError());
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0), (void)hipEventDestroy(e1);
  if (ms_out) *ms_out = ms / reps;
  const size_t last = (size_t)(reps - 1);
  if (Lout) HIPCHK(hipMemcpy(Lout, d_P.p + pp2 * last, sizeof(double) * pp2, hipMemcpyDeviceToHost));
  if (Wout) HIPCHK(hipMemcpy(Wout, d_W.p + pp2 * last, sizeof(double) * pp2, hipMemcpyDeviceToHost));
  if (dinv_out) HIPCHK(hipMemcpy(dinv_out, d_dinv.p + 2 * (size_t)p * last, sizeof(double) * 2 * p, hipMemcpyDeviceToHost));
  if (ptype_out) HIPCHK(hipMemcpy(ptype_out, d_pt.p + (size_t)p * last, sizeof(int) * p, hipMemcpyDeviceToHost));
  if (lperm_out) HIPCHK(hipMemcpy(lperm_out, d_lp.p + (size_t)p * last, sizeof(int) * p, hipMemcpyDeviceToHost));
  if (counters_out) HIPCHK(hipMemcpy(counters_out, d_flags.p, sizeof(int) * 128, hipMemcpyDeviceToHost));
  DBuf<int> *ib[] = {&d_ps, &d_np, &d_nb, &d_par, &d_cp, &d_nodes, &d_pt, &d_lp, &d_flags, &d_one};
  for (auto b : ib) b->release();
  d_zero.release(), d_poff.release(), d_loff.release(), d_P.release(), d_dinv.release(), d_W.release(), d_upd.release(), d_sg.release();
  return 0;
}

int hqpkkt_selftest_mfma(int device, double *max_err) {
  if (!max_err) return HQPKKT_E_NULL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device) return HQPKKT_E_DEVICE;
  HIPCHK(hipSetDevice(device));
  double A[256], B[256], Cx[256], Cd[256];
  for (int i = 0; i < 16; i++)
    for (int j = 0; j < 16; j++) {
      A[i * 16 + j] = (double)((i * 7 + j * 3) % 11 - 5);
      B[i * 16 + j] = (double)((i * 5 + j * 13) % 17 - 8);
    }
  for (int i = 0; i < 16; i++)
    for (int j = 0; j < 16; j++) {
      double s = 0;
      for (int k = 0; k < 16; k++) s += A[i * 16 + k] * B[k * 16 + j];
      Cx[i * 16 + j] = s;
    }
  double *dA, *dB, *dC;
  HIPCHK(hipMalloc((void **)&dA, sizeof(A)));
  HIPCHK(hipMalloc((void **)&dB, sizeof(B)));
  HIPCHK(hipMalloc((void **)&dC, sizeof(Cd)));
  HIPCHK(hipMemcpy(dA, A, sizeof(A), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dB, B, sizeof(B), hipMemcpyHostToDevice));
  k_mfma_selftest<<<1, 64>>>(dA, dB, dC);
  HIPCHK(hipMemcpy(Cd, dC, sizeof(Cd), hipMemcpyDeviceToHost));
  (void)hipFree(dA), (void)hipFree(dB), (void)hipFree(dC);
  double err = 0;
  for (int i = 0; i < 256; i++) err = std::fmax(err, std::fabs(Cd[i] - Cx[i]));
  *max_err = err;
  return 0;
}

}  // extern "C"
