/*
 * Hqp_IpSpBKPHip.C -- see Hqp_IpSpBKPHip.h.  Thin C++ shim: walks the
 * Hqp_Program's row-list SPMATs into int32 CSR, forwards the Hqp_IpMatrix
 * virtual calls to libhqpkkt.so and converts a non-zero status into the
 * reference's error convention (m_error -> longjmp to the IP solver's m_catch,
 * meschach/err.h:63,118-135; hqp/Hqp_IpsMehrotra.C:525-536).
 *
 * No C++ object with a non-trivial destructor is alive across a possible
 * m_error(): everything the shim owns hangs off `this` as Meschach objects.
 */
// (standard headers first: hqp/Meschach.h defines min / max macros, hqp/Meschach.h:38-43)
#include <thread>
#include <vector>

#include "Hqp_IpSpBKPHip.h"

#include <If_Int.h>
#include <If_Real.h>

#include <dlfcn.h>

#include "Hqp_Program.h"
#include "hqpkkt.h"
#include "hqpkkt_rccl.h"
#include "stage_extract.h"

IF_CLASS_DEFINE("SpBKPHip", Hqp_IpSpBKPHip, Hqp_IpMatrix);
IF_CLASS_DEFINE("RedSpBKPHip", Hqp_IpRedSpBKPHip, Hqp_IpMatrix);
IF_CLASS_DEFINE("LQDOCPHip", Hqp_IpLQDOCPHip, Hqp_IpMatrix);

//--------------------------------------------------------------------------
Hqp_IpMatrixHip::Hqp_IpMatrixHip(int mode)
{
  _mode = mode;
  _mode_used = mode;
  _n = _me = _m = 0;
  _sbw = -1;
  _tol = 1.0;
  _device = 0;
  _refine = 1;
  _ngpu = 1;
  _rccl = NULL;
  _rccl_lib = NULL;
  _rank = 0;
  _h = NULL;
  _Qp = _Qi = _Ap = _Ai = _Cp = _Ci = IVNULL;
  _Qx = _Ax = _Cx = VNULL;
  _dense = false;
  _K = _ndyn = 0;
  _nx = _nu = IVNULL;
  _wz_tol = HUGE_VAL;
  _a_sparse = 0;
  _told_ignored = false;
  _logging = getenv("HQPKKT_SHIM_LOGGING") ? atoi(getenv("HQPKKT_SHIM_LOGGING")) : 0;
  if (mode == HQPKKT_MODE_STAGED) {
    _ifList.append(new If_Real("mat_wz_tol", &_wz_tol));
    _ifList.append(new If_Int("mat_a_sparse", &_a_sparse));
    _ifList.append(new If_Int("mat_logging", &_logging));
  }

  // same Tcl-visible members as hqp/Hqp_IpSpBKP.C:58-59, plus the device knobs
  _ifList.append(new If_Int("mat_sbw", &_sbw));
  _ifList.append(new If_Real("mat_tol", &_tol));
  _ifList.append(new If_Int("mat_device", &_device));
  _ifList.append(new If_Int("mat_device_refine", &_refine));
  // mat_ngpu > 1: ONE KKT system over that many GPUs.  The HQP host is then started once per GPU
  // (mpirun / torchrun style: RANK, WORLD_SIZE, LOCAL_RANK in the environment, every process running
  // the same deterministic SQP iteration); the plugins of the processes form an RCCL communicator
  // (libhqpkkt_rccl.so, include/hqpkkt_rccl.h) and share every factorisation
  _ifList.append(new If_Int("mat_ngpu", &_ngpu));
  // host threads that walk the row lists in update() (0: the single-threaded re-extraction)
  _update_threads = (int)std::thread::hardware_concurrency();
  if (_update_threads > 16) _update_threads = 16;
  if (_update_threads < 1) _update_threads = 1;
  _ifList.append(new If_Int("mat_update_threads", &_update_threads));
  // 1: nested dissection of the graph itself instead of the RCM band (irregular sparsity, e.g. what
  // hqp_cute/hqp_cute.tcl:22-46 runs through RedSpBKP); takes effect at the next init()
  _ordering = 0;
  _ifList.append(new If_Int("mat_ordering", &_ordering));
  // LQDOCPHip: a DOCP whose widest stage front (nx_k + nu_k + nx_k+1) is below this goes to the tree engine -
  // the stage-by-stage chain of dense products pays from a few hundred states per stage on (K = 200, nu = 10:
  // nx = 50 takes 15 ms staged against 2.1 ms through the tree, nx = 400 30 against 25)
  _staged_min_front = 800;
  if (getenv("HQPKKT_STAGED_MIN_FRONT")) _staged_min_front = atoi(getenv("HQPKKT_STAGED_MIN_FRONT"));
  _ifList.append(new If_Int("mat_staged_min_front", &_staged_min_front));
}

//--------------------------------------------------------------------------
Hqp_IpMatrixHip::~Hqp_IpMatrixHip()
{
  hqpkkt_destroy(_h);
  if (_rccl && _rccl_lib) {
    typedef int (*destroy_t)(void *);
    destroy_t f = (destroy_t)dlsym(_rccl_lib, "hqpkkt_rccl_destroy");
    if (f) f(_rccl);
  }
  iv_free(_Qp); iv_free(_Qi); iv_free(_Ap); iv_free(_Ai); iv_free(_Cp); iv_free(_Ci);
  iv_free(_nx); iv_free(_nu);
  v_free(_Qx); v_free(_Ax); v_free(_Cx);
}

//--------------------------------------------------------------------------
void Hqp_IpMatrixHip::check(int status, const char *where)
{
  if (status == HQPKKT_OK)
    return;
  // numerical singularity / zero slack -> E_SING, which the IP solvers catch and
  // turn into Hqp_Degenerate (hqp/Hqp_IpsMehrotra.C:525-536, Hqp_IpsFranke.C:303-310)
  if (status == HQPKKT_E_SING)
    m_error(E_SING, where);
  if (status == HQPKKT_E_SIZES)
    m_error(E_SIZES, where);
  if (status == HQPKKT_E_MEM)
    m_error(E_MEM, where);
  if (status == HQPKKT_E_NULL)
    m_error(E_NULL, where);
  if (status == HQPKKT_E_FORMAT)
    m_error(E_FORMAT, where);
  fprintf(stderr, "%s: %s\n", where, hqpkkt_strerror(status));
  m_error(E_INTERN, where);  // HIP runtime failures, call-order violations
}

//--------------------------------------------------------------------------
// one block: SPMAT -> (ptr, idx, val); upper = keep col >= row only
// (the reference reads only that part of Q, meschach/addon2_hqp.c:1078-1086)
// rows [row0, M->m) only (the equality rows behind the dynamics rows of a DOCP).  Returns false when the block
// has more entries than int32 CSR holds (the caller reports HQPKKT_E_SIZES; the dynamics rows of a large DOCP
// never come this way: open_dense)
static bool extract_block(const SPMAT *M, bool upper,
                          IVEC *&ptr, IVEC *&idx, VEC *&val, bool &changed, int row0 = 0)
{
  int i, j, k, m = M->m - row0;
  long long nnz = 0;
  for (i = 0; i < m; i++) {
    const SPROW *row = M->row + row0 + i;
    for (j = 0; j < row->len; j++)
      if (!upper || row->elt[j].col >= i)
        nnz++;
  }
  if (nnz > 0x7fffffffLL)
    return false;
  if (!ptr || (int)ptr->dim != m + 1 || !idx || (long long)idx->dim != nnz)
    changed = true;
  ptr = iv_resize(ptr, m + 1);
  idx = iv_resize(idx, (int)nnz);
  val = v_resize(val, (int)nnz);
  k = 0;
  for (i = 0; i < m; i++) {
    const SPROW *row = M->row + row0 + i;
    if (ptr->ive[i] != k)
      changed = true;
    ptr->ive[i] = k;
    for (j = 0; j < row->len; j++) {
      const row_elt *elt = row->elt + j;
      if (upper && elt->col < i)
        continue;
      if (idx->ive[k] != elt->col)
        changed = true;
      idx->ive[k] = elt->col;
      val->ve[k] = elt->val;
      k++;
    }
  }
  if (ptr->ive[m] != k)
    changed = true;
  ptr->ive[m] = k;
  return true;
}

//--------------------------------------------------------------------------
// update() on an unchanged pattern: the values of one block straight from the SPROW
// arrays into `val` (pinned staging of the library), rows dealt to `nthr` threads; the
// column indices are compared on the way (the entry is in the cache line just read;
// the reference's PARDISO plugin compares them too, hqp/Hqp_IpPARDISO.C:286-296).
// Returns true if the pattern is no longer the analysed one.
static bool refresh_block(const SPMAT *M, bool upper, const IVEC *ptr, const IVEC *idx,
                          double *val, int nthr)
{
  const int m = M->m;
  if (!ptr || (int)ptr->dim != m + 1)
    return true;
  std::vector<char> bad(nthr, 0);
  auto work = [&](int t) {
    const int lo = (int)((long long)m * t / nthr), hi = (int)((long long)m * (t + 1) / nthr);
    for (int i = lo; i < hi; i++) {
      const SPROW *row = M->row + i;
      const row_elt *elt = row->elt;
      int k = ptr->ive[i];
      const int kend = ptr->ive[i + 1];
      for (int j = 0; j < row->len; j++, elt++) {
        if (upper && elt->col < i)
          continue;
        if (k >= kend || idx->ive[k] != elt->col) {
          bad[t] = 1;
          return;
        }
        val[k++] = elt->val;
      }
      if (k != kend) {
        bad[t] = 1;
        return;
      }
    }
  };
  if (nthr <= 1 || m < 4096)
    for (int t = 0; t < nthr; t++) work(t);
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < nthr; t++) th.emplace_back(work, t);
    for (size_t t = 0; t < th.size(); t++) th[t].join();
  }
  for (int t = 0; t < nthr; t++)
    if (bad[t]) return true;
  return false;
}

void Hqp_IpMatrixHip::extract(const Hqp_Program *qp, bool &pattern_changed)
{
  pattern_changed = _dense;  // (the CSR copies held the rows behind the dynamics only)
  _dense = false;
  bool ok = extract_block(qp->Q, true, _Qp, _Qi, _Qx, pattern_changed);
  ok = extract_block(qp->A, false, _Ap, _Ai, _Ax, pattern_changed) && ok;
  ok = extract_block(qp->C, false, _Cp, _Ci, _Cx, pattern_changed) && ok;
  if (!ok)
    check(HQPKKT_E_SIZES, "Hqp_IpMatrixHip: more than 2^31 entries in a block (only LQDOCPHip's dense stage blocks go beyond)");
}

//--------------------------------------------------------------------------
// the row lists of an SPMAT as shim/stage_extract.h wants them
namespace {
struct SpmatRows {
  const SPMAT *M;
  int len(long long i) const { return M->row[i].len; }
  int col(long long i, int j) const { return M->row[i].elt[j].col; }
  double val(long long i, int j) const { return M->row[i].elt[j].val; }
};
}  // namespace

//--------------------------------------------------------------------------
// The values of a DOCP whose structure open_dense() has analysed: every stage's dense block [fx_k fu_k] walked out
// of the row lists of A into one of the library's two pinned stage buffers (rows dealt to the update threads) and
// copied into the engine's arena while the next stage is walked - what Hqp_IpLQDOCP::update does with
// sp_extract_mat (hqp/Hqp_IpLQDOCP.C:748-755) -, then the values of Q, of the other equality rows and of C.
int Hqp_IpMatrixHip::dense_values(const Hqp_Program *qp)
{
  int e = HQPKKT_OK;
  const SpmatRows A = {qp->A};
  long long row0 = 0;
  int col0 = 0;
  for (int k = 0; k < _K && !e; k++) {
    const int np = _nx->ive[k + 1], nz = _nx->ive[k] + _nu->ive[k], next0 = col0 + nz;
    double *buf = NULL;
    long long cap = 0;
    if ((e = hqpkkt_stage_staging(_h, k & 1, &buf, &cap)))
      break;
    if ((long long)np * nz > cap) {
      e = HQPKKT_E_INTERN;
      break;
    }
    int nthr = _update_threads < 1 ? 1 : _update_threads;
    if ((long long)np * nz < (1LL << 18)) nthr = 1;
    std::vector<long long> got(nthr, 0);
    auto work = [&](int t) {
      const int lo = (int)((long long)np * t / nthr), hi = (int)((long long)np * (t + 1) / nthr);
      for (long long x = (long long)lo * nz; x < (long long)hi * nz; x++) buf[x] = 0.0;
      hqpshim::DenseSink sink = {buf, (long long)nz};
      got[t] = hqpshim::stage_rows(A, row0, lo, hi, col0, nz, next0, sink);
    };
    if (nthr == 1)
      work(0);
    else {
      std::vector<std::thread> th;
      for (int t = 0; t < nthr; t++) th.emplace_back(work, t);
      for (size_t t = 0; t < th.size(); t++) th[t].join();
    }
    for (int t = 0; t < nthr; t++)
      if (got[t] < 0) e = HQPKKT_E_FORMAT;  // an entry outside its stage / not the -1.0 staircase
    if (!e)
      e = hqpkkt_set_stage_block(_h, k, buf, nz);
    row0 += np, col0 = next0;
  }
  if (!e)
    e = hqpkkt_set_values_staged(_h, _Qx->ve, NULL, NULL, _Ax->ve, _Cx->ve);
  return e;
}

//--------------------------------------------------------------------------
// LQDOCPHip: stage sizes from the staircase (three ints per row of A), CSR of Q, of the equality rows behind the
// dynamics and of C, the dynamics as dense blocks.  Returns HQPKKT_E_FORMAT where Hqp_IpLQDOCP::init asserts (no
// DOCP staircase) and HQPKKT_E_SIZES where a stage is beyond the STAGED kernels: init() then routes the same
// system to the full-system engine.
int Hqp_IpMatrixHip::open_dense(const Hqp_Program *qp)
{
  const int rows = qp->A->m;
  int e, K = 0, ndyn = 0;
  bool changed = false;
  if (rows < 1 || _n < 1)
    return HQPKKT_E_FORMAT;
  {
    std::vector<int> len(rows), last(rows), prev(rows), nx(rows + 1), nu(rows);
    const SpmatRows A = {qp->A};
    hqpshim::staircase_keys(A, rows, len.data(), last.data(), prev.data());
    if ((e = hqpkkt_detect_stages(_n, rows, len.data(), last.data(), prev.data(), rows, &K, nx.data(), nu.data(), &ndyn)))
      return e;
    _nx = iv_resize(_nx, K + 1);
    _nu = iv_resize(_nu, K);
    for (int k = 0; k <= K; k++) _nx->ive[k] = nx[k];
    for (int k = 0; k < K; k++) _nu->ive[k] = nu[k];
  }
  _K = K, _ndyn = ndyn;
  if (!extract_block(qp->Q, true, _Qp, _Qi, _Qx, changed) ||
      !extract_block(qp->A, false, _Ap, _Ai, _Ax, changed, ndyn) ||
      !extract_block(qp->C, false, _Cp, _Ci, _Cx, changed))
    return HQPKKT_E_SIZES;
  if ((e = create_handle(HQPKKT_MODE_STAGED)))
    return e;
  _sbw = -1;
  if ((e = hqpkkt_analyze_staged(_h, K, _nx->ive, _nu->ive, _n, _me - ndyn, _m, _Qp->ive, _Qi->ive,
                                 _Ap->ive, _Ai->ive, _Cp->ive, _Ci->ive)))
    return e;
  _dense = true;
  return dense_values(qp);
}

//--------------------------------------------------------------------------
// (re)create the handle for `mode` and hand the structure and the values over.
// Returns the status of the first call that fails (the handle then stays
// created, but not analysed).
int Hqp_IpMatrixHip::create_handle(int mode)
{
  hqpkkt_opts opts;
  int e;

  hqpkkt_destroy(_h);
  _h = NULL;
  hqpkkt_default_opts(&opts);
  opts.mode = mode;
  opts.device = _device;
  opts.loc = HQPKKT_LOC_HOST;   // Meschach VEC::ve pointers
  opts.tol = _tol;
  opts.eps = _eps;
  opts.ordering = _ordering;
  if (_ngpu > 1 && !_rccl) {
    // the communicator is made once per plugin object (it outlives re-inits)
    typedef int (*create_t)(void **, int *, int *, int *);
    int nranks = 1, dev = _device;
    _rccl_lib = dlopen("libhqpkkt_rccl.so", RTLD_NOW | RTLD_LOCAL);
    create_t create = _rccl_lib ? (create_t)dlsym(_rccl_lib, "hqpkkt_rccl_create_from_env") : NULL;
    if (!create || create(&_rccl, &_rank, &nranks, &dev) || nranks != _ngpu) {
      fprintf(stderr, "Hqp_IpMatrixHip: mat_ngpu %d needs libhqpkkt_rccl.so and %d processes "
              "(RANK / WORLD_SIZE / LOCAL_RANK in the environment)\n", _ngpu, _ngpu);
      return HQPKKT_E_DEVICE;
    }
    _device = dev;
    opts.device = dev;
  }
  if ((e = hqpkkt_create(&opts, &_h)))
    return e;
  if (_ngpu > 1) {
    hqpkkt_exchange_stream_fn xfn = (hqpkkt_exchange_stream_fn)dlsym(_rccl_lib, "hqpkkt_rccl_exchange");
    if (!xfn || (e = hqpkkt_set_shard_stream(_h, _rank, _ngpu, xfn, _rccl)))
      return e ? e : HQPKKT_E_DEVICE;
  }
  _mode_used = mode;
  return HQPKKT_OK;
}

int Hqp_IpMatrixHip::open(int mode)
{
  int e;
  if ((e = create_handle(mode)))
    return e;
  _dense = false;
  if ((e = hqpkkt_analyze(_h, _n, _me, _m,
                          _Qp->ive, _Qi->ive, _Ap->ive, _Ai->ive, _Cp->ive, _Ci->ive,
                          &_sbw)))
    return e;
  return hqpkkt_set_values(_h, _Qx->ve, _Ax->ve, _Cx->ve);
}

//--------------------------------------------------------------------------
void Hqp_IpMatrixHip::init(const Hqp_Program *qp)
{
  bool changed;
  int e;

  _n = qp->c->dim;
  _me = qp->b->dim;
  _m = qp->d->dim;

  // the handle is created here: mat_tol / mat_eps / mat_device may have been set
  if (_mode == HQPKKT_MODE_STAGED) {
    // the dynamics as dense stage blocks straight from the row lists (no CSR copy of them: open_dense)
    e = open_dense(qp);
    const char *why = NULL;
    if (e == HQPKKT_E_FORMAT || e == HQPKKT_E_SIZES) {
      // not the staircase of a DOCP (where Hqp_IpLQDOCP::init asserts, hqp/Hqp_IpLQDOCP.C:700-707), or a stage with
      // more controls / carried constraint rows than the STAGED kernels hold: the same KKT system through the
      // full-system engine
      why = e == HQPKKT_E_FORMAT ? "no DOCP staircase" : "a stage beyond the STAGED kernels";
      extract(qp, changed);
      e = open(HQPKKT_MODE_FULL);
    } else if (!e) {
      hqpkkt_stats st;
      if (hqpkkt_get_stats(_h, &st) == HQPKKT_OK && st.max_front < _staged_min_front) {
        why = "stages below mat_staged_min_front";   // small stages are faster through the tree engine
        extract(qp, changed);
        e = open(HQPKKT_MODE_FULL);
      }
    }
    if (_logging > 0) {
      fprintf(stderr, "LQDOCPHip: n %d me %d m %d", _n, _me, _m);
      if (_dense)
        fprintf(stderr, ", %d stages (x_0: %d states; widest stage %d states + %d controls), %d dynamics rows as dense "
                "blocks: STAGED engine\n", _K, _nx->ive[0], _nx->ive[_K], _K ? _nu->ive[0] : 0, _ndyn);
      else
        fprintf(stderr, ": full-system engine (%s), mat_sbw %d\n", why ? why : "", _sbw);
    }
    check(e, "Hqp_IpMatrixHip::init");
    return;
  }
  extract(qp, changed);
  e = open(_mode);
  check(e, "Hqp_IpMatrixHip::init");
}

//--------------------------------------------------------------------------
void Hqp_IpMatrixHip::update(const Hqp_Program *qp)
{
  bool changed;
  int e;
  // fast path: same pattern as analysed -> values straight into the library's pinned staging
  // (no C++ object with a destructor is alive when check() may longjmp: the vectors of
  // refresh_block are gone by then)
  double *sq = NULL, *sa = NULL, *sc = NULL;
  if (_dense) {
    // same stage structure (three ints per row of A are enough to see it) -> new values block by block
    const int rows = qp->A->m;
    bool same = false;
    int st = HQPKKT_OK;
    {
      std::vector<int> len(rows), last(rows), prev(rows), nx(rows + 1), nu(rows);
      const SpmatRows A = {qp->A};
      int K = 0, ndyn = 0;
      hqpshim::staircase_keys(A, rows, len.data(), last.data(), prev.data());
      st = hqpkkt_detect_stages(_n, rows, len.data(), last.data(), prev.data(), rows, &K, nx.data(), nu.data(), &ndyn);
      same = !st && K == _K && ndyn == _ndyn;
      for (int k = 0; same && k <= K; k++) same = nx[k] == _nx->ive[k] && (k == K || nu[k] == _nu->ive[k]);
    }
    changed = false;
    if (same) {
      if (!extract_block(qp->Q, true, _Qp, _Qi, _Qx, changed) ||
          !extract_block(qp->A, false, _Ap, _Ai, _Ax, changed, _ndyn) ||
          !extract_block(qp->C, false, _Cp, _Ci, _Cx, changed))
        check(HQPKKT_E_SIZES, "Hqp_IpMatrixHip::update");
    }
    if (same && !changed) {
      e = dense_values(qp);
      if (e == HQPKKT_E_FORMAT || e == HQPKKT_E_SIZES) {
        extract(qp, changed);
        e = open(HQPKKT_MODE_FULL);
      }
      check(e, "Hqp_IpMatrixHip::update");
      return;
    }
    init(qp);  // another structure: as a new program
    return;
  }
  if (_h && _Qp && _update_threads > 0 && hqpkkt_values_staging(_h, &sq, &sa, &sc) == HQPKKT_OK) {
    bool ch = refresh_block(qp->Q, true, _Qp, _Qi, sq, _update_threads);
    ch = ch || refresh_block(qp->A, false, _Ap, _Ai, sa, _update_threads);
    ch = ch || refresh_block(qp->C, false, _Cp, _Ci, sc, _update_threads);
    if (!ch) {
      e = hqpkkt_set_values(_h, sq, sa, sc);
      if (_mode_used == HQPKKT_MODE_STAGED && (e == HQPKKT_E_FORMAT || e == HQPKKT_E_SIZES)) {
        extract(qp, changed);
        e = open(HQPKKT_MODE_FULL);
      }
      check(e, "Hqp_IpMatrixHip::update");
      return;
    }
  }
  extract(qp, changed);
  if (changed) {
    // structure changed behind our back: analyse again (the reference's own
    // plugins assume a fixed pattern between init() calls)
    e = hqpkkt_analyze(_h, _n, _me, _m,
                       _Qp->ive, _Qi->ive, _Ap->ive, _Ai->ive, _Cp->ive, _Ci->ive,
                       &_sbw);
    if (!e)
      e = hqpkkt_set_values(_h, _Qx->ve, _Ax->ve, _Cx->ve);
  } else
    e = hqpkkt_set_values(_h, _Qx->ve, _Ax->ve, _Cx->ve);
  if (_mode_used == HQPKKT_MODE_STAGED && (e == HQPKKT_E_FORMAT || e == HQPKKT_E_SIZES))
    e = open(HQPKKT_MODE_FULL);   // see init()
  check(e, "Hqp_IpMatrixHip::update");
}

//--------------------------------------------------------------------------
void Hqp_IpMatrixHip::factor(const Hqp_Program *qp, const VEC *z, const VEC *w)
{
  int e;
  assert((int)z->dim == _m && (int)w->dim == _m);
  if (_mode == HQPKKT_MODE_STAGED && !_told_ignored && (_wz_tol != HUGE_VAL || _a_sparse != 0)) {
    // hqp/Hqp_IpLQDOCP.C:850-853 would take ExRiccatiFactor() instead of ExRiccatiFactorSc() with mat_wz_tol set,
    // :437, 738, 1357-1368 the sparse forms of fx, fu with mat_a_sparse: neither exists here (one recursion on dense
    // blocks, same solution up to the residual contract) - said once, never silently
    _told_ignored = true;
    fprintf(stderr, "LQDOCPHip: mat_wz_tol (%g) / mat_a_sparse (%d) are set to non-default values; this plugin has one "
            "recursion (the scaled form of ExRiccatiFactorSc) on dense stage blocks and ignores both\n", (double)_wz_tol, _a_sparse);
  }
  hqpkkt_set_tol(_h, _tol);
  e = hqpkkt_factor(_h, z->ve, w->ve);
  if (_mode_used == HQPKKT_MODE_STAGED && e == HQPKKT_E_SIZES) {
    // a stage is left with more constraint rows to carry back than the STAGED kernels hold (decided
    // by the values: the ranks of the stage constraints): from now on the full-system engine
    bool changed;
    extract(qp, changed);
    e = open(HQPKKT_MODE_FULL);
    if (!e)
      e = hqpkkt_factor(_h, z->ve, w->ve);
  }
  check(e, "Hqp_IpMatrixHip::factor");
}

//--------------------------------------------------------------------------
void Hqp_IpMatrixHip::step(const Hqp_Program *, const VEC *z, const VEC *w,
                           const VEC *r1, const VEC *r2, const VEC *r3,
                           const VEC *r4, VEC *dx, VEC *dy, VEC *dz, VEC *dw)
{
  assert((int)r1->dim == _n && (int)dx->dim == _n);
  assert((int)r2->dim == _me && (int)dy->dim == _me);
  assert((int)r3->dim == _m && (int)dz->dim == _m);
  assert((int)r4->dim == _m && (int)dw->dim == _m);
  check(hqpkkt_step(_h, z->ve, w->ve, r1->ve, r2->ve, r3->ve, r4->ve,
                    dx->ve, dy->ve, dz->ve, dw->ve),
        "Hqp_IpMatrixHip::step");
}

//--------------------------------------------------------------------------
Real Hqp_IpMatrixHip::solve(const Hqp_Program *qp, const VEC *z, const VEC *w,
                            const VEC *r1, const VEC *r2, const VEC *r3,
                            const VEC *r4, VEC *dx, VEC *dy, VEC *dz, VEC *dw)
{
  double res = 0.0;
  if (!_refine)  // host-side refinement of the base class, device step()
    return Hqp_IpMatrix::solve(qp, z, w, r1, r2, r3, r4, dx, dy, dz, dw);
  hqpkkt_set_eps(_h, _eps);
  check(hqpkkt_solve(_h, z->ve, w->ve, r1->ve, r2->ve, r3->ve, r4->ve,
                     dx->ve, dy->ve, dz->ve, dw->ve, &res),
        "Hqp_IpMatrixHip::solve");
  return res;
}
