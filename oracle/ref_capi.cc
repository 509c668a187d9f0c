// TEST INFRASTRUCTURE -- not part of the product path.
//
// Plain-C handle API over the REAL reference classes Hqp_IpSpBKP /
// Hqp_IpRedSpBKP (hqp/Hqp_IpSpBKP.C:76-218, hqp/Hqp_IpRedSpBKP.C:184-368,
// hqp/Hqp_IpMatrix.C:65-178), compiled from the sources where they lie under
// /root/reference by oracle/Makefile into oracle/_ref/libhqpref.so.
// This file is OUR code: it only builds an Hqp_Program from CSR arrays, calls
// the reference's virtual methods and copies results out.  Nothing of the
// reference is copied into the repository.
//
// Used (a) to pin oracle/kkt_oracle.c, (b) to generate tests/golden/*.npz,
// (c) as bench.py's cpu_baseline of kind "reference".

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>

#include <If.h>
#include <Hqp_Program.h>
#include <Hqp_IpSpBKP.h>
#include <Hqp_IpRedSpBKP.h>
#include <Hqp_IpLQDOCP.h>

namespace {

// expose protected members of the reference classes for capture
struct ProbeSpBKP : public Hqp_IpSpBKP {
  PERM *qp2j() { return _QP2J; }
  PERM *pivot() { return _pivot; }
  SPMAT *J() { return _J; }
  SPMAT *Jraw() { return _J_raw; }
  int sbw() { return _sbw; }
  void set_tol(double t) { _tol = t; }
  void set_eps(double e) { _eps = e; }
};
struct ProbeRedSpBKP : public Hqp_IpRedSpBKP {
  PERM *qp2j() { return _QP2J; }
  PERM *pivot() { return _pivot; }
  SPMAT *J() { return _J; }
  SPMAT *Jraw() { return _J_raw; }
  int sbw() { return _sbw; }
  void set_tol(double t) { _tol = t; }
  void set_eps(double e) { _eps = e; }
};

struct Handle {
  int kind;  // 0 SpBKP, 1 RedSpBKP, 2 LQDOCP (multistage Riccati, hqp/Hqp_IpLQDOCP.C)
  ProbeSpBKP *full;
  ProbeRedSpBKP *red;
  Hqp_IpMatrix *mat;
  Hqp_Program *qp;
  int n, me, m;
  VEC *z, *w, *r1, *r2, *r3, *r4, *dx, *dy, *dz, *dw;
};

bool g_interp_ready = false;

void fill_values(SPMAT *M, int rows, const int *p, const int *i,
                 const double *x) {
  for (int r = 0; r < rows; r++)
    for (int k = p[r]; k < p[r + 1]; k++) sp_set_val(M, r, i[k], x[k]);
}

VEC *vec_from(const double *src, int dim) {
  VEC *v = v_get(dim > 0 ? dim : 1);
  v->dim = dim;
  if (src)
    for (int k = 0; k < dim; k++) v->ve[k] = src[k];
  return v;
}

void load(VEC *v, const double *src) {
  for (unsigned k = 0; k < v->dim; k++) v->ve[k] = src[k];
}
void store(const VEC *v, double *dst) {
  for (unsigned k = 0; k < v->dim; k++) dst[k] = v->ve[k];
}

double now_s() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

}  // namespace

// Every entry returns 0 on success or the Meschach error number (E_SING = 4)
// raised by the reference (meschach/err.h:84-110); errors are caught with the
// reference's own m_catchall (meschach/err.h) so nothing longjmps across the
// C boundary.
extern "C" {

int hqpref_startup(void) {
  if (g_interp_ready) return 0;
  // the reference's plugin ctors register Tcl commands
  // (iftcl/If_Element.C:42-49), so an interpreter must exist first
  if (If_CreateInterp(0, NULL) != IF_OK) return -1;
  g_interp_ready = true;
  return 0;
}

void *hqpref_create(int kind) {
  if (hqpref_startup() != 0) return NULL;
  Handle *h = (Handle *)calloc(1, sizeof(Handle));
  h->kind = kind;
  if (kind == 0) {
    h->full = new ProbeSpBKP;
    h->mat = h->full;
  } else if (kind == 1) {
    h->red = new ProbeRedSpBKP;
    h->mat = h->red;
  } else {
    h->mat = new Hqp_IpLQDOCP;  // no probe: only the virtual interface is used
  }
  return h;
}

void hqpref_set_params(void *hv, double tol, double eps) {
  Handle *h = (Handle *)hv;
  if (h->full) {
    h->full->set_tol(tol);
    h->full->set_eps(eps);
  } else if (h->red) {
    h->red->set_tol(tol);
    h->red->set_eps(eps);
  }
}

// Q: n x n (any stored entries; the reference reads j >= i only), A: me x n,
// C: m x n, all 0-based CSR.
int hqpref_init(void *hv, int n, int me, int m, const int *Qp, const int *Qi,
                const double *Qx, const int *Ap, const int *Ai,
                const double *Ax, const int *Cp, const int *Ci,
                const double *Cx, double *seconds) {
  Handle *h = (Handle *)hv;
  int err = 0;
  h->n = n;
  h->me = me;
  h->m = m;
  Hqp_Program *qp = new Hqp_Program;
  qp->resize(n, me, m);
  fill_values(qp->Q, n, Qp, Qi, Qx);
  fill_values(qp->A, me, Ap, Ai, Ax);
  fill_values(qp->C, m, Cp, Ci, Cx);
  h->qp = qp;
  h->z = vec_from(NULL, m);
  h->w = vec_from(NULL, m);
  h->r1 = vec_from(NULL, n);
  h->r2 = vec_from(NULL, me);
  h->r3 = vec_from(NULL, m);
  h->r4 = vec_from(NULL, m);
  h->dx = vec_from(NULL, n);
  h->dy = vec_from(NULL, me);
  h->dz = vec_from(NULL, m);
  h->dw = vec_from(NULL, m);
  double t0 = now_s();
  m_catchall(h->mat->init(qp), err = _err_num);
  if (seconds) *seconds = now_s() - t0;
  return err;
}

int hqpref_update(void *hv, const int *Qp, const int *Qi, const double *Qx,
                  const int *Ap, const int *Ai, const double *Ax,
                  const int *Cp, const int *Ci, const double *Cx) {
  Handle *h = (Handle *)hv;
  int err = 0;
  fill_values(h->qp->Q, h->n, Qp, Qi, Qx);
  fill_values(h->qp->A, h->me, Ap, Ai, Ax);
  fill_values(h->qp->C, h->m, Cp, Ci, Cx);
  m_catchall(h->mat->update(h->qp), err = _err_num);
  return err;
}

int hqpref_factor(void *hv, const double *z, const double *w,
                  double *seconds) {
  Handle *h = (Handle *)hv;
  int err = 0;
  load(h->z, z);
  load(h->w, w);
  double t0 = now_s();
  m_catchall(h->mat->factor(h->qp, h->z, h->w), err = _err_num);
  if (seconds) *seconds = now_s() - t0;
  return err;
}

static void load_rhs(Handle *h, const double *z, const double *w,
                     const double *r1, const double *r2, const double *r3,
                     const double *r4) {
  load(h->z, z);
  load(h->w, w);
  load(h->r1, r1);
  load(h->r2, r2);
  load(h->r3, r3);
  load(h->r4, r4);
}
static void store_d(Handle *h, double *dx, double *dy, double *dz,
                    double *dw) {
  store(h->dx, dx);
  store(h->dy, dy);
  store(h->dz, dz);
  store(h->dw, dw);
}

// Hqp_IpMatrix::step of the subclass (one triangular-solve pair, no refinement)
int hqpref_step(void *hv, const double *z, const double *w, const double *r1,
                const double *r2, const double *r3, const double *r4,
                double *dx, double *dy, double *dz, double *dw) {
  Handle *h = (Handle *)hv;
  int err = 0;
  load_rhs(h, z, w, r1, r2, r3, r4);
  m_catchall(h->mat->step(h->qp, h->z, h->w, h->r1, h->r2, h->r3, h->r4,
                          h->dx, h->dy, h->dz, h->dw),
             err = _err_num);
  store_d(h, dx, dy, dz, dw);
  return err;
}

// Hqp_IpMatrix::solve = step + iterative refinement (hqp/Hqp_IpMatrix.C:65-128)
int hqpref_solve(void *hv, const double *z, const double *w, const double *r1,
                 const double *r2, const double *r3, const double *r4,
                 double *dx, double *dy, double *dz, double *dw, double *res,
                 double *seconds) {
  Handle *h = (Handle *)hv;
  int err = 0;
  double r = -1.0;
  load_rhs(h, z, w, r1, r2, r3, r4);
  double t0 = now_s();
  m_catchall(r = h->mat->solve(h->qp, h->z, h->w, h->r1, h->r2, h->r3, h->r4,
                               h->dx, h->dy, h->dz, h->dw),
             err = _err_num);
  if (seconds) *seconds = now_s() - t0;
  store_d(h, dx, dy, dz, dw);
  if (res) *res = r;
  return err;
}

// Hqp_IpMatrix::residuum on caller-provided d* (hqp/Hqp_IpMatrix.C:131-178)
int hqpref_residuum(void *hv, const double *z, const double *w,
                    const double *r1, const double *r2, const double *r3,
                    const double *r4, const double *dx, const double *dy,
                    const double *dz, const double *dw, double *res) {
  Handle *h = (Handle *)hv;
  int err = 0;
  double r = -1.0;
  load_rhs(h, z, w, r1, r2, r3, r4);
  load(h->dx, dx);
  load(h->dy, dy);
  load(h->dz, dz);
  load(h->dw, dw);
  m_catchall(r = h->mat->residuum(h->qp, h->z, h->w, h->r1, h->r2, h->r3,
                                  h->r4, h->dx, h->dy, h->dz, h->dw),
             err = _err_num);
  if (res) *res = r;
  return err;
}

int hqpref_sbw(void *hv) {
  Handle *h = (Handle *)hv;
  return h->full ? h->full->sbw() : h->red ? h->red->sbw() : -1;
}

int hqpref_dim(void *hv) {
  Handle *h = (Handle *)hv;
  return h->red ? h->n + h->me : h->n + h->me + h->m;
}

void hqpref_get_perm(void *hv, int *qp2j) {
  Handle *h = (Handle *)hv;
  PERM *p = h->full ? h->full->qp2j() : h->red->qp2j();
  for (unsigned k = 0; k < p->size; k++) qp2j[k] = (int)p->pe[k];
}

void hqpref_get_pivot(void *hv, int *pivot) {
  Handle *h = (Handle *)hv;
  PERM *p = h->full ? h->full->pivot() : h->red->pivot();
  for (unsigned k = 0; k < p->size; k++) pivot[k] = (int)p->pe[k];
}

// which = 0: _J_raw (assembled, before w/z + scaling), 1: _J (after factor)
long hqpref_matrix_nnz(void *hv, int which) {
  Handle *h = (Handle *)hv;
  SPMAT *J = h->full ? (which ? h->full->J() : h->full->Jraw())
                     : (which ? h->red->J() : h->red->Jraw());
  if (!J) return -1;
  long nnz = 0;
  for (int r = 0; r < J->m; r++) nnz += J->row[r].len;
  return nnz;
}

// CSR copy of the reference's row-list matrix (rowptr has dim+1 entries)
void hqpref_get_matrix(void *hv, int which, int *rowptr, int *col,
                       double *val) {
  Handle *h = (Handle *)hv;
  SPMAT *J = h->full ? (which ? h->full->J() : h->full->Jraw())
                     : (which ? h->red->J() : h->red->Jraw());
  long k = 0;
  for (int r = 0; r < J->m; r++) {
    rowptr[r] = (int)k;
    for (int e = 0; e < J->row[r].len; e++, k++) {
      col[k] = J->row[r].elt[e].col;
      val[k] = J->row[r].elt[e].val;
    }
  }
  rowptr[J->m] = (int)k;
}

void hqpref_destroy(void *hv) {
  Handle *h = (Handle *)hv;
  if (!h) return;
  delete h->mat;
  delete h->qp;
  v_free(h->z);
  v_free(h->w);
  v_free(h->r1);
  v_free(h->r2);
  v_free(h->r3);
  v_free(h->r4);
  v_free(h->dx);
  v_free(h->dy);
  v_free(h->dz);
  v_free(h->dw);
  free(h);
}

}  // extern "C"
