// TEST INFRASTRUCTURE -- not part of the product path.
//
// Drives the REFERENCE's own interior-point QP solvers (hqp/Hqp_IpsMehrotra.C,
// hqp/Hqp_IpsFranke.C, unmodified, compiled from /root/reference by
// oracle/Makefile) on a QP given as CSR arrays, with the KKT plugin chosen by
// name through the reference's own plugin registry ("qp_mat_solver <name>",
// iftcl/If_Module.h:50-96).  With name = "SpBKP"/"RedSpBKP" this is the pure
// reference; with "SpBKPHip"/"RedSpBKPHip" (only in libhqphost_hip.so, which also
// holds shim/Hqp_IpSpBKPHip.C) the SAME solver object code calls our C ABI --
// the drop-in check of BASELINE.json's north_star ("Hqp_IpsMehrotra /
// Hqp_IpsFranke call it unchanged").  This file is our code; nothing of the
// reference is copied.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>

#include <If.h>
#include <Hqp_Program.h>
#include <Hqp_IpsMehrotra.h>
#include <Hqp_IpsFranke.h>

extern "C" int hqpref_startup(void);

static double now_s() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static void fill(SPMAT *M, int rows, const int *p, const int *i, const double *x) {
  for (int r = 0; r < rows; r++)
    for (int k = p[r]; k < p[r + 1]; k++) sp_set_val(M, r, i[k], x[k]);
}

extern "C" {

// qp_init_method of the solvers created below (hqp/Hqp_IpsMehrotra.C:124), set through the
// reference's own Tcl variable after the solver exists
static int g_init_method = 0;
void hqpip_set_init_method(int v) { g_init_method = v; }
// qp_mu0 of the Franke solvers created below (hqp/Hqp_IpsFranke.C:77,87), set the same way
static double g_mu0 = 0.0;
void hqpip_set_mu0(double v) { g_mu0 = v; }

// solver: 0 = Mehrotra, 1 = Franke, 2 = MehrotraHip, 3 = FrankeHip (our Hqp_Solver plugins).  Returns 0, or the Meschach error number,
// or -1 (setup) / -2 (unknown plugin name).
// out[0] = iterations, out[1] = Hqp_Result (0 optimal), out[2] = seconds in
// cold_start + solve, out[3] = seconds in init + update, out[4] = mat_sbw after the solve.
int hqpip_solve(int solver, const char *mat_solver, int n, int me, int m, const int *Qp,
                const int *Qi, const double *Qx, const double *c, const int *Ap,
                const int *Ai, const double *Ax, const double *b, const int *Cp,
                const int *Ci, const double *Cx, const double *d, double qp_eps,
                int max_iters, double *x, double *y, double *z, double *out) {
  if (hqpref_startup() != 0) return -1;
  Hqp_Solver *S;
  if (solver >= 2) {
    // our device-resident solver classes (shim/Hqp_IpsMehrotraHip.C), created BY NAME through
    // the reference's solver factory (iftcl/If_Class.h:92-105) as "sqp_qp_solver MehrotraHip"
    // would do; only registered in libhqphost_hip.so
    S = If_ClassList_Hqp_Solver() ? If_ClassList_Hqp_Solver()->createObject(solver == 2 ? "MehrotraHip" : "FrankeHip")
                                  : NULL;
    if (!S) return -2;
  } else
    S = solver == 0 ? (Hqp_Solver *)new Hqp_IpsMehrotra : (Hqp_Solver *)new Hqp_IpsFranke;
  // select the plugin exactly as a user would: Tcl command qp_mat_solver
  if (If_SetString("qp_mat_solver", mat_solver) != IF_OK) {
    delete S;
    return -2;
  }
  Hqp_Program *qp = new Hqp_Program;
  qp->resize(n, me, m);
  fill(qp->Q, n, Qp, Qi, Qx);
  fill(qp->A, me, Ap, Ai, Ax);
  fill(qp->C, m, Cp, Ci, Cx);
  for (int i = 0; i < n; i++) qp->c->ve[i] = c[i], qp->x->ve[i] = 0.0;
  for (int i = 0; i < me; i++) qp->b->ve[i] = b[i];
  for (int i = 0; i < m; i++) qp->d->ve[i] = d[i];
  S->qp(qp);
  S->eps(qp_eps);
  S->max_iters(max_iters);
  if (solver == 0 || solver == 2) (void)If_SetInt("qp_init_method", g_init_method);
  if (solver == 1 || solver == 3) (void)If_SetReal("qp_mu0", g_mu0);
  int err = 0;
  double t0 = now_s(), t1 = t0, t2 = t0;
  m_catchall(S->init(); S->update(); t1 = now_s(); S->cold_start(); S->solve(); t2 = now_s(),
             err = _err_num);
  if (!err) {
    for (int i = 0; i < n; i++) x[i] = qp->x->ve[i];
    for (int i = 0; i < me; i++) y[i] = S->y()->ve[i];
    for (int i = 0; i < m; i++) z[i] = S->z()->ve[i];
    out[0] = S->iter();
    out[1] = (double)S->result();
    out[2] = t2 - t1;
    out[3] = t1 - t0;
    int sbw = -2;
    (void)If_GetInt("mat_sbw", &sbw);
    out[4] = sbw;   // the plugin's read-only member (our STAGED engine reports -1: no band)
  }
  delete S;
  delete qp;
  return err;
}

// Diagnosis: Hqp_IpsFranke from a cold start, step by step (what its solve() does around step(), hqp/Hqp_IpsFranke.C:381-416,
// without the hot-start branches), the solver's scalars after every step: trace[6 k ..] = gap, alpha, alphabar, zeta,
// rhomin, Hqp_Result.  Returns 0 / the Meschach error; *niter = steps taken.
namespace {
struct FrankeProbe : public Hqp_IpsFranke {
  void get(double *t) const { t[0] = _gap, t[1] = _alpha, t[2] = _alphabar, t[3] = _zeta, t[4] = _rhomin, t[5] = (double)_result; }
};
}  // namespace
int hqpip_trace_franke(const char *mat_solver, int n, int me, int m, const int *Qp, const int *Qi, const double *Qx,
                       const double *c, const int *Ap, const int *Ai, const double *Ax, const double *b, const int *Cp,
                       const int *Ci, const double *Cx, const double *d, double qp_eps, int max_iters, double *trace,
                       int *niter) {
  if (hqpref_startup() != 0) return -1;
  FrankeProbe *S = new FrankeProbe;
  if (If_SetString("qp_mat_solver", mat_solver) != IF_OK) {
    delete S;
    return -2;
  }
  Hqp_Program *qp = new Hqp_Program;
  qp->resize(n, me, m);
  fill(qp->Q, n, Qp, Qi, Qx);
  fill(qp->A, me, Ap, Ai, Ax);
  fill(qp->C, m, Cp, Ci, Cx);
  for (int i = 0; i < n; i++) qp->c->ve[i] = c[i], qp->x->ve[i] = 0.0;
  for (int i = 0; i < me; i++) qp->b->ve[i] = b[i];
  for (int i = 0; i < m; i++) qp->d->ve[i] = d[i];
  S->qp(qp);
  S->eps(qp_eps);
  S->max_iters(max_iters);
  (void)If_SetReal("qp_mu0", g_mu0);
  int err = 0, k = 0;
  m_catchall(S->init(); S->update(); S->cold_start();
             for (k = 0; k < max_iters;) {
               S->step();
               S->get(trace + 6 * k);
               k++;
               const int r = (int)S->result();
               if (r == Hqp_Optimal || r == Hqp_Suboptimal || r == Hqp_Degenerate) break;
             },
             err = _err_num);
  *niter = k;
  delete S;
  delete qp;
  return err;
}

// Diagnosis: a HOT-started Hqp_IpsFranke step by step - (c, b, d) solved from a cold start, then (c2, b2, d2) after update()
// + hot_start() and the loop of Hqp_IpsFranke::solve() (hqp/Hqp_IpsFranke.C:381-416) spelt out around step(), so that
// the scalars of every step can be recorded: trace[8 k ..] = gap, alpha, alphabar, zeta, rhomin, Hqp_Result, hot (1 while
// the hot start is alive, 0 after the restart from a cold start), iter as the solver counts it.  out[0] = total
// iterations (with the failed ones), out[1] = result, out[2] = iterations of the first solve.
namespace {
struct FrankeHotProbe : public Hqp_IpsFranke {
  void get(double *t) const {
    t[0] = _gap, t[1] = _alpha, t[2] = _alphabar, t[3] = _zeta, t[4] = _rhomin, t[5] = (double)_result, t[6] = (double)_hot_started,
    t[7] = (double)_iter;
  }
  int hot() const { return _hot_started; }
  int iters() const { return _iter; }
  int max_warm() const { return _max_warm_iters; }
};
}  // namespace
int hqpip_trace_franke_hot(const char *mat_solver, int n, int me, int m, const int *Qp, const int *Qi, const double *Qx,
                           const double *c, const int *Ap, const int *Ai, const double *Ax, const double *b, const int *Cp,
                           const int *Ci, const double *Cx, const double *d, const double *c2, const double *b2,
                           const double *d2, double qp_eps, int max_iters, double *trace, int *nsteps, double *out) {
  if (hqpref_startup() != 0) return -1;
  FrankeHotProbe *S = new FrankeHotProbe;
  if (If_SetString("qp_mat_solver", mat_solver) != IF_OK) {
    delete S;
    return -2;
  }
  Hqp_Program *qp = new Hqp_Program;
  qp->resize(n, me, m);
  fill(qp->Q, n, Qp, Qi, Qx);
  fill(qp->A, me, Ap, Ai, Ax);
  fill(qp->C, m, Cp, Ci, Cx);
  for (int i = 0; i < n; i++) qp->c->ve[i] = c[i], qp->x->ve[i] = 0.0;
  for (int i = 0; i < me; i++) qp->b->ve[i] = b[i];
  for (int i = 0; i < m; i++) qp->d->ve[i] = d[i];
  S->qp(qp);
  S->eps(qp_eps);
  S->max_iters(max_iters);
  (void)If_SetReal("qp_mu0", g_mu0);
  int err = 0, k = 0, it1 = 0, fail = 0;
  // (the loop as a lambda: the commas of its body would split the macro's arguments; an error longjmps out of it)
  auto run = [&]() {
    S->init();
    S->update();
    S->cold_start();
    S->solve();
    it1 = S->iter();
    for (int i = 0; i < n; i++) qp->c->ve[i] = c2[i];
    for (int i = 0; i < me; i++) qp->b->ve[i] = b2[i];
    for (int i = 0; i < m; i++) qp->d->ve[i] = d2[i];
    S->update();
    S->hot_start();
    double gap1 = 0.0, t[8];
    for (;;) {
      for (;;) {
        S->step();
        if (k < max_iters) {
          S->get(trace + 8 * k);
          k++;
        }
        if (S->hot()) {
          S->get(t);
          if (S->iters() == 1)
            gap1 = t[0];
          else if (t[0] > gap1) {
            fail += S->iters();
            S->cold_start();
          }
        }
        if (S->iters() + fail >= max_iters) break;
        if (S->hot() && S->iters() >= S->max_warm()) break;
        const int r = (int)S->result();
        if (r == Hqp_Optimal || r == Hqp_Suboptimal || r == Hqp_Degenerate) break;
      }
      if (S->hot() && S->result() != Hqp_Optimal) {
        fail += S->iters();
        S->cold_start();
      } else
        break;
    }
  };
  m_catchall(run();, err = _err_num);
  *nsteps = k;
  out[0] = S->iter() + fail, out[1] = (double)S->result(), out[2] = it1;
  delete S;
  delete qp;
  return err;
}

// Time of Hqp_Solver::update() (= plugin update(): new values on the same pattern, once per SQP
// iteration, hqp/Hqp_SqpSolver.C:285-296) with the plugin `mat_solver`: out[0] = median seconds of
// `reps` updates (values scaled a little in between), out[1] = seconds of init + first update.
int hqpip_time_update(const char *mat_solver, int n, int me, int m, const int *Qp, const int *Qi,
                      const double *Qx, const int *Ap, const int *Ai, const double *Ax, const int *Cp,
                      const int *Ci, const double *Cx, int reps, double *out) {
  if (hqpref_startup() != 0) return -1;
  Hqp_Solver *S = new Hqp_IpsMehrotra;
  if (If_SetString("qp_mat_solver", mat_solver) != IF_OK) {
    delete S;
    return -2;
  }
  Hqp_Program *qp = new Hqp_Program;
  qp->resize(n, me, m);
  fill(qp->Q, n, Qp, Qi, Qx);
  fill(qp->A, me, Ap, Ai, Ax);
  fill(qp->C, m, Cp, Ci, Cx);
  S->qp(qp);
  int err = 0;
  double ts[64];
  if (reps > 64) reps = 64;
  double t0 = now_s(), t1 = t0;
  m_catchall(S->init(); S->update(); t1 = now_s();
             for (int r = 0; r < reps; r++) {
               for (int i = 0; i < n; i++) {
                 SPROW *row = qp->Q->row + i;
                 for (int j = 0; j < row->len; j++) row->elt[j].val *= 1.0009765625;
               }
               const double a = now_s();
               S->update();
               ts[r] = now_s() - a;
             },
             err = _err_num);
  if (!err) {
    for (int a = 0; a < reps; a++)
      for (int b = a + 1; b < reps; b++)
        if (ts[b] < ts[a]) { const double t = ts[a]; ts[a] = ts[b], ts[b] = t; }
    out[0] = reps ? ts[reps / 2] : 0.0;
    out[1] = t1 - t0;
  }
  delete S;
  delete qp;
  return err;
}

// Two QPs in a row with the same matrices, as an SQP iteration makes them: (c, b, d) solved
// from a cold start, then (c2, b2, d2) after update() + hot_start()
// (hqp/Hqp_SqpSolver.C:285-296, hqp/Hqp_IpsMehrotra.C:330-352, 696-733).  x, y, z and out
// receive the SECOND solve; out[4] = iterations of the first.
int hqpip_solve_hot(int solver, const char *mat_solver, int n, int me, int m, const int *Qp,
                    const int *Qi, const double *Qx, const double *c, const int *Ap,
                    const int *Ai, const double *Ax, const double *b, const int *Cp,
                    const int *Ci, const double *Cx, const double *d, const double *c2,
                    const double *b2, const double *d2, double qp_eps, int max_iters, double *x,
                    double *y, double *z, double *out) {
  if (hqpref_startup() != 0) return -1;
  Hqp_Solver *S;
  if (solver == 2) {
    S = If_ClassList_Hqp_Solver() ? If_ClassList_Hqp_Solver()->createObject("MehrotraHip") : NULL;
    if (!S) return -2;
  } else
    S = solver == 0 ? (Hqp_Solver *)new Hqp_IpsMehrotra : (Hqp_Solver *)new Hqp_IpsFranke;
  if (If_SetString("qp_mat_solver", mat_solver) != IF_OK) {
    delete S;
    return -2;
  }
  Hqp_Program *qp = new Hqp_Program;
  qp->resize(n, me, m);
  fill(qp->Q, n, Qp, Qi, Qx);
  fill(qp->A, me, Ap, Ai, Ax);
  fill(qp->C, m, Cp, Ci, Cx);
  for (int i = 0; i < n; i++) qp->c->ve[i] = c[i], qp->x->ve[i] = 0.0;
  for (int i = 0; i < me; i++) qp->b->ve[i] = b[i];
  for (int i = 0; i < m; i++) qp->d->ve[i] = d[i];
  S->qp(qp);
  S->eps(qp_eps);
  S->max_iters(max_iters);
  if (solver == 0 || solver == 2) (void)If_SetInt("qp_init_method", g_init_method);
  if (solver == 1 || solver == 3) (void)If_SetReal("qp_mu0", g_mu0);
  int err = 0, it1 = 0;
  double t1 = now_s(), t2 = t1;
  m_catchall(S->init(); S->update(); S->cold_start(); S->solve(); it1 = S->iter();
             for (int i = 0; i < n; i++) qp->c->ve[i] = c2[i];
             for (int i = 0; i < me; i++) qp->b->ve[i] = b2[i];
             for (int i = 0; i < m; i++) qp->d->ve[i] = d2[i];
             S->update(); t1 = now_s(); S->hot_start(); S->solve(); t2 = now_s(),
             err = _err_num);
  if (!err) {
    for (int i = 0; i < n; i++) x[i] = qp->x->ve[i];
    for (int i = 0; i < me; i++) y[i] = S->y()->ve[i];
    for (int i = 0; i < m; i++) z[i] = S->z()->ve[i];
    out[0] = S->iter();
    out[1] = (double)S->result();
    out[2] = t2 - t1;
    out[3] = 0.0;
    out[4] = it1;
  }
  delete S;
  delete qp;
  return err;
}

}  // extern "C"
