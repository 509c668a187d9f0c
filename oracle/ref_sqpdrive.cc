// TEST INFRASTRUCTURE -- not part of the product path.
//
// BASELINE.json configs[0]: the reference's hqp_docp demo (hqp_docp/Docp_Main.C: Prg_DID, the
// double integrator, solved by Hqp_SqpPowell) with the reference's OWN SQP solver, SQP program
// classes and interior-point solvers compiled unmodified from /root/reference by
// oracle/Makefile.  The demo's main program goes through Hqp_Init and the Tcl procedure
// hqp_solve (hqp/hqp_solve.tcl, compiled into the library by the reference's build with its
// tpc tool); neither is built here.  This file (our code) is the host program instead: it
// creates the solver and the program objects the way Hqp_Init does (hqp/Hqp_Init.C:199-204),
// configures them through the reference's Tcl variables, and restates the control flow of
// `hqp_solve` (hqp/hqp_solve.tcl:77-250, cold start branch) around the reference's own Tcl
// commands sqp_qp_update / sqp_qp_solve / sqp_step.  With sqp_qp_solver / qp_mat_solver =
// Mehrotra|Franke / SpBKP|RedSpBKP|LQDOCP this is the pure reference; with MehrotraHip|FrankeHip /
// SpBKPHip|RedSpBKPHip|LQDOCPHip (only in libhqphost_hip.so) the same SQP object code drives
// our solver classes and plugins.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <ctime>

#include <If.h>
#include <Hqp_SqpPowell.h>
#include <Hqp_SqpProgram.h>
#include <Hqp_MipSolver.h>
#include "Prg_DID.h"

extern "C" int hqpref_startup(void);

// the application-level objects every HQP host program holds (hqp/Hqp_Init.C:60-66)
Hqp_SqpProgram *theSqpProgram = NULL;
Hqp_SqpSolver *theSqpSolver = NULL;
Hqp_MipSolver *theMipSolver = NULL;

static double now_s() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
static double getr(const char *name) {
  Real v = 0.0;
  (void)If_GetReal(name, &v);
  return v;
}
static int geti(const char *name) {
  int v = 0;
  (void)If_GetInt(name, &v);
  return v;
}
static bool qp_optimal() {
  const char *r = NULL;
  (void)If_GetString("qp_result", &r);
  return r && !strcmp(r, "optimal");
}

// hqp_solve (hqp/hqp_solve.tcl:77-250), cold start branch, around the reference's own Tcl commands;
// 0 = optimal, -1 a command failed, -3 the qp solver made no iteration, -4 iteration limits, -5 stall
static int hqp_solve_loop(double sqp_eps, int &qp_iters) {
  int rc = 0;
  int nullsteps = 0; bool hela_restart = false;
  while (!rc) {
    if (If_Eval("sqp_qp_update") != IF_OK) { rc = -1; break; }      // :100-104, :141
    if (getr("sqp_xQx") < 0.0) {                                      // :160-165
      (void)If_Eval("sqp_hela_restart");
      hela_restart = true;
    } else
      hela_restart = false;
    if (geti("sqp_iter") > 0 && getr("sqp_norm_inf") < sqp_eps && getr("sqp_norm_grd_L") < sqp_eps) break;  // :168-172
    if (If_Eval("sqp_qp_solve") != IF_OK) { rc = -1; break; }       // :174
    const int qi = geti("qp_iter");
    qp_iters += qi;
    if (qi == 0) { rc = -3; break; }                                  // :179-182
    const double sQs = getr("sqp_sQs");
    if (sQs < 0.0) (void)If_Eval("sqp_hela_restart");                 // :187-189
    if (geti("sqp_iter") > 0 && sQs >= 0.0 && !hela_restart && getr("sqp_norm_inf") < sqp_eps && qp_optimal()) {  // :191-201
      if (sQs < sqp_eps * sqp_eps) break;
      if (geti("sqp_iter") > 2 && getr("sqp_norm_s") < sqp_eps * getr("sqp_norm_x") &&
          getr("sqp_norm_df") < sqp_eps * std::fabs(getr("prg_f")) && sQs < sqp_eps)
        break;
    }
    if (If_Eval("sqp_step") != IF_OK) { rc = -1; break; }           // :203
    if (qi >= geti("qp_max_iters")) { rc = -4; break; }               // :206-208 (the "feasible" exemption is not needed here)
    if (geti("sqp_iter") >= geti("sqp_max_iters")) { rc = -4; break; }  // :209-211
    if (geti("sqp_inf_iters") >= geti("sqp_max_inf_iters")) { rc = -4; break; }  // :212-218
    if (getr("sqp_alpha") < 1e-8 && getr("sqp_norm_df") < sqp_eps * std::fabs(getr("prg_f")))  // :220-228
      nullsteps++;
    else
      nullsteps = 0;
    if (nullsteps > 5) { rc = -5; break; }
  }
  return rc;
}


// ---------------------------------------------------------------------------------------------
// BASELINE.json configs[4] ("CUTE-style sparse NLP, full SQP loop"): the CUTE collection needs Fortran
// libraries that are not in this image (hqp/Prg_CUTE.C:194-206 binds csize_/cfn_/csgrsh_), so the
// stand-in is a program class of OUR OWN with the sparsity of a discretised control problem, handed to
// the reference's unmodified Hqp_SqpPowell exactly as Prg_CUTE is (same Hqp_SqpProgram interface,
// hqp/Hqp_SqpProgram.h:60-83; analytic Lagrangian Hessian + sqp_hela Gerschgorin as in
// hqp_cute/hqp_cute.tcl:27,40-41).  One variable per cell of a gx x gy grid:
//   f(x)  = sum_i a_i/2 (x_i - t_i)^2 + g/4 x_i^4 + sum_{i~j} k_ij/2 (x_i - x_j)^2   (right / lower neighbours)
//   g_c(x)= x_c x_{c+1} + x_{c+gx} - beta_c = 0          for every eq_every-th cell c
//   lo <= x_i <= hi                                      for a fraction of the cells
// far > 0 adds that many couplings k/2 (x_i - x_j)^2 between random cells i < j far apart in the numbering (the
// irregular part: "1 % far couplings" on top of the mesh's five entries per row)
class Prg_GridNLP : public Hqp_SqpProgram {
  int _gx, _gy, _n, _me, _nb, _hela, _nfar;
  IVEC *_fi, *_fj;
  VEC *_kf;
  unsigned long long _rs;
  VEC *_a, *_t, *_kr, *_kd, *_beta;
  IVEC *_cells, *_bnd;
  double _gam, _lo, _hi;
  double rnd() {  // xorshift64*: the same program on every host
    _rs ^= _rs >> 12, _rs ^= _rs << 25, _rs ^= _rs >> 27;
    return (double)((_rs * 2685821657736338717ULL) >> 11) / 9007199254740992.0;
  }

 public:
  Prg_GridNLP(int gx, int gy, int seed, int eq_every, double bound_frac, int hela, int far = 0)
      : _gx(gx), _gy(gy), _n(gx * gy), _me(0), _nb(0), _hela(hela), _nfar(0), _rs(88172645463325252ULL + 7919ULL * (unsigned)seed),
        _gam(0.5), _lo(-1.2), _hi(1.2) {
    _a = v_get(_n), _t = v_get(_n), _kr = v_get(_n), _kd = v_get(_n);
    for (int i = 0; i < _n; i++)
      _a->ve[i] = 0.5 + rnd(), _t->ve[i] = 2.0 * rnd() - 1.0, _kr->ve[i] = 0.5 + rnd(), _kd->ve[i] = 0.5 + rnd();
    _cells = iv_get(_n), _bnd = iv_get(_n);
    for (int r = 0; r + 1 < gy; r++)
      for (int c = 0; c + 1 < gx; c++)
        if ((r + c) % eq_every == 0) _cells->ive[_me++] = r * gx + c;
    _beta = v_get(_me > 0 ? _me : 1);
    for (int k = 0; k < _me; k++) _beta->ve[k] = 0.4 * rnd() - 0.2;
    for (int i = 0; i < _n; i++)
      if (rnd() < bound_frac) _bnd->ive[_nb++] = i;
    _fi = iv_get(far > 0 ? far : 1), _fj = iv_get(far > 0 ? far : 1), _kf = v_get(far > 0 ? far : 1);
    for (int k = 0; k < far; k++) {  // (drawn last: the mesh part is the same program with and without them)
      int i = (int)(rnd() * _n), j = (int)(rnd() * _n);
      if (i > j) { const int t = i; i = j; j = t; }
      if (j - i < _n / 8 || j >= _n) continue;  // far apart in the numbering, and not a mesh neighbour
      _fi->ive[_nfar] = i, _fj->ive[_nfar] = j, _kf->ve[_nfar] = 0.05 + 0.1 * rnd(), _nfar++;
    }
  }
  ~Prg_GridNLP() {
    v_free(_a), v_free(_t), v_free(_kr), v_free(_kd), v_free(_beta), iv_free(_cells), iv_free(_bnd);
    iv_free(_fi), iv_free(_fj), v_free(_kf);
  }
  const char *name() { return "GridNLP"; }
  int n() const { return _n; }
  int me() const { return _me; }
  int m() const { return 2 * _nb; }

  void setup() {
    _x = v_resize(_x, _n);
    _qp->resize(_n, _me, 2 * _nb, 3, 3, 1);
    for (int i = 0; i < _n; i++) {
      sp_set_val(_qp->Q, i, i, 1.0);
      if (_hela) {
        if (i % _gx + 1 < _gx) sp_set_val(_qp->Q, i, i + 1, 0.0);
        if (i / _gx + 1 < _gy) sp_set_val(_qp->Q, i, i + _gx, 0.0);
      }
    }
    if (_hela)
      for (int k = 0; k < _nfar; k++) sp_set_val(_qp->Q, _fi->ive[k], _fj->ive[k], 0.0);
    for (int k = 0; k < _me; k++) {
      const int c = _cells->ive[k];
      sp_set_val(_qp->A, k, c, 0.0), sp_set_val(_qp->A, k, c + 1, 0.0), sp_set_val(_qp->A, k, c + _gx, 1.0);
    }
    for (int k = 0; k < _nb; k++)
      sp_set_val(_qp->C, 2 * k, _bnd->ive[k], 1.0), sp_set_val(_qp->C, 2 * k + 1, _bnd->ive[k], -1.0);
    init_x();
  }
  void init_x() {
    for (int i = 0; i < _n; i++) _x->ve[i] = 0.1;
  }
  void update_fbd() {
    const double *x = _x->ve;
    double f = 0.0;
    for (int i = 0; i < _n; i++) {
      const double e = x[i] - _t->ve[i];
      f += 0.5 * _a->ve[i] * e * e + 0.25 * _gam * x[i] * x[i] * x[i] * x[i];
      if (i % _gx + 1 < _gx) f += 0.5 * _kr->ve[i] * (x[i] - x[i + 1]) * (x[i] - x[i + 1]);
      if (i / _gx + 1 < _gy) f += 0.5 * _kd->ve[i] * (x[i] - x[i + _gx]) * (x[i] - x[i + _gx]);
    }
    for (int k = 0; k < _nfar; k++) {
      const double e = x[_fi->ive[k]] - x[_fj->ive[k]];
      f += 0.5 * _kf->ve[k] * e * e;
    }
    _f = f;
    for (int k = 0; k < _me; k++) {
      const int c = _cells->ive[k];
      _qp->b->ve[k] = x[c] * x[c + 1] + x[c + _gx] - _beta->ve[k];
    }
    for (int k = 0; k < _nb; k++) {
      const int i = _bnd->ive[k];
      _qp->d->ve[2 * k] = x[i] - _lo, _qp->d->ve[2 * k + 1] = _hi - x[i];
    }
  }
  void update(const VECP y, const VECP) {
    update_fbd();
    const double *x = _x->ve;
    double *g = _qp->c->ve;
    for (int i = 0; i < _n; i++) g[i] = _a->ve[i] * (x[i] - _t->ve[i]) + _gam * x[i] * x[i] * x[i];
    for (int i = 0; i < _n; i++) {
      if (i % _gx + 1 < _gx) {
        const double e = _kr->ve[i] * (x[i] - x[i + 1]);
        g[i] += e, g[i + 1] -= e;
      }
      if (i / _gx + 1 < _gy) {
        const double e = _kd->ve[i] * (x[i] - x[i + _gx]);
        g[i] += e, g[i + _gx] -= e;
      }
    }
    for (int k = 0; k < _nfar; k++) {
      const double e = _kf->ve[k] * (x[_fi->ive[k]] - x[_fj->ive[k]]);
      g[_fi->ive[k]] += e, g[_fj->ive[k]] -= e;
    }
    for (int k = 0; k < _me; k++) {
      const int c = _cells->ive[k];
      sp_set_val(_qp->A, k, c, x[c + 1]), sp_set_val(_qp->A, k, c + 1, x[c]);
    }
    if (_hela) {  // Hessian of L = f - y'g - z'h (the sign of Hqp_SqpSolver::grd_L, hqp/Hqp_SqpSolver.C:430-443)
      for (int i = 0; i < _n; i++) {
        double dii = _a->ve[i] + 3.0 * _gam * x[i] * x[i];
        if (i % _gx + 1 < _gx) dii += _kr->ve[i], sp_set_val(_qp->Q, i, i + 1, -_kr->ve[i]);
        if (i / _gx + 1 < _gy) dii += _kd->ve[i], sp_set_val(_qp->Q, i, i + _gx, -_kd->ve[i]);
        if (i % _gx > 0) dii += _kr->ve[i - 1];
        if (i / _gx > 0) dii += _kd->ve[i - _gx];
        sp_set_val(_qp->Q, i, i, dii);
      }
      for (int k = 0; k < _nfar; k++) {  // (several couplings may share an end: the diagonals are added up)
        const int i = _fi->ive[k], j = _fj->ive[k];
        sp_set_val(_qp->Q, i, j, -_kf->ve[k]);
        sp_set_val(_qp->Q, i, i, sp_get_val(_qp->Q, i, i) + _kf->ve[k]);
        sp_set_val(_qp->Q, j, j, sp_get_val(_qp->Q, j, j) + _kf->ve[k]);
      }
      if ((const VEC *)y)
        for (int k = 0; k < _me; k++) {
          const int c = _cells->ive[k];
          sp_set_val(_qp->Q, c, c + 1, sp_get_val(_qp->Q, c, c + 1) - y->ve[k]);
        }
    }
  }
};

extern "C" {

// out[0] objective prg_f, out[1] SQP iterations, out[2] sum of qp iterations, out[3] seconds
// of the solve, out[4] sqp_norm_inf, out[5] sqp_norm_grd_L.  Returns 0 (optimal), a Meschach
// error number, -1 setup, -2 unknown solver / plugin name, -3 qp solver made no iteration,
// -4 iteration limits, -5 stall.
int hqpsqp_did(int kmax, const char *qp_solver, const char *mat_solver, double sqp_eps, int sqp_max_iters,
               double *out) {
  if (hqpref_startup() != 0) return -1;
  int err = 0, rc = 0;
  double t0 = 0.0, t1 = 0.0;
  int qp_iters = 0;
  m_catchall(
      // as Hqp_Init does (hqp/Hqp_Init.C:199-204) and Docp_Main.C:37
      theSqpProgram = NULL; theSqpSolver = new Hqp_SqpPowell; Prg_DID *prg = new Prg_DID(); theSqpProgram = prg;
      if (If_SetString("sqp_qp_solver", qp_solver) != IF_OK) rc = -2;
      if (!rc && If_SetString("qp_mat_solver", mat_solver) != IF_OK) rc = -2;
      if (!rc) {
        (void)If_SetInt("prg_kmax", kmax);
        (void)If_SetReal("sqp_eps", sqp_eps);
        (void)If_SetInt("sqp_max_iters", sqp_max_iters);
        // Docp_Main.C:66-68
        if (If_Eval("prg_setup") != IF_OK || If_Eval("prg_simulate") != IF_OK || If_Eval("sqp_init") != IF_OK) rc = -1;
      }
      t0 = now_s();
      rc = rc ? rc : hqp_solve_loop(sqp_eps, qp_iters);
      t1 = now_s();
      out[0] = getr("prg_f"); out[1] = geti("sqp_iter"); out[2] = qp_iters; out[3] = t1 - t0;
      out[4] = getr("sqp_norm_inf"); out[5] = getr("sqp_norm_grd_L");
      delete prg; delete theSqpSolver; theSqpSolver = NULL; theSqpProgram = NULL,
      err = _err_num);
  return err ? err : rc;
}

// BASELINE.json configs[4] stand-in (Prg_GridNLP above) through the reference's Hqp_SqpPowell.  hela 1: analytic
// Lagrangian Hessian + sqp_hela Gerschgorin; 0: sqp_hela DScale (hqp_cute/hqp_cute.tcl:36-42).  ordering is
// handed to mat_ordering where the plugin has it (ours).  out as hqpsqp_did plus out[6..8] = n, me, m.
int hqpsqp_gridfar(int gx, int gy, int far, int seed, int eq_every, double bound_frac, int hela, const char *qp_solver,
                   const char *mat_solver, int ordering, double sqp_eps, int sqp_max_iters, double *out);
int hqpsqp_grid(int gx, int gy, int seed, int eq_every, double bound_frac, int hela, const char *qp_solver,
                const char *mat_solver, int ordering, double sqp_eps, int sqp_max_iters, double *out) {
  return hqpsqp_gridfar(gx, gy, 0, seed, eq_every, bound_frac, hela, qp_solver, mat_solver, ordering, sqp_eps, sqp_max_iters, out);
}
// ... with `far` couplings between distant cells (Prg_GridNLP above)
int hqpsqp_gridfar(int gx, int gy, int far, int seed, int eq_every, double bound_frac, int hela, const char *qp_solver,
                   const char *mat_solver, int ordering, double sqp_eps, int sqp_max_iters, double *out) {
  if (hqpref_startup() != 0) return -1;
  int err = 0, rc = 0;
  double t0 = 0.0, t1 = 0.0;
  int qp_iters = 0;
  m_catchall(
      theSqpProgram = NULL; theSqpSolver = new Hqp_SqpPowell;
      Prg_GridNLP *prg = new Prg_GridNLP(gx, gy, seed, eq_every, bound_frac, hela, far); theSqpProgram = prg;
      if (If_SetString("sqp_qp_solver", qp_solver) != IF_OK) rc = -2;
      if (!rc && If_SetString("qp_mat_solver", mat_solver) != IF_OK) rc = -2;
      if (!rc) {
        if (ordering) (void)If_SetInt("mat_ordering", ordering);
        (void)If_SetString("sqp_hela", hela ? "Gerschgorin" : "DScale");
        (void)If_SetReal("sqp_eps", sqp_eps);
        (void)If_SetInt("sqp_max_iters", sqp_max_iters);
        (void)If_SetInt("qp_max_iters", 999);
        prg->setup();
        if (If_Eval("sqp_init") != IF_OK) rc = -1;
      }
      out[6] = prg->n(); out[7] = prg->me(); out[8] = prg->m();
      t0 = now_s();
      rc = rc ? rc : hqp_solve_loop(sqp_eps, qp_iters);
      t1 = now_s();
      out[0] = getr("prg_f"); out[1] = geti("sqp_iter"); out[2] = qp_iters; out[3] = t1 - t0;
      out[4] = getr("sqp_norm_inf"); out[5] = getr("sqp_norm_grd_L");
      delete prg; delete theSqpSolver; theSqpSolver = NULL; theSqpProgram = NULL,
      err = _err_num);
  return err ? err : rc;
}

}  // extern "C"
