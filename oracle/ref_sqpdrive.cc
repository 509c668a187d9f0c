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
      // ---- hqp_solve (hqp/hqp_solve.tcl:77-250), cold start
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
      t1 = now_s();
      out[0] = getr("prg_f"); out[1] = geti("sqp_iter"); out[2] = qp_iters; out[3] = t1 - t0;
      out[4] = getr("sqp_norm_inf"); out[5] = getr("sqp_norm_grd_L");
      delete prg; delete theSqpSolver; theSqpSolver = NULL; theSqpProgram = NULL,
      err = _err_num);
  return err ? err : rc;
}

}  // extern "C"
