/*
 * Hqp_IpsMehrotraHip.h -- Hqp_Solver plugin that runs the WHOLE interior-point
 * iteration of the reference's Hqp_IpsMehrotra (hqp/Hqp_IpsMehrotra.C) on the
 * device through hqpkkt_mehrotra (include/hqpkkt.h): x, y, z, w, the right-hand
 * sides and the steps stay in HBM, the host sees one call per QP.
 *
 * Reference-side binding, compiled against the reference's headers like
 * Hqp_IpSpBKPHip.C.  Registered with the solver factory (iftcl/If_Class.h:53-60):
 *     sqp_qp_solver MehrotraHip      (and FrankeHip, below)
 * and it owns an Hqp_IpMatrixHip plugin selected the usual way
 *     qp_mat_solver RedSpBKPHip      (default, as the reference defaults to RedSpBKP,
 *                                     hqp/Hqp_IpsMehrotra.C:92) | SpBKPHip | LQDOCPHip
 * cold_start() / hot_start() select how the next solve() begins (the hot start with the
 * reference's own fall-back to a cold start, hqp/Hqp_IpsMehrotra.C:696-733).
 * Difference to Hqp_IpsMehrotra: no qp_step (single iterations are not exposed).
 */
#ifndef Hqp_IpsMehrotraHip_H
#define Hqp_IpsMehrotraHip_H

#include "Hqp_Solver.h"

class Hqp_IpMatrix;

class Hqp_IpsMehrotraHip : public Hqp_Solver {
 protected:
  int _n, _me, _m;
  VEC *_w;
  Real _gap, _alpha, _gammaf;
  int _n_factor, _n_solve;  // plugin calls of the last solve (read-only for the user)
  Real _ms_total;           // device time of the last solve, milliseconds
  int _hot;                 // how the next solve() starts: hqpkkt_ip_opts.hot_start
  int _max_warm_iters;      // qp_max_warm_iters (hqp/Hqp_IpsMehrotra.C:111,122)
  int _init_method;         // qp_init_method (hqp/Hqp_IpsMehrotra.C:112,124)
  Real _mu0;                // FrankeHip: qp_mu0 (hqp/Hqp_IpsFranke.C:77,87)
  Hqp_IpMatrix *_matrix;

 public:
  Hqp_IpsMehrotraHip();
  ~Hqp_IpsMehrotraHip();

  void init();
  void update();
  void cold_start();
  void hot_start();
  void step();
  void solve();

  const char *name() { return "MehrotraHip"; }

 protected:
  void run(bool franke);
};

// sqp_qp_solver FrankeHip: the reference's Hqp_IpsFranke (hqp/Hqp_IpsFranke.C, the default
// of Hqp_SqpSolver) on the device through hqpkkt_franke; same members (qp_max_warm_iters 15,
// hqp/Hqp_IpsFranke.C:81)
class Hqp_IpsFrankeHip : public Hqp_IpsMehrotraHip {
 public:
  Hqp_IpsFrankeHip();
  void solve();
  const char *name() { return "FrankeHip"; }
};

#endif
