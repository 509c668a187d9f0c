/*
 * Hqp_IpsMehrotraHip.C -- see Hqp_IpsMehrotraHip.h.
 */
#include "Hqp_IpsMehrotraHip.h"

#include <If_Int.h>
#include <If_Real.h>
#include <If_Module.h>

#include "Hqp_Program.h"
#include "Hqp_IpSpBKPHip.h"
#include "hqpkkt.h"

IF_CLASS_DEFINE("MehrotraHip", Hqp_IpsMehrotraHip, Hqp_Solver);
IF_CLASS_DEFINE("FrankeHip", Hqp_IpsFrankeHip, Hqp_Solver);

//--------------------------------------------------------------------------
Hqp_IpsMehrotraHip::Hqp_IpsMehrotraHip()
{
  _n = _me = _m = 0;
  _w = VNULL;
  _gap = 0.0;
  _alpha = 1.0;
  _gammaf = 0.01;  // hqp/Hqp_IpsMehrotra.C:95
  _hot = 2;
  _max_warm_iters = 25;  // hqp/Hqp_IpsMehrotra.C:111
  _init_method = 0;      // hqp/Hqp_IpsMehrotra.C:112
  _mu0 = 0.0;            // hqp/Hqp_IpsFranke.C:77
  _n_factor = _n_solve = 0;
  _ms_total = 0.0;
  _matrix = new Hqp_IpRedSpBKPHip;

  // the Tcl-visible members of hqp/Hqp_IpsMehrotra.C:104-125 that apply here
  _ifList.append(new If_Real("qp_gap", &_gap));
  _ifList.append(new If_Real("qp_alpha", &_alpha));
  _ifList.append(new If_Int("qp_n_factor", &_n_factor));
  _ifList.append(new If_Int("qp_n_solve", &_n_solve));
  _ifList.append(new If_Real("qp_device_ms", &_ms_total));
  _ifList.append(new If_Int("qp_max_warm_iters", &_max_warm_iters));
  _ifList.append(new If_Int("qp_init_method", &_init_method));
  _ifList.append(new IF_MODULE("qp_mat_solver", &_matrix, Hqp_IpMatrix));
}

//--------------------------------------------------------------------------
Hqp_IpsMehrotraHip::~Hqp_IpsMehrotraHip()
{
  v_free(_w);
  delete _matrix;
}

//--------------------------------------------------------------------------
void Hqp_IpsMehrotraHip::init()
{
  _n = _qp->Q->n;
  _me = _qp->A->m;
  _m = _qp->C->m;
  _y = v_resize(_y, _me);
  _z = v_resize(_z, _m);
  _w = v_resize(_w, _m);
  _matrix->init(_qp);
}

//--------------------------------------------------------------------------
void Hqp_IpsMehrotraHip::update()
{
  _matrix->update(_qp);
}

//--------------------------------------------------------------------------
void Hqp_IpsMehrotraHip::cold_start()
{
  _iter = 0;
  _alpha = 1.0;
  _result = Hqp_Infeasible;
  _hot = 2;  // cold, but keep what a later hot start needs (hqp/Hqp_IpsMehrotra.C:318-319, 475-478)
}

// hqp/Hqp_IpsMehrotra.C:330-352: x, y of the last solve and the (z, w) kept on the device
void Hqp_IpsMehrotraHip::hot_start()
{
  _iter = 0;
  _alpha = 1.0;
  _result = Hqp_Infeasible;
  _hot = 1;
}

void Hqp_IpsMehrotraHip::step()
{
  m_error(E_INTERN, "Hqp_IpsMehrotraHip::step: single iterations run on the device, use qp_solve");
}

//--------------------------------------------------------------------------
void Hqp_IpsMehrotraHip::solve()
{
  run(false);
}

// hqp/Hqp_IpsFranke.C on the device (hqpkkt_franke; qp_beta 0.995, qp_mu0 as the reference's interface variable)
Hqp_IpsFrankeHip::Hqp_IpsFrankeHip()
{
  _max_warm_iters = 15;
  _ifList.append(new If_Real("qp_mu0", &_mu0));
}

void Hqp_IpsFrankeHip::solve()
{
  run(true);
}

void Hqp_IpsMehrotraHip::run(bool franke)
{
  Hqp_IpMatrixHip *mat = dynamic_cast<Hqp_IpMatrixHip *>(_matrix);
  if (!mat)
    m_error(E_INTERN, "Hqp_IpsMehrotraHip::solve: qp_mat_solver must be SpBKPHip, RedSpBKPHip or LQDOCPHip");

  hqpkkt_ip_opts opts;
  hqpkkt_ip_result res;
  hqpkkt_default_ip_opts(&opts);
  opts.eps = _eps;
  opts.max_iters = _max_iters;
  opts.gammaf = _gammaf;
  opts.hot_start = _hot;
  opts.max_warm_iters = _max_warm_iters;
  opts.init_method = _init_method;
  opts.qp_mu0 = franke ? _mu0 : 0.0;
  opts.norm_Q = sp_norm_inf(_qp->Q), opts.norm_C = sp_norm_inf(_qp->C), opts.norm_d = v_norm_inf(_qp->d);
  // hqp/Hqp_IpsMehrotra.C:462-464
  opts.norm_data = max(max(max(max(max(opts.norm_Q, sp_norm_inf(_qp->A)), opts.norm_C), v_norm_inf(_qp->c)),
                           v_norm_inf(_qp->b)), opts.norm_d);

  int status = (franke ? hqpkkt_franke : hqpkkt_mehrotra)(mat->handle(), &opts, _qp->c->ve, _qp->b->ve, _qp->d->ve,
                                                          _qp->x->ve, _y->ve, _z->ve, _w->ve, &res);
  if (status != HQPKKT_OK) {
    fprintf(stderr, "Hqp_IpsMehrotraHip::solve: %s\n", hqpkkt_strerror(status));
    m_error(status == HQPKKT_E_MEM ? E_MEM : E_INTERN, "Hqp_IpsMehrotraHip::solve");
  }
  _result = (Hqp_Result)res.result;
  _iter = res.iters;
  _gap = res.gap;
  _alpha = res.alpha;
  _n_factor = res.n_factor;
  _n_solve = res.n_solve;
  _ms_total = res.ms_total;
}
