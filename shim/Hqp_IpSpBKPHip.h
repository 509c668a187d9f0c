/*
 * Hqp_IpSpBKPHip.h -- Hqp_IpMatrix plugins that run the KKT linear-system path
 * on an MI355X through the C ABI of include/hqpkkt.h.
 *
 * Reference-side binding: this file and Hqp_IpSpBKPHip.C are compiled INTO (or
 * next to) an HQP build, against the reference's own headers, exactly like the
 * reference's Hqp_IpPARDISO (hqp/Hqp_IpPARDISO.{h,C}) which forwards to a C
 * function.  Two classes are registered with the plugin factory
 * (iftcl/If_Class.h:53-60):
 *     qp_mat_solver SpBKPHip      -- semantics of Hqp_IpSpBKP    (full system)
 *     qp_mat_solver RedSpBKPHip   -- semantics of Hqp_IpRedSpBKP (reduced)
 *     qp_mat_solver LQDOCPHip     -- semantics of Hqp_IpLQDOCP   (multistage, STAGED engine)
 * Hqp_IpsMehrotra / Hqp_IpsFranke / Hqp_SqpSolver call them unchanged through
 * the Hqp_IpMatrix virtual interface (hqp/Hqp_IpMatrix.h:63-88).
 */
#ifndef Hqp_IpSpBKPHip_H
#define Hqp_IpSpBKPHip_H

#include "Hqp_IpMatrix.h"

struct hqpkkt;

class Hqp_IpMatrixHip : public Hqp_IpMatrix {
 protected:
  int _mode;          // HQPKKT_MODE_FULL / HQPKKT_MODE_REDUCED / HQPKKT_MODE_STAGED
  int _mode_used;     // the engine the current handle runs (STAGED falls back to FULL, see init())
  int _n, _me, _m;    // dimensions of the analysed program
  int _sbw;           // mat_sbw (read-only for the user, as in hqp/Hqp_IpSpBKP.C:58)
  Real _tol;          // mat_tol (hqp/Hqp_IpSpBKP.C:59)
  int _device;        // mat_device: HIP device ordinal
  int _refine;        // mat_device_refine: run Hqp_IpMatrix::solve's refinement on the GPU
  int _ngpu;          // mat_ngpu: ONE system over this many GPUs (one HQP process per GPU, RCCL)
  void *_rccl;        // communicator context of libhqpkkt_rccl.so (include/hqpkkt_rccl.h)
  void *_rccl_lib;    // its dlopen handle
  int _rank;
  int _ordering;      // mat_ordering: 0 nested dissection of the RCM band, 1 of the graph itself
  int _staged_min_front; // mat_staged_min_front: LQDOCPHip uses the STAGED engine from this stage width on
  int _update_threads; // mat_update_threads: host threads of update()'s walk over the row lists
  // LQDOCPHip (hqp/Hqp_IpLQDOCP.C:177-179 registers the same three): mat_wz_tol and mat_a_sparse are accepted and
  // not used - factor() says so once on stderr when either is set to a non-default value (the first selects the reference's other recursion, off by default: HUGE_VAL, :111, 850-853; the
  // second its sparse products with fx, fu, which have no counterpart on dense MFMA blocks), mat_logging > 0 prints
  // the stage structure and the engine chosen at init()
  Real _wz_tol;
  int _a_sparse, _logging;
  bool _told_ignored;
  struct hqpkkt *_h;
  // STAGED engine with the dynamics handed over as dense blocks (hqpkkt_analyze_staged): stage sizes, the number of
  // dynamics rows, CSR of the OTHER equality rows only - the dynamics rows of A are never copied into a CSR
  bool _dense;
  int _K, _ndyn;
  IVEC *_nx, *_nu;
  // CSR copies of the pattern the handle was analysed for (pattern-change
  // detection like hqp/Hqp_IpPARDISO.C:247-248,293-296) and value staging
  IVEC *_Qp, *_Qi, *_Ap, *_Ai, *_Cp, *_Ci;
  VEC *_Qx, *_Ax, *_Cx;

  void extract(const Hqp_Program *qp, bool &pattern_changed);
  int create_handle(int mode);
  int open(int mode);
  int open_dense(const Hqp_Program *qp);
  int dense_values(const Hqp_Program *qp);
  void check(int status, const char *where);

 public:
  Hqp_IpMatrixHip(int mode);
  ~Hqp_IpMatrixHip();

  // the C-ABI handle (for Hqp_IpsMehrotraHip, which runs the whole iteration on it)
  struct hqpkkt *handle() { return _h; }

  void init(const Hqp_Program *);
  void update(const Hqp_Program *);
  void factor(const Hqp_Program *, const VEC *z, const VEC *w);
  void step(const Hqp_Program *, const VEC *z, const VEC *w,
            const VEC *r1, const VEC *r2, const VEC *r3, const VEC *r4,
            VEC *dx, VEC *dy, VEC *dz, VEC *dw);
  // refinement loop on the device (same algorithm as hqp/Hqp_IpMatrix.C:65-128)
  Real solve(const Hqp_Program *, const VEC *z, const VEC *w,
             const VEC *r1, const VEC *r2, const VEC *r3, const VEC *r4,
             VEC *dx, VEC *dy, VEC *dz, VEC *dw);
};

class Hqp_IpSpBKPHip : public Hqp_IpMatrixHip {
 public:
  Hqp_IpSpBKPHip() : Hqp_IpMatrixHip(0) {}
  const char *name() { return "SpBKPHip"; }
};

class Hqp_IpRedSpBKPHip : public Hqp_IpMatrixHip {
 public:
  Hqp_IpRedSpBKPHip() : Hqp_IpMatrixHip(1) {}
  const char *name() { return "RedSpBKPHip"; }
};

// Drop-in for users that select the multistage plugin (hqp_docp/Docp_Main.C:42-49,
// odc/crane.tcl:58: qp_mat_solver LQDOCP).  Like Hqp_IpLQDOCP (hqp/Hqp_IpLQDOCP.C:693-976)
// it finds the stages from the -1.0 staircase of A, keeps fx, fu and the cost-to-go
// Hessians as dense per-stage blocks and runs the extended Riccati recursion over them
// (HQPKKT_MODE_STAGED: fp64 MFMA products on the device) - from mat_staged_min_front (default 800) rows per
// stage front on; narrower stages are faster through the tree engine.  Where the reference asserts
// (no DOCP structure) or a stage is beyond the STAGED kernels (> 512 controls, > 256
// carried constraint rows, a free x_0 of > 4096 components) the same KKT system goes to the full-system engine.
class Hqp_IpLQDOCPHip : public Hqp_IpMatrixHip {
 public:
  Hqp_IpLQDOCPHip() : Hqp_IpMatrixHip(2) {}
  const char *name() { return "LQDOCPHip"; }
};

#endif
