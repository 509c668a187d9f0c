/*
 * hqpkkt.h -- C ABI of the MI355X-native interior-point KKT linear-system path.
 *
 * This is the drop-in boundary for the reference's Hqp_IpMatrix plugin point
 * (hqp/Hqp_IpMatrix.h:63-88).  A thin C++ subclass of Hqp_IpMatrix (see
 * shim/Hqp_IpSpBKPHip.C and INTEGRATION.md) forwards its virtual methods to
 * these entry points exactly like the reference's own Hqp_IpPARDISO forwards to
 * a dlopen'ed C function (hqp/pardiso_wrapper.h:33-48,
 * hqp/Hqp_IpPARDISO.C:156-222).
 *
 * Conventions
 *  - plain pointers and sizes only; int32 indices, fp64 values; 0-based CSR
 *    with column indices sorted inside each row (as Meschach's SPROW keeps
 *    them, meschach/sparse.h:44-63);
 *  - every function returns a status: 0 = ok, HQPKKT_E_SING (= Meschach's
 *    E_SING, meschach/err.h:88) for a numerically singular system or a zero
 *    z/w component (meschach/vecop.c:346-348), other HQPKKT_E_* otherwise.
 *    Nothing throws, longjmps or aborts across this boundary; the shim turns a
 *    non-zero status into m_error(...) (meschach/err.h:63);
 *  - calls are synchronous from the caller's point of view; a handle is not
 *    re-entrant (the reference drives the plugin from one thread);
 *  - there is NO CPU fallback: numeric entry points need a gfx950 device and
 *    fail with HQPKKT_E_DEVICE without one.  hqpkkt_analyze is host-only.
 */
#ifndef HQPKKT_H
#define HQPKKT_H

#ifdef __cplusplus
extern "C" {
#endif

#define HQPKKT_VERSION 1

/* status codes (values 1..17 mirror meschach/err.h:84-110 where they exist) */
#define HQPKKT_OK 0
#define HQPKKT_E_SIZES 1   /* E_SIZES  */
#define HQPKKT_E_MEM 3     /* E_MEM    */
#define HQPKKT_E_SING 4    /* E_SING   */
#define HQPKKT_E_FORMAT 6  /* E_FORMAT: unsorted / out-of-range CSR */
#define HQPKKT_E_NULL 8    /* E_NULL   */
#define HQPKKT_E_RANGE 10  /* E_RANGE  */
#define HQPKKT_E_INTERN 17 /* E_INTERN: call order (e.g. factor before analyze) */
#define HQPKKT_E_DEVICE 100 /* HIP runtime error / no gfx950 device */

/* which reference plugin's semantics the handle reproduces */
#define HQPKKT_MODE_FULL 0    /* Hqp_IpSpBKP    (hqp/Hqp_IpSpBKP.C:76-218)    */
#define HQPKKT_MODE_REDUCED 1 /* Hqp_IpRedSpBKP (hqp/Hqp_IpRedSpBKP.C:184-368) */
#define HQPKKT_MODE_STAGED 2  /* Hqp_IpLQDOCP    (hqp/Hqp_IpLQDOCP.C:693-976): multistage (DOCP)
                                 structure, dense per-stage blocks, see hqpkkt_set_stages */

/* where the vectors z,w,r1..r4,dx..dw (and Qx,Ax,Cx) live */
#define HQPKKT_LOC_HOST 0   /* Meschach VEC::ve pointers; copied H2D / D2H per call */
#define HQPKKT_LOC_DEVICE 1 /* device pointers on opts.device, used in place */
/* Device vectors are read and written by the handle's own (non-blocking) stream: whatever the caller has queued on
 * other streams for these vectors - producers of the inputs, but also writes to the OUTPUT vectors such as clearing
 * them - must be complete before a call; every call returns with its results complete.  hqpkkt_factor / hqpkkt_solve
 * stage the vectors through buffers of the handle on the first call; from the second call in a row with the same
 * pointers (none of them overlapping another) they work on the caller's vectors themselves. */

typedef struct hqpkkt hqpkkt_t;

typedef struct hqpkkt_opts {
  int mode;          /* HQPKKT_MODE_*                                         */
  int device;        /* HIP device ordinal                                    */
  int loc;           /* HQPKKT_LOC_*                                          */
  double tol;        /* mat_tol of the reference (hqp/Hqp_IpSpBKP.C:46,59):
                        Bunch-Kaufman alpha = tol*(1+sqrt 17)/8, 0 < tol <= 1  */
  double eps;        /* mat_eps (hqp/Hqp_IpMatrix.C:45-47): refinement target */
  double pivot_eps;  /* static pivot perturbation, relative to max|K_ij|      */
  int leaf_size;     /* nested-dissection leaf size in rows (0 = default)     */
  int max_pivots;    /* max pivots per supernode, <= 192 (0 = default: 160, and 192 in
                        the chains of separators of >= 768 vertices)            */
  int zd_policy;     /* placement of variables with a structurally zero diagonal
                        (equality multipliers): 2 = behind all their neighbours - the
                        pivot is the complete Schur complement A H^-1 A', all pivots 1x1
                        when the Hessian diagonal is strong (fast); 0 = right behind one
                        matched neighbour, so that a 2x2 pivot with it is available inside
                        the pivot block - what QPs with weak Hessian diagonals need (a
                        state with Q_ii = 1e-4 coupled by 1.0 to a multiplier in an
                        ancestor supernode is otherwise eliminated with a multiplier of
                        1e4: in the last iterations of Prg_DID the residual of a solve is
                        1e-1 instead of 1e-15).  -1 (default) = 2, and where the values say
                        that 0 may be needed (some x with |Q_ii| < 0.01 max_r |A_ri|) the
                        first hqpkkt_solve whose refinement does not reach mat_eps switches
                        the handle to 0 (symbolic phase, upload, values and factorisation
                        once more; mat_sbw does not change) and repeats itself             */
  int slack_policy;  /* FULL mode, order of the slack rows inside a supernode:
                        2 = a slack row in front of one of its own x variables is
                        moved right behind it (default: avoids the run-time
                        interchange the Bunch-Kaufman test makes when w/z is small;
                        ~3 % more fill), 0 = band order as it comes (better when
                        the x variables carry weak diagonals, e.g. DOCP states),
                        1 = behind all x variables of the node (~10 % more fill)  */
  int no_small_fronts; /* 1 = do not use the fused one-wavefront kernels for fronts with
                        <= 32 pivots and <= 16 border rows (tests: both paths must agree) */
  int upd_pingpong_mb; /* update blocks (b x b per supernode) beyond this many MB are not kept for
                        the whole factorisation: tree levels are re-assigned as late as possible
                        and the blocks of even / odd levels alternate between two half-arenas
                        (0 = default 16384, < 0 = never)                                      */
  int amalgamation;  /* 1 = a separator of the nested dissection absorbs its child separators while
                        the merged pivot set still fits a small front (<= 32 pivots, elimination
                        order unchanged).  For narrow bands (a handful of rows per separator) the
                        tree levels above the leaves shrink to a third, and a level costs launch
                        latency there, not arithmetic: +7..10 % interior-point iterations/s on
                        the Prg_DID structure, same iteration counts.  Default 0 (DESIGN.md
                        section 4)                                                              */
  int ordering;      /* elimination tree: 0 = nested dissection of the RCM band (default: banded and
                        multistage systems, mat_sbw wide separators on every level), 1 = nested
                        dissection of the graph itself by breadth-first level structures, for
                        irregular sparsity (discretised / CUTE-style programs, hqp_cute/hqp_cute.tcl:
                        22-46 selects RedSpBKP for them): separators shrink with the piece.  mat_sbw
                        and the RCM permutation are reported as before either way; 2 = as 1 without the
                        reference-faithful RCM pass (hqp/sprcm.C:226-384 re-sorts a level after every
                        parent: quadratic in the level width, seconds for a 10^6-node mesh): mat_sbw
                        and hqpkkt_get_perm then describe a plain reverse Cuthill-McKee numbering       */
} hqpkkt_opts;

typedef struct hqpkkt_stats {
  /* structure (valid after analyze) */
  int dim;               /* order of the factored matrix                       */
  int sbw;               /* mat_sbw: semi-bandwidth under the RCM order        */
  int n_supernodes;
  int n_levels;          /* height of the assembly tree                        */
  int max_front;         /* largest front order (pivots + border)              */
  long long nnz_kkt;     /* stored entries of the permuted upper KKT matrix    */
  long long nnz_factor;  /* entries of L kept (panels), incl. diagonal blocks  */
  long long flops_factor;/* flops of one numeric factorisation as implemented  */
  long long bytes_panels, bytes_updates; /* device arena sizes                 */
  /* last numeric calls (valid after factor / solve) */
  int n_2x2;             /* 2x2 pivots chosen in the last factor               */
  int n_perturbed;       /* pivots replaced by +-pivot_eps*max|K| in last factor */
  int refine_rounds;     /* refinement rounds of the last solve                */
  double kmax;           /* max |scaled K_ij| of the last factor               */
  /* device time of the last call of each phase, HIP events on the handle's
     stream, milliseconds */
  float ms_assemble, ms_factor, ms_step, ms_residual, ms_solve;
  /* one system sharded over several ranks (valid after analyze; hqpkkt_set_shard) */
  int shard_rank, shard_count;
  int n_top;              /* supernodes of the replicated top of the tree          */
  int n_exchange_blocks;  /* subtree roots whose update blocks are all-gathered    */
  long long flops_local, flops_top; /* factor flops of this rank's subtrees / top  */
  long long bytes_exchange_factor;  /* all-gather volume per factor (all slots)    */
  long long bytes_exchange_step;    /* all-gather + all-reduce volume per step     */
  int n_slow_pivots;      /* pivots of the last factor that failed the cheap test
                             |a_kk| >= alpha max|column| and took the complete
                             Bunch-Kaufman decision (k_factor_diag's slow path)   */
  int n_poll_fallbacks;   /* times a launch that spans tree levels gave up waiting for
                             another workgroup's words since the handle was created; it
                             has run on per-level launches from the first one on        */
} hqpkkt_stats;

/* Fill *opts with the defaults (mode FULL, device 0, host pointers, tol 1.0,
 * eps 1e-10 as the reference's constructors set them). */
int hqpkkt_default_opts(hqpkkt_opts *opts);

/* ctor / dtor of the plugin object (Hqp_IpSpBKP::Hqp_IpSpBKP, ~Hqp_IpSpBKP,
 * hqp/Hqp_IpSpBKP.C:43-73).  Host-only; no device is touched yet. */
int hqpkkt_create(const hqpkkt_opts *opts, hqpkkt_t **out);
int hqpkkt_destroy(hqpkkt_t *h);

/* Hqp_IpSpBKP::init / Hqp_IpRedSpBKP::init, structure part
 * (hqp/Hqp_IpSpBKP.C:76-114, hqp/Hqp_IpRedSpBKP.C:184-265): RCM ordering of the
 * KKT graph (hqp/sprcm.C:62-420), semi-bandwidth, supernode partition and
 * symbolic factorisation.  Q is n x n (only entries with col >= row are read,
 * meschach/addon2_hqp.c:1078-1086), A is me x n, C is m x n.  Host pointers
 * always.  Host-only: runs without a GPU.  *sbw receives mat_sbw. */
int hqpkkt_analyze(hqpkkt_t *h, int n, int me, int m,
                   const int *Qp, const int *Qi, const int *Ap, const int *Ai,
                   const int *Cp, const int *Ci, int *sbw);

/* Hqp_IpSpBKP::update / Hqp_IpRedSpBKP::update (hqp/Hqp_IpSpBKP.C:117-136,
 * hqp/Hqp_IpRedSpBKP.C:268-278): new values on the analysed pattern.  Pointers
 * per opts.loc.  First call uploads the symbolic structure to the device. */
int hqpkkt_set_values(hqpkkt_t *h, const double *Qx, const double *Ax,
                      const double *Cx);

/* Pinned host buffers of the handle for the three value arrays (after hqpkkt_analyze; nnz(Q), nnz(A),
 * nnz(C) doubles; they live until the next hqpkkt_analyze / hqpkkt_destroy).  A host whose matrices are
 * row lists (Meschach SPMAT: one heap array per row) writes the values of an update() straight into
 * them - several threads, no intermediate copy - and passes the same pointers to hqpkkt_set_values:
 * the transfer is then ONE DMA per block from page-locked memory (opts.loc = HQPKKT_LOC_HOST).  The
 * shim does this when the pattern is unchanged (the reference's PARDISO plugin re-walks its matrices per
 * update as well, hqp/Hqp_IpPARDISO.C:240-330). */
int hqpkkt_values_staging(hqpkkt_t *h, double **Qx, double **Ax, double **Cx);

/* Hqp_IpSpBKP::factor / Hqp_IpRedSpBKP::factor (hqp/Hqp_IpSpBKP.C:139-180,
 * hqp/Hqp_IpRedSpBKP.C:281-320) including spBKPfactor (hqp/spBKP.C:369-645):
 * insert w/z (resp. C'ZW^-1C), symmetric scaling, LDL' with 1x1/2x2 pivots.
 * HQPKKT_E_SING: an exactly zero pivot (hqp/spBKP.C:699-700, 731-732) in a root
 * front or on a variable without a diagonal of its own (equality multiplier, x
 * without Q_ii).  An exactly zero pivot elsewhere is perturbed (the pivot search
 * ends at the supernode, the rest of the column is still to come); hqpkkt_solve
 * then returns HQPKKT_E_SING if its refinement does not reach opts.eps. */
int hqpkkt_factor(hqpkkt_t *h, const double *z, const double *w);

/* Hqp_IpSpBKP::step / Hqp_IpRedSpBKP::step (hqp/Hqp_IpSpBKP.C:183-218,
 * hqp/Hqp_IpRedSpBKP.C:323-368) including spBKPsolve (hqp/spBKP.C:647-797):
 * one solve of  [-Q A' C' 0; A 0 0 0; C 0 0 -I; 0 0 W Z] d = r  with the
 * current factors, no refinement.  FULL mode: dw = C dx - r3 as the reference
 * computes it, except for active constraints (w_j < z_j), whose dw_j comes from
 * z_j dw_j + w_j dz_j = r4_j - the same number, to a relative accuracy. */
int hqpkkt_step(hqpkkt_t *h, const double *z, const double *w,
                const double *r1, const double *r2, const double *r3,
                const double *r4, double *dx, double *dy, double *dz,
                double *dw);

/* Hqp_IpMatrix::residuum (hqp/Hqp_IpMatrix.C:131-178): max inf-norm of the
 * four block residuals of the unreduced system; the residual vectors stay on
 * the device for the next refinement round. */
int hqpkkt_residual(hqpkkt_t *h, const double *z, const double *w,
                    const double *r1, const double *r2, const double *r3,
                    const double *r4, const double *dx, const double *dy,
                    const double *dz, const double *dw, double *res);

/* Hqp_IpMatrix::solve (hqp/Hqp_IpMatrix.C:65-128): step, then at most five
 * rounds of iterative refinement with the reference's back-off
 * (alpha = 1, .7, .4, .1) while the residual exceeds opts.eps.  *res is the
 * value the reference's solve() returns. */
int hqpkkt_solve(hqpkkt_t *h, const double *z, const double *w,
                 const double *r1, const double *r2, const double *r3,
                 const double *r4, double *dx, double *dy, double *dz,
                 double *dw, double *res);

/* mat_sbw (hqp/Hqp_IpSpBKP.C:58) and _QP2J (hqp/Hqp_IpSpBKP.C:91-93):
 * perm[qp_index] = position in the RCM order; dim entries. */
int hqpkkt_get_sbw(const hqpkkt_t *h, int *sbw);
int hqpkkt_get_perm(const hqpkkt_t *h, int *perm);

/* mat_tol / mat_eps setters (Tcl-visible members of the reference plugin,
 * hqp/Hqp_IpSpBKP.C:58-59, hqp/Hqp_IpMatrix.C:47) */
int hqpkkt_set_tol(hqpkkt_t *h, double tol);
int hqpkkt_set_eps(hqpkkt_t *h, double eps);

/* Run the handle's kernels on a caller-provided hipStream_t (NULL = the
 * handle's own stream). */
int hqpkkt_set_stream(hqpkkt_t *h, void *hip_stream);

int hqpkkt_get_stats(const hqpkkt_t *h, hqpkkt_stats *out);

/* ---- one system over several GPUs (nested-dissection / SPIKE cut
 * of the RCM band; the reference's spBKPfactor, hqp/spBKP.C:369-645, is
 * sequential and has no counterpart) ------------------------------------------
 * Every rank holds one handle on its own device and makes the same calls with
 * the same (replicated) arguments.  The symbolic phase splits the assembly tree:
 * the top (the outermost separators) is replicated, the subtrees below it are
 * dealt to the ranks.  Each factor needs ONE all-gather (the update blocks of the
 * subtree roots), each step one all-gather (their contribution vectors) and one
 * all-reduce (the solution in elimination order).  The library does not link a
 * communication library: it calls back into the host, after draining the
 * handle's stream, and continues when the callback returns -- the callback must
 * not return before the result is complete in `buf` (device memory on the
 * handle's device).  With torch.distributed (backend "nccl" = RCCL over xGMI)
 * this is hqp_amd.dist.make_exchange(); a C++ host passes a function that calls
 * ncclAllGather / ncclAllReduce on its communicator.
 *   op HQPKKT_XCHG_ALLGATHER:     buf holds nslots slots of slot_elems doubles,
 *                                 slot `rank` is filled; fill all of them.
 *   op HQPKKT_XCHG_ALLREDUCE_SUM: buf holds slot_elems doubles (nslots = 1);
 *                                 replace them by the sum over the ranks.
 * Returns 0 on success.  Call hqpkkt_set_shard before hqpkkt_analyze. */
#define HQPKKT_XCHG_ALLGATHER 0
#define HQPKKT_XCHG_ALLREDUCE_SUM 1
/* op HQPKKT_XCHG_BCAST_BASE + r (r = 0 .. nranks-1): buf holds slot_elems doubles that rank r has
 * filled (nslots = 1); bring them to every rank.  The STAGED engine gathers the strips of V_k this way
 * when their sizes differ much (one broadcast per rank, issued back to back: hqpkkt_rccl_exchange
 * puts them into one ncclGroup) instead of padding every strip to the largest. */
#define HQPKKT_XCHG_BCAST_BASE 16
typedef int (*hqpkkt_exchange_fn)(void *ctx, int op, double *buf, long long slot_elems, int nslots);
int hqpkkt_set_shard(hqpkkt_t *h, int rank, int count, hqpkkt_exchange_fn fn, void *ctx);
/* The stream-ordered form of the same hook: the callback puts the collective into `hip_stream` (the
 * handle's stream, behind the kernels that fill `buf`) and returns at once; nothing is drained, the
 * kernels that follow wait in the stream.  hqpkkt_rccl_exchange of libhqpkkt_rccl.so
 * (include/hqpkkt_rccl.h: ncclAllGather / ncclAllReduce of RCCL over xGMI) has this signature.
 * STAGED mode over several ranks: the state columns of every stage are cut into one range per rank
 * and the memory goes with them - a rank keeps its columns of every F_k (every rank is handed the
 * same blocks and copies its share) and its rows of every V_k: bytes_panels per rank <= 1 / P of the
 * single-rank figure + 10 %.  Per stage of a factorisation: W_p = V+ F_p (local), the gather of the
 * ranks' F blocks (static data, requested a stage ahead), the blocks of G_xx = F'V+F dealt out in a
 * ring and computed as W_p' F_q, ONE gather of those blocks (n^2/2 doubles in all) on the critical
 * path, V_k = G_xx - Y'Rm by every rank; the control-sized work is done by every rank on identical
 * data.  The solve gathers one state-sized vector per stage and direction.  Needs an even number of
 * states per stage. */
typedef int (*hqpkkt_exchange_stream_fn)(void *ctx, int op, double *buf, long long slot_elems, int nslots,
                                         void *hip_stream);
int hqpkkt_set_shard_stream(hqpkkt_t *h, int rank, int count, hqpkkt_exchange_stream_fn fn, void *ctx);

/* ---- STAGED mode (Hqp_IpLQDOCP, hqp/Hqp_IpLQDOCP.C) ---------------------------------
 * The QP of a discrete-time optimal control problem as Hqp_Docp::setup_qp lays it out
 * (hqp/Hqp_Docp.C:585-755): x = [x_0, u_0, x_1, u_1, ..., x_K]; the first rows of A are the
 * dynamics  fx_k x_k + fu_k u_k - x_{k+1}  (the -1.0 is the last entry of each row), the
 * other equality rows, all rows of C and all rows of Q stay inside one stage.  The engine
 * keeps fx, fu and the cost-to-go Hessians as dense blocks and runs the reference's extended
 * Riccati recursion (ExRiccatiFactorSc / ExRiccatiSolveSc, :1794-2182) as fp64 MFMA matrix
 * products over them; the equality constraints of a stage are eliminated with the controls
 * they determine and carried back to the previous stage otherwise (GE_QP's job,
 * meschach/addon_hqp.c:399-475), a fixed initial state is recognised as in
 * Check_Structure (:343-351).  hqpkkt_analyze finds the stage sizes from the staircase
 * of A exactly as Hqp_IpLQDOCP::Get_Dim does (:201-287) unless hqpkkt_set_stages has given
 * them (K stages, nx[K+1] states, nu[K] controls; K <= 0 returns to the detection);
 * HQPKKT_E_FORMAT: the pattern / the values are not such a staircase (the reference
 * asserts), HQPKKT_E_SIZES: a stage with more than 512 controls, more than 256 constraint rows
 * carried from one stage to the one before it, or a FREE initial state with more than 4096 components
 * + carried rows (a fixed x_0 has no limit).  Up to ~64 controls and ~130 for the order of a stage's
 * [G_uu N_u'; N_u 0] the control-sized work of a stage runs in the LDS of one CU; beyond that the same
 * elimination runs out of global memory (one workgroup: correct and slow, ~20 ms per stage at 300
 * controls).  mat_sbw is -1. */
int hqpkkt_set_stages(hqpkkt_t *h, int K, const int *nx, const int *nu);
/* The same with the dynamics handed over as DENSE blocks instead of CSR rows - what a DOCP of
 * 10^6 variables needs (K = 200 stages of 5000 states: the CSR form of fx alone would hold
 * 5*10^9 entries, beyond int32 row pointers; Hqp_IpLQDOCP::update extracts exactly these dense
 * blocks fx[k], fu[k] from A, hqp/Hqp_IpLQDOCP.C:748-755).  hqpkkt_analyze_staged replaces
 * hqpkkt_analyze: stage sizes as in hqpkkt_set_stages; n_total = number of variables the blocks Q, E,
 * C are built for (must equal nx[K] + sum of nx[k] + nu[k]: HQPKKT_E_SIZES otherwise); E (me_rest x n)
 * holds the equality rows other than the dynamics.  The vectors r2 / dy of factor / step / solve keep the reference's
 * row order: the sum of nx[1..K] dynamics rows first, then the me_rest rows of E.
 * hqpkkt_set_values_staged replaces hqpkkt_set_values: F[k] points to the row-major
 * nx[k+1] x (nx[k] + nu[k]) block [fx_k fu_k] with leading dimension ldF[k] (the -1.0 of the
 * staircase is implied); pointers per opts.loc (the array F itself is a host array).  The
 * blocks are copied: the caller may release them afterwards.  hqpkkt_mehrotra / _franke run on this
 * form too (their products with the dynamics rows go through the dense blocks; b / y in the same row
 * order as r2 / dy). */
/* Stage sizes from the staircase of the dynamics rows, for hosts that keep A as row lists and must never make a CSR
 * copy of the dynamics (the reference-side binding, shim/Hqp_IpSpBKPHip.C: 5*10^9 entries at K = 200, nx = 5000):
 * per row of A its length, the column of its last entry and of the one before it (three ints per row).  What
 * Hqp_IpLQDOCP::Get_Dim reads off the same rows (hqp/Hqp_IpLQDOCP.C:201-287).  nx holds cap + 1, nu cap entries;
 * returns K, nx[0..K], nu[0..K-1] and the number of dynamics rows (the first rows of A); HQPKKT_E_FORMAT: not a
 * staircase, HQPKKT_E_SIZES: more than cap stages.  Host-only, no handle, no device. */
int hqpkkt_detect_stages(int n, int rows, const int *row_len, const int *last_col, const int *prev_col, int cap, int *K,
                         int *nx, int *nu, int *dyn_rows);
int hqpkkt_analyze_staged(hqpkkt_t *h, int K, const int *nx, const int *nu, int n_total, int me_rest, int m, const int *Qp,
                          const int *Qi, const int *Ep, const int *Ei, const int *Cp, const int *Ci);
/* The dense blocks one at a time, for hosts that extract them from row lists stage by stage (two stage-sized pinned
 * buffers instead of K of them): hqpkkt_stage_staging returns pinned buffer `which` (0 / 1; large enough for the
 * largest block; it waits until the copy that last read the buffer is over), hqpkkt_set_stage_block copies block k =
 * [fx_k fu_k] (nx[k+1] x (nx[k] + nu[k]), row-major, leading dimension ldF; any pointer per opts.loc) into the engine's
 * arena, asynchronously in the handle's stream.  hqpkkt_set_values_staged with F = NULL then takes the other values
 * and ends the hand-over (HQPKKT_E_INTERN unless every block has been set since the analysis). */
int hqpkkt_stage_staging(hqpkkt_t *h, int which, double **buf, long long *elems);
int hqpkkt_set_stage_block(hqpkkt_t *h, int k, const double *F, long long ldF);
int hqpkkt_set_values_staged(hqpkkt_t *h, const double *Qx, const double *const *F, const long long *ldF,
                             const double *Ex, const double *Cx);
/* tests: rank and number of carried rows per stage (2 ints each, K+1 stages) of the last factor */
int hqpkkt_debug_stage_ranks(hqpkkt_t *h, int *out, int cap);

/* Micro-benchmark and self-check of the dense fp64 MFMA product the STAGED engine is made of:
 * C (M x N) = A'B for pseudo-random k-major operands (K x M, K x N), `reps` timed launches
 * (lower: only the tiles of the lower triangle, mirror: the upper one written from them).
 * *ms: average device time of a launch; *max_err: largest |C_ij - exact| / sum_k |a_ki b_kj|
 * over 4096 sampled entries. */
int hqpkkt_debug_dgemm(int device, int M, int N, int K, int lower, int mirror, int reps, double *ms, double *max_err);

/* Test hook, host only (no device needed): the work list of the cut form of that product (k_dgemm_tn_sk) for `tiles`
 * tiles of `nslab` k-slabs on `grid` workgroups - unequal shares for the two workgroups of a CU, sk_table.hpp.
 * units (or NULL): six ints per unit, (b * stride + i) * 6 for unit i of workgroup b: tile (-1: end of the list),
 * first and one-past-last k-slab, first parking slot of the tile, pieces of the tile, number of this piece.
 * Returns the stride (units per workgroup incl. the end mark), or 0 (no table for these sizes; cap_ints too small).
 * *pieces: parking slots; *whole_a / *whole_b: whole tiles per workgroup of the first / second half of the launch. */
int hqpkkt_debug_sk_table(long long tiles, int nslab, int grid, int *units, long long cap_ints, long long *pieces, int *whole_a, int *whole_b);

/* Per-kernel-class device timing for bench.py's roofline line: with on != 0
 * every kernel launch is bracketed by HIP events on the handle's stream and the
 * elapsed times are summed per class (hqpkkt_profile_class_name(c), c = 0..) at
 * the end of each call.  set_profile also zeroes the sums.  get_profile fills
 * up to n_classes entries and returns the number of classes that exist (as a
 * positive value, not a status). */
int hqpkkt_set_profile(hqpkkt_t *h, int on);
int hqpkkt_get_profile(const hqpkkt_t *h, int n_classes, double *ms, long long *launches);
const char *hqpkkt_profile_class_name(int c);
const char *hqpkkt_strerror(int status);

/* ---- device-resident interior-point loop ------------------
 * The reference's Mehrotra predictor-corrector solver (hqp/Hqp_IpsMehrotra.C:
 * cold_start :209-327, step :355-693, solve :696-735) with all vector work on the
 * device: per iteration one factor and two (rarely three) solves of this library plus
 * a handful of kernels over the CSR blocks; only the scalars that steer the iteration
 * come back to the host.  The handle must hold the QP's matrices (hqpkkt_analyze +
 * hqpkkt_set_values with Q, A, C of the Hqp_Program, hqp/Hqp_Program.h:43-60); c, b, d
 * and the outputs x, y, z, w follow opts.loc of the handle.  Cold start
 * (qp_init_method 0-3) or hot start from the handle's previous solve.  result uses the reference's Hqp_Result numbering
 * (hqp/Hqp_impl.h:37-43): 0 optimal, 3 suboptimal, 4 degenerate. */
typedef struct hqpkkt_ip_opts {
  double eps;       /* qp_eps (hqp/Hqp_Solver.C:53)                                   */
  int max_iters;    /* qp_max_iters (hqp/Hqp_Solver.C:52)                             */
  double gammaf;    /* step damping (hqp/Hqp_IpsMehrotra.C:95)                        */
  double norm_data; /* max inf-norm of Q, A, C, c, b, d (hqp/Hqp_IpsMehrotra.C:462-464);
                       the caller holds the data, 0 = use 1                            */
  int hot_start;    /* 0 = cold start (Hqp_IpsMehrotra::cold_start, :209-327);
                       1 = Hqp_IpsMehrotra::hot_start (:330-352) if the previous call on this handle
                       (same dimensions, hot_start != 0) left its x, y and the (z, w) of its last
                       iteration far from the solution (:475-478); as in Hqp_IpsMehrotra::solve
                       (:696-733) a hot start that does not reduce phi by 1.2 per iteration, takes a
                       step below 1e-5, runs max_warm_iters or does not end optimal is thrown away,
                       the QP is solved again from a cold start and its iterations are added to
                       iters; 2 = cold start, but keep what the next hot start needs           */
  int max_warm_iters; /* qp_max_warm_iters (hqp/Hqp_IpsMehrotra.C:111), 0 = 25                */
  int init_method;  /* qp_init_method of the cold start (hqp/Hqp_IpsMehrotra.C:226-250, 294-297):
                       0 z = w = 1, r4 = 0 (default); 1 w = max(|d|,1e-10) |Q| / |C|; 2 w =
                       |C| / max(|d|,1e-10) / |Q|; 3 as 0 with r4 = -z.*w and dz, dw added to z, w */
  int reserved[1];
  double norm_Q, norm_C, norm_d; /* inf-norms of Q, C (largest absolute row sum) and d: init_method 1, 2 */
  double qp_mu0;    /* hqpkkt_franke: qp_mu0 (hqp/Hqp_IpsFranke.C:77,87): > 0 chooses the cold start's Ltilde
                       from it (:167-173), 0 (default) "according Wright" (:175-182)                    */
} hqpkkt_ip_opts;
typedef struct hqpkkt_ip_result {
  int result, iters;     /* Hqp_Result, iterations                                    */
  int n_factor, n_solve; /* plugin calls made                                         */
  double gap, mu, phi, pcost, alpha; /* of the last iteration                         */
  float ms_total;        /* device time of the whole call                             */
  int attempts;          /* 1; 2: the first run ended degenerate / singular and the loop was run again, from a cold
                          * start, with static pivoting (cancelled multiplier pivots replaced): n_factor, n_solve
                          * and ms_total are then the totals over both runs, iters the count of the second (the
                          * field takes the place of the structure's tail padding: size and offsets are those of
                          * earlier builds) */
} hqpkkt_ip_result;
int hqpkkt_default_ip_opts(hqpkkt_ip_opts *opts);
int hqpkkt_mehrotra(hqpkkt_t *h, const hqpkkt_ip_opts *opts, const double *c, const double *b,
                    const double *d, double *x, double *y, double *z, double *w, hqpkkt_ip_result *res);

/* The reference's other interior-point solver, Hqp_IpsFranke (hqp/Hqp_IpsFranke.C: potential
 * reduction with the infeasibility measure zeta; cold_start :156-216, step :271-378, solve
 * :381-416), the default of Hqp_SqpSolver: per iteration one factor and one solve of this
 * library, whose returned residual is part of its optimality test (:372).  Same conventions as
 * hqpkkt_mehrotra; of hqpkkt_ip_opts it reads eps, max_iters, hot_start (1 = Hqp_IpsFranke::
 * hot_start, :222-266, from the iterate the previous hqpkkt_franke call on this handle ended with;
 * thrown away as in :388-411) and max_warm_iters (0 = 15); qp_beta 0.995 and qp_mu0 0 are the
 * reference's defaults (qp_mu0: hqpkkt_ip_opts.qp_mu0).  res->result: 0 optimal, 3 suboptimal, 4
 * degenerate, 1 feasible / 2 infeasible when max_iters ends the run. */
int hqpkkt_franke(hqpkkt_t *h, const hqpkkt_ip_opts *opts, const double *c, const double *b,
                  const double *d, double *x, double *y, double *z, double *w, hqpkkt_ip_result *res);

/* ---- introspection of the symbolic structure (host arrays; used by the
 * structure tests, not by the reference-side shim) ------------------------ */
/* what: 0 elim (QP index -> elimination index, dim), 1 node_piv_start,
 * 2 node_npiv, 3 node_nborder, 4 node_parent, 5 node_level (n_supernodes
 * each), 6 border_ptr (n_supernodes+1), 7 border_idx (border_ptr[last]),
 * 8 entry_row, 9 entry_col (elimination indices, nnz_kkt each), 10 node_owner
 * (rank per supernode, -1 = replicated top), 11 exchanged subtree roots; STAGED: 20 states
 * per stage, 21 controls, 22 first column, 23 / 24 own equality rows (ptr / rows), 25 rows
 * that fix x_0, 26 capacity of carried rows, 27 column cuts of the ranks ((K+1) x (ranks+1)), 28 two counters of
 * the last factorisation: stages whose K was inverted by the blocked elimination, and those of them that fell back to the
 * one-workgroup elimination (device -> host copy); 30 (zero-diagonal policy in use, last
 * values have weak Hessian diagonals), 31 (fronts of the tree's top that the solve handles in one launch, first
 * such level, LDS bytes of that launch).
 * *len receives the element count; out may be NULL to query it. */
int hqpkkt_debug_get(const hqpkkt_t *h, int what, int *out, long long *len);
/* diagnostics of the solve's fused top (k_solve_top): one solve on the vectors of the last one with time stamps inside
 * the launch; out (cap >= 8 x fronts doubles) receives per front its tree level and six times in microseconds: start,
 * static data in, children arrived, forward done, border solution arrived, backward done. */
int hqpkkt_debug_solve_top_stamps(hqpkkt_t *h, double *out, int cap);

/* Numeric blocks of one supernode after factor (device -> host copy, tests only):
 * what 0 = panel ((p+b) x p, column-major: L11 below the diagonal, L21), 1 = the
 * explicit inverse of the unit lower L11 (p x p, column-major, diagonal blocks of
 * 16 complete), 2 = X = A21 P' L11^-T (b x p), 3 = update block (b x b).
 * *len receives the element count; out may be NULL to query it. */
int hqpkkt_debug_read(hqpkkt_t *h, int what, int node, double *out, long long cap, long long *len);

/* MFMA f64 16x16x4 layout self-test on the device: C = A * B for integer-valued
 * asymmetric 16x16x16 operands; *max_err is max |C - exact|. */
int hqpkkt_selftest_mfma(int device, double *max_err);

/* The pivot-block kernel on ONE dense symmetric p x p block (1 <= p <= 192; A row-major), without a tree:
 * what the tests check P A P' = L D L' and M L = I with, for every pivot count and for blocks that need
 * interchanges and 2x2 pivots (hqp/spBKP.C:392, 431-438, 471, 480 restricted to the block).
 * variant 0: k_factor_blk as the library launches it (8 wavefronts up to 128 pivots, 16 beyond);
 * 1: k_factor_diag of rounds 1-3 (p <= 128); 2: k_factor_blk with 16 wavefronts whatever p.
 * Out (each may be NULL): L (p x p column-major, unit lower factor below the diagonal), dinv (2 p: D^-1, for a
 * 2x2 pivot i11 i21 | i22 i21), ptype (0 / 1 / 2), lperm (row of A at every pivot position), W = L^-1 (p x p
 * column-major), counters (128 ints: status, 2x2 pivots, perturbed, slow steps, ...; instrumented builds: time stamps from [9] on), ms (average of `reps`
 * launches of one workgroup). */
int hqpkkt_debug_factor_block(int device, int p, const double *A, double tol, double pivot_eps, int variant,
                              int reps, double *L, double *dinv, int *ptype, int *lperm, double *W,
                              int *counters, double *ms);

#ifdef __cplusplus
}
#endif
#endif /* HQPKKT_H */
