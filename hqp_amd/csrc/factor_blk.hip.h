// Pivot block of a general front (33 .. 256 pivots): dense Bunch-Kaufman LDL' and the
// explicit inverse M = L11^-1, on the matrix pipe.  Included by hqpkkt.hip after
// kernels.hip.h (round 4; k_factor_diag of rounds 1-3 stays selectable for same-box
// comparisons).
//
// What it replaces: the pivot-block part of spBKPfactor (hqp/spBKP.C:369-645), same pivot
// rule (alpha = tol (1+sqrt 17)/8, test order of hqp/spBKP.C:392, 431-438, 471, 480,
// restricted to the supernode's pivot block), same outputs as k_factor_diag: L11 (unit
// lower, scaled) in the panel arena, D^-1 / pivot types / pivot order, M = L11^-1.
//
// Structure.  The p x p block is cut into 16 x 16 blocks.  One workgroup per front holds
// the LOWER block triangle in registers, block (I, J) as the four accumulator registers
// of a v_mfma_f64_16x16x4 (lane l, register q: row 16 I + (l & 15), column
// 16 J + (l >> 4) + 4 q - the transposed block in the C/D layout), dealt round-robin to
// the wavefronts from the last block row up, so that the live blocks of every wavefront
// are a prefix of its list at any time.  The elimination runs on the AUGMENTED matrix
// [A | I]: a column that has been eliminated holds the entries of M from then on, so a
// block row always works on (I + 1) blocks, the inverse is finished with the last pivot
// (no tail), and one kind of update serves both:
//     rows below  -=  L(rows, panel) * (the 16 pivot rows of [M | U])
// Per panel of 16 pivots:
//   1. the diagonal block goes to LDS; ONE wavefront eliminates it (16 lanes = 16 rows,
//      the pivot row reaches the other rows through the DPP row broadcast inside the
//      v_fmac_f64 itself - no readlane, no LDS round trip, no barrier per pivot), and
//      leaves U, N = L_kk^-1 and D^-1;
//   2. every block below the diagonal block: C' = N A' (4 MFMAs; C = unscaled column,
//      L = C D^-1), column maxima for the Bunch-Kaufman test; the pivot rows of M are
//      multiplied by N (4 MFMAs per block);
//   3. the test |d| >= alpha max|column| of all 16 pivots at once; the pivots in front of
//      the first failure are accepted;
//   4. every live block: 4 MFMAs with both operands from LDS (16 x ld images of the pivot
//      rows and of -L).
// A pivot that fails the test takes the slow step: the complete Bunch-Kaufman decision,
// symmetric interchange and 1x1 / 2x2 elimination element by element on the register
// blocks (vectors through LDS), then the panels go on behind it (a panel may start in the
// middle of a block).
#pragma once

namespace kktdev {

// leading dimension of the 16-row LDS images: = 16 mod 32 doubles, so that the four
// k-rows a wavefront reads with one ds_read_b64 fall into different halves of the banks
// (a compile-time constant of the two instantiations: 144 for blocks of up to 128 pivots, 208 up to 192 - the
// k-rows of an operand are immediate offsets of one address then)
__host__ __device__ inline int fb_ld_of(int p) { return p <= 128 ? 144 : 208; }
// Where k-row k of a 16-row operand image starts (round 6): rows 2 j and 2 j + 1 are skewed by j doubles.  The reads of
// the matrix products (ds_read_b64: a half-wave touches k-rows k and k + 1 with k even, 16 consecutive doubles of
// each) stay conflict-free - ld = 16 mod 32 keeps the pair in different halves of the banks, the pair shares its
// skew - and the TRANSPOSED stores (16 lanes = 16 k-rows of one column: the rows of M of the next panel, the pivot
// rows times N, N itself) spread over eight bank pairs instead of hitting ONE (16-way conflicts: they were the 52 %
// of profiles/r05_pmc_tree.txt).  An image takes 16 ld + 8 doubles.
#define FBROW(k) ((k) * ld + ((k) >> 1))
__host__ __device__ inline int fb_img_of(int ld) { return 16 * ld + 8; }
__host__ __device__ inline size_t fb_lds_bytes(int p) {
  const int pp = ((p + 15) / 16) * 16, ld = fb_ld_of(p);
  // Op, Lb, Yb, Xq | Tb, Ld (two each), Gb | Ab, G2 | dvals, dinvs, cmaxf, flags (two each) | rm0 | dv | lp, pt (ints)
  return sizeof(double) * ((size_t)4 * fb_img_of(ld) + 5 * 272 + 512 + 96 + (size_t)pp + 2 * (size_t)pp + (size_t)pp + 16);
}

// Barrier of the panel loop: waits for this wavefront's LDS operations only.  __syncthreads() also waits for the
// global stores in flight (the finished columns of L11 and rows of M leave all the time; nobody in the
// workgroup reads them back, except the slow step, which uses __syncthreads()): microseconds per barrier.
__device__ __forceinline__ void fb_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// max over the 16 lanes of a DPP row of a non-negative float; every lane of the row ends with it
__device__ __forceinline__ float row16_max_f(float v) {
  v = fmaxf(v, dpp_move_f<0xb1, 0xf>(v));
  v = fmaxf(v, dpp_move_f<0x4e, 0xf>(v));
  v = fmaxf(v, dpp_move_f<0x124, 0xf>(v));
  v = fmaxf(v, dpp_move_f<0x128, 0xf>(v));
  return v;
}
// max over the 16 lanes of a DPP row of an unsigned (the bits of a non-negative float: a NaN is the largest);
// the DPP operand inside the max itself (the nops: a DPP read behind a VALU write of the same register)
__device__ __forceinline__ unsigned row16_max_u(unsigned v) {
  asm volatile("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf"
               : "+v"(v));
  return v;
}
// max over the 16 lanes of a DPP row; every lane of the row ends with it
__device__ __forceinline__ double row16_max(double v) {
  v = fmax(v, dpp_move<0xb1, 0xf>(v));   // quad_perm [1,0,3,2]
  v = fmax(v, dpp_move<0x4e, 0xf>(v));   // quad_perm [2,3,0,1]
  v = fmax(v, dpp_move<0x124, 0xf>(v));  // row_ror 4
  v = fmax(v, dpp_move<0x128, 0xf>(v));  // row_ror 8
  return v;
}
// |v| with NaN mapped to +inf (fmax would drop a NaN)
__device__ __forceinline__ double abs_nan_inf(double v) {
  return v == v ? fabs(v) : __longlong_as_double(0x7ff0000000000000LL);
}
__device__ __forceinline__ void lds_max_pos(double *addr, double v) {
  atomicMax((unsigned long long *)addr, (unsigned long long)__double_as_longlong(v));
}

// g[c] += (lane S of this lane's row of 16).g[c] * nl : the rank-1 update of a Gaussian
// elimination step with the pivot row taken through the DPP row broadcast of the
// multiply-add itself
#define FB_FMAC(c, S, NL) \
  asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:" #S " row_mask:0xf bank_mask:0xf" : "+v"(g[c]) : "v"(NL))
// a quarter of the update of step SP that is still pending (columns != SP, SP+1), issued between
// the dependent operations of the next step's reciprocal
#define FB_PEND(SP, Q)                                              \
  if ((SP) >= 0) {                                                  \
    _Pragma("unroll") for (int c = (Q); c < 16; c += 4) {           \
      if (c != (SP) && c != (SP) + 1) FB_FMAC(c, SP, nlp);          \
    }                                                               \
  }
// One elimination step.  The chain is d -> 1/d (estimate and one cubic step) -> multiplier -> the next
// diagonal entry -> its broadcast; the other 14 columns of the rank-1 update wait until the next step and
// fill the latencies of its chain.  No branch: a step in front of `off` runs with multipliers 0 (a branch
// around asm statements that modify g[] costs a copy of all of g at its join).  (The DPP reads follow VALU
// writes inside asm statements the compiler cannot see into: the nops are the hazard's wait states.)
#define FB_STEP(S, SP)                                                                    \
  {                                                                                       \
    double x = __builtin_amdgcn_rcp(d);                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    FB_PEND(SP, 0)                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    const double e = fma(-d, x, 1.0);                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    FB_PEND(SP, 1)                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    const double e2 = fma(e, e, e);                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    FB_PEND(SP, 2)                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    const double di = fma(x, e2, x);                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    FB_PEND(SP, 3)                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    const bool act = (r > (S)) & ((S) >= off);                                            \
    const double nl = act ? -(g[S] * di) : 0.0;                                           \
    /* the test inside the block (hqp/spBKP.C:431-438): |d| >= alpha |column| <=> |multiplier| <= 1 / alpha */ \
    badm |= __any(!(fabs(nl) <= ialpha)) ? (1u << (S)) : 0u;                              \
    /* the pivot and its inverse as this step computed it (0 in front of `off`): d and di are the same in every lane */ \
    if (lane == 0) dvals[S] = d, dinvs[S] = (S) >= off ? di : 0.0;                         \
    if ((S) < 15) {                                                                       \
      asm volatile("s_nop 1" ::: "memory");                                               \
      FB_FMAC(((S) + 1) & 15, S, nl);                                                     \
      asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:" FB_STR(FB_NEXT(S)) " row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(g[((S) + 1) & 15])); \
    }                                                                                     \
    g[S] = act ? nl : g[S];                                                               \
    Lbk[(S)*LD + ((S) >> 1) + r] = nl;                                                    \
    nlp = nl;                                                                             \
  }
#define FB_NEXT(S) FB_NEXT_##S
#define FB_NEXT_0 1
#define FB_NEXT_1 2
#define FB_NEXT_2 3
#define FB_NEXT_3 4
#define FB_NEXT_4 5
#define FB_NEXT_5 6
#define FB_NEXT_6 7
#define FB_NEXT_7 8
#define FB_NEXT_8 9
#define FB_NEXT_9 10
#define FB_NEXT_10 11
#define FB_NEXT_11 12
#define FB_NEXT_12 13
#define FB_NEXT_13 14
#define FB_NEXT_14 15
#define FB_NEXT_15 15

// Gaussian elimination of the 16 x 16 diagonal block from pivot `off` on, without
// interchanges, by one wavefront: lane l works on row r = l & 15 (the four rows of 16
// lanes do the same work).  Gb: the block as it stands (row-major, ld 17; its lower
// triangle is used for the part that is still to be eliminated, columns < off hold
// entries of M).  Leaves: Tb[s][c] = row s after the elimination (c < s: N resp. the M
// columns < off, c == s: 1, c > s: U), identity rows for s < off; -L of the block in
// Lbk[s * LD + r]; the pivots and their inverses (0 for the pivots in front of `off`); the pivots that failed
// |d| >= alpha |column| against the rows of the block itself or |d| >= pert.  Tn (16 rows of stride LD): N alone,
// Lbk (likewise): -L, both where the update's operand images keep the columns of this block.
// In three pieces (round 6), so that the look-ahead can run the first steps of the NEXT block's elimination in front
// of a panel's middle barrier and the rest behind it, with the block in registers across the barrier.
#define FB_STR2(x) #x
#define FB_STR(x) FB_STR2(x)
#ifndef FB_SPEC_STEPS
#define FB_SPEC_STEPS 4  // steps of the next block's elimination in front of a panel's middle barrier
#endif
// (the block's state as separate arguments - g: this lane's row of the block, d: the pivot on its way through the lanes,
// nlp: the pending multiplier, badm: failed pivots - not as a structure: behind a structure hipcc turns the select chain
// over g in fb_elim_finish into an indexed load and keeps the whole block in scratch memory)
__device__ __forceinline__ void fb_elim_load(double (&g)[16], double &d, double &nlp, unsigned int &badm, const double *Gb, int off, int lane) {
  const int r = lane & 15;
#pragma unroll
  for (int c = 0; c < 16; c++) {
    const int hi = max(r, c), lo = min(r, c);
    g[c] = Gb[(c < off) ? r * 17 + c : hi * 17 + lo];
  }
  nlp = 0.0, badm = 0u;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(g[0]));
}
// The block on its way through the elimination, parked in Gb (row-major, all 16 columns of every row; the pending
// multipliers in the pad column) across a barrier, and taken up again in front of step FB_SPEC_STEPS
__device__ __forceinline__ void fb_elim_park(const double (&g)[16], double nlp, double *Gb, int lane) {
  const int r = lane & 15;
  if (lane < 16) {
#pragma unroll
    for (int c = 0; c < 16; c++) Gb[r * 17 + c] = g[c];
    Gb[r * 17 + 16] = nlp;
  }
}
__device__ __forceinline__ void fb_elim_resume(double (&g)[16], double &d, double &nlp, const double *Gb, int lane) {
  const int r = lane & 15;
#pragma unroll
  for (int c = 0; c < 16; c++) g[c] = Gb[r * 17 + c];
  nlp = Gb[r * 17 + 16];
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:" FB_STR(FB_SPEC_STEPS) " row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(g[FB_SPEC_STEPS]));
}
// steps S0 .. S1 - 1 (S0 = 0 or the step the call before ended with)
#define FB_STEP_IF(S, SP) \
  if constexpr (S0 <= (S) && (S) < S1) FB_STEP(S, SP)
template <int LD, int S0, int S1>
__device__ __forceinline__ void fb_elim_steps(double (&g)[16], double &d, double &nlp, unsigned int &badm, double *Lbk, double *dvals, double *dinvs,
                                              double alpha, int off, int lane) {
  const int r = lane & 15;
  const double ialpha = 1.0 / alpha;
  FB_STEP_IF(0, -1) FB_STEP_IF(1, 0) FB_STEP_IF(2, 1) FB_STEP_IF(3, 2) FB_STEP_IF(4, 3) FB_STEP_IF(5, 4) FB_STEP_IF(6, 5) FB_STEP_IF(7, 6)
  FB_STEP_IF(8, 7) FB_STEP_IF(9, 8) FB_STEP_IF(10, 9) FB_STEP_IF(11, 10) FB_STEP_IF(12, 11) FB_STEP_IF(13, 12) FB_STEP_IF(14, 13) FB_STEP_IF(15, 14)
  // (step 15 has no row below it: nothing pending)
}
template <int LD>
__device__ __forceinline__ void fb_elim_finish(double (&g)[16], unsigned int badm, double *Tb, double *Tn, double *dvals, double *dinvs, int *bad_in,
                                               double pert, int off, int lane) {
  const int r = lane & 15, grp = lane >> 4;
  // the pivots as the steps left them (read back instead of picked out of g by a chain of selects: hipcc turns that
  // chain into an indexed load and puts the whole block into scratch memory for it)
  const double myd = dvals[r];
  // A row that a NaN has reached is of no use to the update with the accepted pivots either (a zero pivot makes the
  // rows below it NaN, and 0 * NaN of the steps behind it every row above: the whole panel goes to the slow step then,
  // as it did when this test looked at the row's diagonal entry at the END of the steps).
  bool nan_row = false;
#pragma unroll
  for (int c = 0; c < 16; c++) nan_row |= !(g[c] == g[c]);
  badm |= (unsigned int)(__ballot(r >= off && (nan_row || !(fabs(myd) >= pert))) & 0xffffull);
  if (lane == 0) *bad_in = (int)badm;  // bit s: pivot s failed against a row of its own block, or is tiny
#pragma unroll
  for (int c = 0; c < 16; c++) {
    const double v = (c == r) ? 1.0 : (r < off ? 0.0 : g[c]);
    if ((c & 3) == grp) Tb[r * 17 + c] = v;
    // N alone (unit lower; the entries of M in front of `off`), as 16 k-rows of the operand images
    if ((c & 3) == grp) Tn[r * LD + (r >> 1) + c] = c < r ? v : (c == r ? 1.0 : 0.0);
  }
}
template <int LD>
__device__ __forceinline__ void fb_eliminate_block(const double *Gb, double *Tb, double *Tn, double *Lbk,
                                                   double *dvals, double *dinvs, int *bad_in, double alpha,
                                                   double pert, int off, int lane) {
  double g[16], d, nlp;
  unsigned int badm;
  fb_elim_load(g, d, nlp, badm, Gb, off, lane);
  fb_elim_steps<LD, 0, 16>(g, d, nlp, badm, Lbk, dvals, dinvs, alpha, off, lane);
  fb_elim_finish<LD>(g, badm, Tb, Tn, dvals, dinvs, bad_in, pert, off, lane);
}

#ifdef HQPKKT_STAMPS
// instrumented build: s_memtime of thread 0 at the phase boundaries of the first panels (counters[8 + slot])
#define FBSTAMP(slot)                                                                    \
  do {                                                                                   \
    if (blockIdx.x == 0 && threadIdx.x == 0 && (slot) < 54) counters[8 + (slot)] = (int)__builtin_amdgcn_s_memtime(); \
  } while (0)
// per-wavefront stamps of panel 3 (sixteen per wavefront) in a symbol of their own (hqpkkt_debug_fb_stamps)
__device__ int g_fb_stamps[16 * 16];
#define FBWSTAMP(j)                                                                      \
  do {                                                                                   \
    if (blockIdx.x == 0 && npan == 3 && (threadIdx.x & 63) == 0)                         \
      g_fb_stamps[16 * (threadIdx.x >> 6) + (j)] = (int)__builtin_amdgcn_s_memtime();    \
  } while (0)
#define FBWSTAMP2(j, value) do { } while (0)
#else
#define FBSTAMP(slot)
#define FBWSTAMP(j)
#endif

template <int NW, int NS, int LD, int WPE, bool OWNSIMD>
__global__ void __launch_bounds__(64 * NW, WPE)
k_factor_blk(DevTree T, const int *__restrict__ level_nodes, double *__restrict__ panel,
             double *__restrict__ dinv, int *__restrict__ ptype, int *__restrict__ lperm,
             const signed char *__restrict__ esign, double *__restrict__ linv,
             const long long *__restrict__ linv_off, double alpha, double pivot_eps,
             const unsigned long long *__restrict__ kmax_bits, int *__restrict__ counters,
             const double *__restrict__ upd) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  // The last wavefront eliminates the diagonal blocks and holds no block.  The fp64 multiply-adds of its chain and
  // the fp64 matrix products of the others run on the same units of a SIMD: the wavefronts that share its SIMD
  // (every fourth) hold no block either.
  constexpr int NT = 64 * NW, NWK = OWNSIMD ? NW - NW / 4 : NW - 1;
  constexpr int ld = LD;
  const int node = level_nodes[blockIdx.x];
  const int p = T.npiv[node], b = T.nbor[node];
  const long long F = p + b;
  const int Fi = p + b;
  const int e0 = T.piv_start[node];
  double *P = panel + T.panel_off[node];
  double *W = linv + linv_off[node];  // p x p column-major
  const int nb = (p + 15) >> 4, pp = nb << 4;
  double *Op = lds;                 // 16 x ld: the pivot rows of [M | U] (k-major)
  constexpr int IMG = 16 * ld + 8, Q4 = 4 * ld + 2;  // an image (FBROW: skewed k-rows); from k-row k to k-row k + 4
  double *Lb = Op + IMG;            // 16 x ld: -L of the panel (k-major)
  double *Yb = Lb + IMG;            // 16 x ld: the rows of M of the next panel's pivots as they stand (k-major)
  double *Xq = Yb + IMG;            // 16 x ld, columns 0..63: what the elimination of a diagonal block leaves as
                                    // operand images, for two panels in turn: -L of the block (columns 16 par ..),
                                    // N of the block (columns 32 + 16 par ..)
  double *Tb = Xq + IMG;            // two of 16 x 17: the eliminated diagonal block (rows of [N | U])
  double *Ldg = Tb + 2 * 272;       // two of 16 x 17: -L inside the diagonal block, [pivot][row]
  double *Gb = Ldg + 2 * 272;       // 16 x 17: the diagonal block to eliminate
  double *Ab = Gb + 272, *G2 = Ab + 256;  // register images of the two blocks the next elimination starts from
  double *dvals = G2 + 256, *dinvs = dvals + 32;  // two of 16 each
  float *cmaxf = (float *)(dinvs + 32);           // two of 16 column maxima (fp32 bits order like the values)
  int *badin = (int *)(dinvs + 48);               // two: pivots that failed inside their own block (bit mask)
  float *rm0 = (float *)(dinvs + 64);  // pp: largest entry of every row of the block as assembled (fp32)
  double *dv = dinvs + 64 + pp;     // 2 pp: inverse pivot data
  int *lp = (int *)(dv + 2 * pp);   // pp
  int *pt = lp + pp;                // pp
  int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_ge = wave == NW - 1;
  const bool is_worker = OWNSIMD ? (wave & 3) != 3 : !is_ge;
  const int wrank = OWNSIMD ? wave - (wave >> 2) : wave;  // rank among the wavefronts that hold blocks
  if (is_ge) __builtin_amdgcn_s_setprio(3);  // its chain of dependent operations is the critical path of a panel
  const int ln0 = lane & 15, lg0 = lane >> 4;
  int ln = ln0, lg = lg0;
  const double pert = fmax(pivot_eps * __longlong_as_double((long long)*kmax_bits), 1e-300);
  const int nslots = nb * (nb + 1) / 2;

  // ---- this wavefront's blocks: slot t = wrank + NWK s, block rows from the last one up.  Lane s of the
  // wavefront keeps (block row, block column) of slot s: which slots a step concerns is one ballot (a list of NS
  // pairs in scalar registers, compared slot by slot in every step of every panel, cost more instructions than
  // the work itself), the pair of a slot that takes part comes by v_readlane.
  int myI = -1, myJ = -1;
  {
    const int t = wrank + NWK * lane;
    if (is_worker && lane < NS && t < nslots) {
      int I = nb - 1, base = 0;
      while (t >= base + I + 1) base += I + 1, I--;
      myI = I, myJ = t - base;
    }
  }
#define SLOT_I(s) __builtin_amdgcn_readlane(myI, (s))
#define SLOT_J(s) __builtin_amdgcn_readlane(myJ, (s))
#define SLOT_IN(mask, s) (((mask) >> (s)) & 1u)
  const unsigned m_all = (unsigned)__ballot(myI >= 0);
  FBSTAMP(0);
  double4_t R[NS];
  // ---- load: lower triangle of the block (+ the mirror image inside the diagonal
  // blocks), identity behind row p, and the children's update blocks (slot order)
#pragma unroll
  for (int s = 0; s < NS; s++) {
    R[s] = double4_t{0.0, 0.0, 0.0, 0.0};
    if (!SLOT_IN(m_all, s)) continue;
    const int i = 16 * SLOT_I(s) + ln, c0 = 16 * SLOT_J(s) + lg;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int c = c0 + 4 * q;
      const int hi = max(i, c), lo = min(i, c);
      const double v = P[hi < p ? lo * Fi + hi : 0];  // (32-bit offsets: a front has fewer than 2^31 entries)
      R[s][q] = hi < p ? v : (hi == lo ? 1.0 : 0.0);
    }
  }
  // two children at a time: their index maps travel together, then their values (one round trip each instead of two)
  for (int cc0 = T.child_ptr[node], cc1 = T.child_ptr[node + 1], cc = cc0; cc < cc1; cc += 2) {
    const bool two = cc + 1 < cc1;
    const int chA = T.child_idx[cc], chB = T.child_idx[two ? cc + 1 : cc];
    const int bcA = T.nbor[chA], bcB = T.nbor[chB];
    const int *ivA = T.pinv + T.pinv_off[chA], *ivB = T.pinv + T.pinv_off[chB];
    const double *UA = upd + T.upd_off[chA], *UB = upd + T.upd_off[chB];
#pragma unroll
    for (int s = 0; s < NS; s++) {
      if (!SLOT_IN(m_all, s)) continue;
      const int i = 16 * SLOT_I(s) + ln, c0 = 16 * SLOT_J(s) + lg;
      const int ciA = i < p ? ivA[i] : -1, ciB = (i < p && two) ? ivB[i] : -1;
      int cjA[4], cjB[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int c = c0 + 4 * q;
        cjA[q] = c < p ? ivA[c] : -1, cjB[q] = (c < p && two) ? ivB[c] : -1;
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const bool okA = ciA >= 0 && cjA[q] >= 0, okB = ciB >= 0 && cjB[q] >= 0;
        const double vA = UA[okA ? min(ciA, cjA[q]) * bcA + max(ciA, cjA[q]) : 0];
        const double vB = UB[okB ? min(ciB, cjB[q]) * bcB + max(ciB, cjB[q]) : 0];
        R[s][q] = (R[s][q] + (okA ? vA : 0.0)) + (okB ? vB : 0.0);
      }
    }
  }
  for (int i = tid; i < pp; i += NT) lp[i] = i, pt[i] = 0, rm0[i] = 0.0f, dv[2 * i] = 0.0, dv[2 * i + 1] = 0.0;
  __syncthreads();
  // row maxima of the block as assembled (what a pivot of a row without a diagonal of its own is measured
  // against, SOFT_PIVOT_REL; fp32 is plenty for that): non-negative floats order like their bits
#pragma unroll
  for (int s = 0; s < NS; s++) {
    if (!SLOT_IN(m_all, s)) continue;
    const int I_ = SLOT_I(s), J_ = SLOT_J(s);
    unsigned vi = 0u;  // (bit patterns of the fp32 magnitudes: they order like the values)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const unsigned a = __float_as_uint((float)R[s][q]) & 0x7fffffffu;
      vi = max(vi, a);
      const unsigned vc = row16_max_u(a);
      if (ln == 0) atomicMax((unsigned int *)&rm0[16 * J_ + lg + 4 * q], vc);
    }
    atomicMax((unsigned int *)&rm0[16 * I_ + ln], vi);
  }
  __syncthreads();

  // Block row `row`: its rows of M (as they stand) to Yb; diag: its diagonal block as rows to Gb (the
  // elimination reads it next)
  auto publish_block_row = [&](int row, bool diag) {
    const unsigned m = (unsigned)__ballot(myI == row && (diag || myJ != row));
#pragma unroll
    for (int s = 0; s < NS; s++) {
      if (!SLOT_IN(m, s)) continue;
      const int J_ = SLOT_J(s);
      if (J_ == row) {
#pragma unroll
        for (int q = 0; q < 4; q++) Gb[ln * 17 + lg + 4 * q] = R[s][q];
      } else {
        double *y = Yb + (FBROW(ln) + 16 * J_ + lg);
#pragma unroll
        for (int q = 0; q < 4; q++) y[4 * q] = R[s][q];
      }
    }
  };
  // the two blocks of block row `row` the elimination wavefront starts its next block from (register images)
  auto publish_start_blocks = [&](int row) {
    const unsigned m = (unsigned)__ballot(myI == row && myJ >= row - 1);
#pragma unroll
    for (int s = 0; s < NS; s++) {
      if (!SLOT_IN(m, s)) continue;
      double *d = SLOT_J(s) == row ? G2 : Ab;
#pragma unroll
      for (int q = 0; q < 4; q++) d[64 * q + lane] = R[s][q];
    }
  };

  int k = 0, par = 0;
  // hot: the diagonal block of the panel is eliminated already (by the look-ahead of the panel before) and the
  // start blocks of the next block row are published; false for the first panel and after a slow step
  bool hot = false;
  [[maybe_unused]] int npan = 0;
  FBSTAMP(1);
  while (k < p) {
    k = __builtin_amdgcn_readfirstlane(k);
    // (opaque copies per trip: otherwise every index and address below that depends only on the lane or on the
    // block list is computed in front of the loop and kept - hundreds of values, spilt)
    ln = ln0, lg = lg0;
    asm volatile("" : "+v"(ln), "+v"(lg), "+v"(tid), "+v"(lane));
    asm volatile("" : "+v"(myI), "+v"(myJ));
    const int kb = k >> 4, off = k & 15, kend = min(16, p - 16 * kb);
    // which slots the steps of this panel concern
    const unsigned m_solve = (unsigned)__ballot(myJ == kb && myI > kb);            // blocks below the diagonal block
    const unsigned m_mrow = (unsigned)__ballot(myI == kb && myJ < kb && myJ >= 0);  // the pivot rows of M
    const unsigned m_live = (unsigned)__ballot(myI >= kb);
    // ================= panel: pivots off .. 15 of block kb without interchanges =============
    if (!hot) {
      publish_block_row(kb, true);
      if (kb + 1 < nb) publish_start_blocks(kb + 1);
      fb_barrier();
      if (is_ge) {
        if (lane < 16) cmaxf[16 * par + lane] = 0.0f;
        fb_eliminate_block<LD>(Gb, Tb + 272 * par, Xq + 32 + 16 * par, Xq + 16 * par, dvals + 16 * par,
                               dinvs + 16 * par, badin + par, alpha, pert, off, lane);
        // (this wavefront holds no block: telling the compiler so frees the registers of R for the elimination)
#pragma unroll
        for (int s = 0; s < NS; s++) R[s] = double4_t{0.0, 0.0, 0.0, 0.0};
      }
      fb_barrier();
    }
    FBSTAMP(2 + 5 * npan);
    FBWSTAMP(0);
    const double *Tp = Tb + 272 * par, *dvp = dvals + 16 * par, *dip = dinvs + 16 * par;
    const double *Tnq = Xq + 32 + 16 * par, *Ldq = Xq + 16 * par;
    float *cmp = cmaxf + 16 * par;
    // ---- C' = N A' for the blocks below the diagonal block, M rows <- N (M rows); the elimination wavefront:
    // the next diagonal block as this panel leaves it.  N (row ln, columns lg + 4 q: the A operand of the first
    // product, the B operand of the second) and D^-1 once per wavefront.
    unsigned int spec_badm = 0u;  // (of the next diagonal block's first elimination steps, in front of the middle barrier)
    if (is_ge || (m_solve | m_mrow)) {
      double tn[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int c = lg + 4 * q;
        const double t = Tp[ln * 17 + c];
        tn[q] = (c <= ln && (c >= off || c == ln)) ? t : 0.0;  // (rows in front of `off` are rows of the identity)
      }
      const bool rowon = ln >= off;  // (as the A operand of C' = N A': no pivot in front of `off`)
      FBWSTAMP(1);
      if (is_ge) {
        if (kb + 1 < nb) {
          // Ahead of the test: with the whole panel accepted the block (kb+1, kb+1) becomes G - L C' with
          // C' = N (block (kb+1, kb))', both in registers as the accumulator layout of the products wants them
          double4_t A_, G_;
          double dvi[4];
#pragma unroll
          for (int q = 0; q < 4; q++) A_[q] = Ab[64 * q + lane], G_[q] = G2[64 * q + lane], dvi[q] = dip[lg + 4 * q];
          double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int q = 0; q < 4; q++) acc = mfma_f64(rowon ? tn[q] : 0.0, A_[q], acc);
#pragma unroll
          for (int q = 0; q < 4; q++) G_ = mfma_f64(acc[q], -(acc[q] * dvi[q]), G_);
#pragma unroll
          for (int q = 0; q < 4; q++) Gb[ln * 17 + lg + 4 * q] = G_[q];
          // ... and the first steps of its elimination at once, in front of the panel's middle barrier (speculative: the
          // test behind the barrier may still reject a pivot of THIS panel - the block in registers is dropped then,
          // what the steps wrote lies in the other parity's buffers, which nobody reads before they are written again).
          // The elimination wavefront used to idle through the solve phase and then be the last to reach the
          // panel's end: 8560 of a panel's 13 400 cycles (profiles/r06_fb_panel_stamps.txt).
#ifndef FB_NO_SPEC
          if (lane < 16) cmaxf[16 * (par ^ 1) + lane] = 0.0f;
          double eg[16], ed, enlp;
          fb_elim_load(eg, ed, enlp, spec_badm, Gb, 0, lane);
          fb_elim_steps<LD, 0, FB_SPEC_STEPS>(eg, ed, enlp, spec_badm, Xq + 16 * (par ^ 1), dvals + 16 * (par ^ 1), dinvs + 16 * (par ^ 1), alpha, 0, lane);
          fb_elim_park(eg, enlp, Gb, lane);  // (across the barrier in LDS: in registers it would be live in every wavefront's allocation)
#endif
        }
      } else {
#pragma unroll
        for (int s = 0; s < NS; s++) {
          if (SLOT_IN(m_solve, s)) {
            double dvi[4];  // D^-1 (0 in front of `off`)
#pragma unroll
            for (int q = 0; q < 4; q++) dvi[q] = dip[lg + 4 * q];
            double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; q++) acc = mfma_f64(rowon ? tn[q] : 0.0, R[s][q], acc);
            const int row = 16 * SLOT_I(s) + ln;
            double *o = Op + (FBROW(lg) + row), *l = Lb + (FBROW(lg) + row);
#pragma unroll
            for (int q = 0; q < 4; q++) {
              // acc[q] = C(row ln of block I, pivot lg + 4 q); zero for the pivots in front of `off`
              o[q * Q4] = acc[q];
              l[q * Q4] = -(acc[q] * dvi[q]);
              // column maxima in fp32, compared as bit patterns (a NaN is the largest)
              const unsigned v = row16_max_u(__float_as_uint((float)acc[q]) & 0x7fffffffu);
              if (ln == 0) atomicMax((unsigned int *)&cmp[lg + 4 * q], v);
            }
          } else if (SLOT_IN(m_mrow, s)) {
            const int J_ = SLOT_J(s);
            const double *y = Yb + (FBROW(lg) + 16 * J_ + ln);
            double a[4];
#pragma unroll
            for (int q = 0; q < 4; q++) a[q] = y[q * Q4];  // A operand: m = ln (column of block j), k = lg + 4 q
            double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; q++) acc = mfma_f64(a[q], tn[q], acc);
            double *o = Op + (FBROW(ln) + 16 * J_ + lg);
#pragma unroll
            for (int q = 0; q < 4; q++) o[4 * q] = acc[q];
          }
        }
      }
    }
    FBWSTAMP(2);
    fb_barrier();
    FBWSTAMP(3);
    FBSTAMP(3 + 5 * npan);
    // ---- the test of all pivots of the panel: the pivots in front of the first failure are accepted
    int done;
    {
      const int sl = lane & 15;
      const bool ok = fabs(dvp[sl]) >= alpha * (double)cmp[sl];  // against the rows below the block
      unsigned int badm = (unsigned int)(__ballot(!ok && sl >= off && sl < kend) & 0xffffull);
      badm |= (unsigned int)badin[par] & 0xffffu & (0xffffu << off) & ((1u << kend) - 1u);
      done = badm ? (int)__builtin_ctz(badm) : kend;
      done = __builtin_amdgcn_readfirstlane(done);
    }
    FBWSTAMP(4);
    // Look-ahead: with the whole block accepted the elimination of the next diagonal block (prepared above)
    // runs beside the update.
    const bool la = done == 16 && kb + 1 < nb;
    if (la && is_ge) {
#ifndef FB_SKIP_GE  // (timing experiments only: wrong results)
#ifdef FB_NO_SPEC
      if (lane < 16) cmaxf[16 * (par ^ 1) + lane] = 0.0f;
      fb_eliminate_block<LD>(Gb, Tb + 272 * (par ^ 1), Xq + 32 + 16 * (par ^ 1), Xq + 16 * (par ^ 1),
                             dvals + 16 * (par ^ 1), dinvs + 16 * (par ^ 1), badin + (par ^ 1), alpha, pert, 0, lane);
#else
      double eg[16], ed, enlp;
      fb_elim_resume(eg, ed, enlp, Gb, lane);
      fb_elim_steps<LD, FB_SPEC_STEPS, 16>(eg, ed, enlp, spec_badm, Xq + 16 * (par ^ 1), dvals + 16 * (par ^ 1), dinvs + 16 * (par ^ 1), alpha, 0, lane);
      fb_elim_finish<LD>(eg, spec_badm, Tb + 272 * (par ^ 1), Xq + 32 + 16 * (par ^ 1), dvals + 16 * (par ^ 1), dinvs + 16 * (par ^ 1),
                         badin + (par ^ 1), pert, 0, lane);
#endif
#endif
#pragma unroll
      for (int s = 0; s < NS; s++) R[s] = double4_t{0.0, 0.0, 0.0, 0.0};
      FBWSTAMP(5);
    } else if (done == 16) {
      // ---- update with a whole panel: every live block (block rows >= kb) -= (its rows of L) (the 16 pivot rows).
      // Both operands are 16 k-rows of stride ld: the pivot rows from Op (the panel's own block column: N), -L
      // from Lb (the panel's own block row: what the elimination left).
#pragma unroll
      for (int s = 0; s < NS; s++) {
        if (!SLOT_IN(m_live, s)) continue;
        const int I_ = SLOT_I(s), J_ = SLOT_J(s);
        const double *ab = J_ == kb ? Tnq + (FBROW(lg) + ln) : Op + (FBROW(lg) + 16 * J_ + ln);
        const double *lb = I_ == kb ? Ldq + (FBROW(lg) + ln) : Lb + (FBROW(lg) + 16 * I_ + ln);
        double a[4], l[4];
#pragma unroll
        for (int q = 0; q < 4; q++) a[q] = ab[q * Q4], l[q] = lb[q * Q4];
        double4_t acc = R[s];
        if (J_ == kb) {  // the eliminated columns of the panel's own block turn into columns of M: from zero
#pragma unroll
          for (int q = 0; q < 4; q++) acc[q] = (lg + 4 * q < off) ? acc[q] : 0.0;
          if (I_ > kb && 16 * I_ + ln < p) {  // and its B operand is the block's part of the L11 columns of the panel
            double *pc = P + ((16 * kb + lg) * Fi + 16 * I_ + ln);
#pragma unroll
            for (int q = 0; q < 4; q++)
              if (lg + 4 * q >= off && 16 * kb + lg + 4 * q < p) pc[4 * q * Fi] = -l[q];
          }
        }
#ifndef FB_SKIP_UPD  // (timing experiments only: wrong results)
#pragma unroll
        for (int q = 0; q < 4; q++) acc = mfma_f64(a[q], l[q], acc);
#else
        acc[0] += a[0] + l[0] + a[3] + l[3];
#endif
        R[s] = acc;
      }
      FBWSTAMP(5);
      // what the next panel and the elimination behind it start from
      if (la) {
        publish_block_row(kb + 1, false);
        if (kb + 2 < nb) publish_start_blocks(kb + 2);
      }
    } else if (done > off) {
      // ---- update with the pivots off .. done - 1 only (the rest of the panel goes to the slow step): operands
      // masked (what a rejected pivot left may be anything), the rows of [N | U] from the 16 x 17 image
      double town[4], ldg[4];
      bool km[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int sp = lg + 4 * q;
        km[q] = sp >= off && sp < done;
        const double t = Tp[sp * 17 + ln];
        town[q] = (!km[q] || (ln > sp && ln < done)) ? 0.0 : t;
        ldg[q] = km[q] ? Ldq[FBROW(sp) + ln] : 0.0;
      }
#pragma unroll
      for (int s = 0; s < NS; s++) {
        if (!SLOT_IN(m_live, s)) continue;
        const int I_ = SLOT_I(s), J_ = SLOT_J(s);
        const bool own = J_ == kb, dgr = I_ == kb;
        const double *o = Op + (FBROW(lg) + 16 * J_ + ln);
        const double *lb = Lb + (FBROW(lg) + 16 * I_ + ln);
        double4_t acc = R[s];
        if (own) {  // the columns of the accepted pivots turn into columns of M: from zero
#pragma unroll
          for (int q = 0; q < 4; q++) acc[q] = km[q] ? 0.0 : acc[q];
        }
        double *pc = P + ((16 * kb + lg) * Fi + 16 * I_ + ln);
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const double ta = o[q * Q4], tl = lb[q * Q4];
          if (own && !dgr && km[q] && 16 * I_ + ln < p) pc[4 * q * Fi] = -tl;  // L11 columns of the accepted pivots
          const double a = own ? town[q] : (km[q] ? ta : 0.0);
          const double l = dgr ? ldg[q] : (km[q] ? tl : 0.0);
          acc = mfma_f64(a, l, acc);
        }
        R[s] = acc;
      }
    }
    FBWSTAMP(6);
    if (done > off && !is_ge) {
      // L11 inside the diagonal block, pivot data of the accepted pivots
      if (tid < 256) {
        const int sp = tid >> 4, rr = tid & 15;
        if (sp >= off && sp < done && rr > sp && 16 * kb + rr < p)
          P[(16 * kb + sp) * Fi + 16 * kb + rr] = -Ldq[FBROW(sp) + rr];
      }
      if (tid >= off && tid < done) dv[2 * (16 * kb + tid)] = dip[tid], dv[2 * (16 * kb + tid) + 1] = 0.0, pt[16 * kb + tid] = 0;
    }
    FBSTAMP(5 + 5 * npan);
    FBWSTAMP(7);
    if (done < kend)
      __syncthreads();  // the slow step reads L11 columns back
    else
      fb_barrier();
    FBWSTAMP(8);
    FBSTAMP(6 + 5 * npan);
    npan++;
    k = 16 * kb + done;
    hot = la;
    if (la) par ^= 1;
    if (done < kend) {
      // ================= slow step: one pivot with the complete test =========================
      // Vectors over the index range of the block: entry x < k of a row vector is the row's entry of M,
      // entry x >= k its entry of the symmetric remainder.
      if (tid == 0) atomicAdd(&counters[2], 1);
      double *V0 = Op, *V1 = Op + ld, *V2 = Op + 2 * ld, *V5 = Op + 5 * ld, *V6 = Op + 6 * ld, *V7 = Op + 7 * ld,
             *V8 = Op + 8 * ld;
      // rows g1 (and g2 >= 0) of [M | remainder] as vectors
      auto publish_rows = [&](double *vec1, int g1, double *vec2, int g2) {
#pragma unroll
        for (int s = 0; s < NS; s++) {
          if (SLOT_IN(m_live, s)) {
            const int i = 16 * SLOT_I(s) + ln, c0 = 16 * SLOT_J(s) + lg;
#pragma unroll
            for (int q = 0; q < 4; q++) {
              const int c = c0 + 4 * q;
              const double v = R[s][q];
              // (the mirror image inside a diagonal block is not used; entry ld - 1 of a vector is nobody's:
              // stores without a match go there instead of behind a branch each)
              const bool low = i >= c;
              vec1[(low && i == g1) ? c : ld - 1] = v;
              vec1[(low && c == g1 && i > g1) ? i : ld - 1] = v;
              if (g2 >= 0) {
                vec2[(low && i == g2) ? c : ld - 1] = v;
                vec2[(low && c == g2 && i > g2) ? i : ld - 1] = v;
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      publish_rows(V0, k, V0, -1);
      __syncthreads();
      // ---- decision, redundantly per wavefront (hqp/spBKP.C:431-437, 471, 480)
      int r = p;
      double lambda = 0.0;
      const double akk = fabs(V0[k]);
      {
        float t[4];
        float tm = 0.0f;
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int i = k + 1 + lane + 64 * u;
          t[u] = i < p ? fabsf((float)V0[i]) : -1.0f;
          tm = fmaxf(tm, t[u]);
        }
        const float tmax = wave_max_dpp_f(tm);
#pragma unroll
        for (int u = 3; u >= 0; u--) {
          const unsigned long long m1 = __ballot(t[u] == tmax);
          if (m1) r = k + 1 + 64 * u + (int)__builtin_ctzll(m1);
        }
        r = __builtin_amdgcn_readfirstlane(r);
        if (r < p) lambda = fabs(V0[r]);
      }
      int kind = 0;
      if (r < p && !(akk >= alpha * lambda)) {  // block-uniform
        publish_rows(V1, r, V2, (k + 1 < p && r != k + 1) ? k + 1 : -1);
        __syncthreads();
        double sg = 0.0;
        for (int t = k + lane; t < p; t += 64)
          if (t != r) sg = fmax(sg, fabs(V1[t]));
        const double sigma = wave_max_dpp(sg);
        if (sigma * akk >= alpha * lambda * lambda)
          kind = 0;
        else if (fabs(V1[r]) >= alpha * sigma)
          kind = 1;
        else
          kind = 2;
      }
      kind = __builtin_amdgcn_readfirstlane(kind);
      const int p1 = (kind == 2) ? k + 1 : k;
      const bool swp = kind != 0 && r != p1;  // symmetric interchange p1 <-> r
      const int klast = (kind == 2) ? k + 1 : k;
      // ---- the pivot row(s) as they stand after the interchange: PR1 = new row k, PR2 = new row k+1
      {
        const double *s1 = (swp && p1 == k) ? V1 : V0;
        const double *s2 = swp ? V1 : (r == k + 1 ? V1 : V2);
        for (int x = tid; x < pp; x += NT) {
          const int px = (swp && x >= k) ? (x == p1 ? r : (x == r ? p1 : x)) : x;
          V7[x] = s1[px];
          if (kind == 2) V8[x] = s2[px];
        }
      }
      __syncthreads();
      double i11 = 0.0, i21 = 0.0, i22 = 0.0;
      const int lpk = swp && p1 == k ? lp[r] : lp[k];                              // rows of A at the pivot positions
      const int lpk1 = kind == 2 ? (swp ? lp[r] : lp[k + 1]) : 0;
      if (kind != 2) {
        double d = V7[k];
        bool pertd = false;
        if (fabs(d) < SOFT_PIVOT_REL * (double)rm0[lpk]) {
          const int sgs = esign[e0 + lpk];
          if (sgs == 2 || sgs == -2) {
            if (tid == 0) counters[4] = 1;  // see SOFT_PIVOT_REL
            if (tiny_replace(counters, d)) d = (sgs < 0 ? -1.0 : 1.0) * fmax(soft_pivot_pert * (double)rm0[lpk], pert), pertd = true;
          }
        }
        if (!(fabs(d) >= pert)) {
          const int sg = esign[e0 + lpk];
          if (d == 0.0 && tid == 0) counters[zero_pivot_slot(sg, sg, b)] = 4;
          d = sg < 0 ? -pert : pert;
          pertd = true;
        }
        i11 = fast_rcp(d);
        if (tid == 0) {
          dv[2 * k] = i11, dv[2 * k + 1] = 0.0, pt[k] = 0;
          if (pertd) atomicAdd(&counters[1], 1);
        }
      } else {
        double d11 = V7[k], d21 = V7[k + 1], d22 = V8[k + 1];
        double det = d11 * d22 - d21 * d21;
        bool pertd = false;
        if (!(fabs(det) >= pert * pert)) {  // degenerate 2x2: perturbed diagonal pair
          const int sg1 = esign[e0 + lpk], sg2 = esign[e0 + lpk1];
          if (det == 0.0 && tid == 0) counters[zero_pivot_slot(sg1, sg2, b)] = 4;  // hqp/spBKP.C:731-732
          d11 = sg1 < 0 ? -pert : pert;
          d22 = sg2 < 0 ? -pert : pert;
          d21 = 0.0;
          det = d11 * d22;
          pertd = true;
        }
        const double rdet = fast_rcp(det);
        i11 = d22 * rdet, i21 = -d21 * rdet, i22 = d11 * rdet;
        if (tid == 0) {
          dv[2 * k] = i11, dv[2 * k + 1] = i21, dv[2 * k + 2] = i22, dv[2 * k + 3] = i21;
          pt[k] = 1, pt[k + 1] = 2;
          atomicAdd(&counters[0], 1);
          if (pertd) atomicAdd(&counters[1], 2);
        }
      }
      // multipliers (zero up to the pivot rows) and the L11 column(s)
      for (int x = tid; x < pp; x += NT) {
        const bool on = x > klast && x < p;
        const double c1 = on ? V7[x] : 0.0, c2 = (on && kind == 2) ? V8[x] : 0.0;
        const double l1 = kind == 2 ? c1 * i11 + c2 * i21 : c1 * i11;
        const double l2 = kind == 2 ? c1 * i21 + c2 * i22 : 0.0;
        V5[x] = l1, V6[x] = l2;
        if (on) {
          P[(long long)k * F + x] = l1;
          if (kind == 2) P[(long long)(k + 1) * F + x] = l2;
        }
        if (kind == 2 && x == k + 1) P[(long long)k * F + x] = 0.0;
      }
      if (swp) {  // rows p1 <-> r of the L11 columns that are written already
        for (int c = tid; c < k; c += NT) {
          unsigned long long *x1 = (unsigned long long *)&P[(long long)c * F + p1], *x2 = (unsigned long long *)&P[(long long)c * F + r];
          const unsigned long long t1 = __hip_atomic_load(x1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned long long t2 = __hip_atomic_load(x2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(x1, t2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(x2, t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      __syncthreads();
      if (swp && tid == 0) {
        const int t = lp[p1];
        lp[p1] = lp[r];
        lp[r] = t;
      }
      // ---- one pass over the register blocks: the interchange, then rows below -= l1 (row k) [+ l2 (row k+1)]
      {
        const double *Vp1 = (p1 == k) ? V0 : V2;
#pragma unroll
        for (int s = 0; s < NS; s++) {
          if (SLOT_IN(m_live, s)) {
            const int i = 16 * SLOT_I(s) + ln, c0 = 16 * SLOT_J(s) + lg;
            const double l1 = V5[i], l2 = V6[i];
#pragma unroll
            for (int q = 0; q < 4; q++) {
              const int c = c0 + 4 * q;
              double v = R[s][q];
              if (swp) {
                if (c < k) {
                  if (i == p1) v = V1[c];
                  if (i == r) v = Vp1[c];
                } else if (i >= k) {
                  const int a = i == p1 ? r : (i == r ? p1 : i), bq = c == p1 ? r : (c == r ? p1 : c);
                  if (a != i || bq != c)  // old entry (a, bq) of the symmetric remainder: one of the two is p1 or r
                    v = (a == p1) ? Vp1[bq] : (a == r) ? V1[bq] : (bq == p1) ? Vp1[a] : V1[a];
                }
              }
              if (i > klast) {
                if (c == k)
                  v = -l1;
                else if (kind == 2 && c == k + 1)
                  v = -l2;
                else {
                  v = fma(-l1, V7[c], v);
                  if (kind == 2) v = fma(-l2, V8[c], v);
                }
              } else if (i >= k) {  // the pivot rows themselves: rows of M from now on
                if (c == i) v = 1.0;
                if (kind == 2 && i == k + 1 && c == k) v = 0.0;
              }
              R[s][q] = v;
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __syncthreads();
      k = klast + 1;
    }
  }
  // ---- M = L11^-1: every block row is complete (zeros above the diagonal inside the diagonal blocks)
#pragma unroll
  for (int s = 0; s < NS; s++) {
    if (!SLOT_IN(m_all, s)) continue;
    const int I_ = SLOT_I(s), J_ = SLOT_J(s);
    const int row = 16 * I_ + ln;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int cl = lg + 4 * q, col = 16 * J_ + cl;
      double v = R[s][q];
      if (J_ == I_) v = cl < ln ? v : (cl == ln ? 1.0 : 0.0);
      if (row < p && col < p) W[col * p + row] = v;
    }
  }
  __syncthreads();
  FBSTAMP(53);
  for (int i = tid; i < p; i += NT) {
    lperm[e0 + i] = lp[i];
    ptype[e0 + i] = pt[i];
    dinv[2 * (e0 + i)] = dv[2 * i];
    dinv[2 * (e0 + i) + 1] = dv[2 * i + 1];
  }
}

}  // namespace kktdev
