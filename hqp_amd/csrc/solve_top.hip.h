// ---- the solve sweeps of the general fronts across tree levels: one launch per sweep (or one for both)
//
// Where a tree level is a handful of fronts, a level of the solve is a latency chain (gather the children's
// contributions, y = M P t, c - L21 y; then v = yd - L21' x, x = P' M' v) and its cost is the launch, the first-touch
// memory latency of M and L21 and the dependent index loads: 22 + 19 us per level on the C2 tree against 1 - 2 us of
// arithmetic.  k_solve_top gives every front ONE workgroup of 16 wavefronts:
//   - before anything else it loads what depends on the front alone: the lower triangle of M = L11^-1 into LDS
//     (packed by columns), L21 into registers (lane = border row within a 64-row slab, wavefront w holds the columns
//     k = w (mod 16): NS slabs x NU columns per lane), permutation, pivot data, index lists;
//   - forward step when its children's contributions have arrived, backward step when the solution at its border
//     rows has.
// Data between fronts travels WITHOUT flags and without cache maintenance: the words themselves are the signal
// (kernels.hip.h, "data words as signals").  A front writes its contribution vector (forward) and its part of the
// solution (backward) with agent-scope atomic stores into small exchange arrays whose words hold a sentinel until then;
// the consumer's lanes poll exactly the words they need.  One memory round trip per level, where a flag protocol pays
// release (L2 write-back), flag, acquire (invalidate), data (measured: 13 us per level and direction).  The arrays exist
// twice: solve e uses copy e & 1 and every front puts the sentinel back into ITS words of the other copy, which nobody
// reads during this solve; e is the device counter k_rhs_* bumps at the start of every solve (the launches are replayed
// from a graph).
// MODE 1 / 2 - the form in use: the forward sweep (`nodes` leaves first) and the backward sweep (root first) as launches
// of their own, for any number of fronts: a waiting front only waits for fronts before it in the launch, which relies on
// workgroups being dispatched in index order when the chip does not hold them all.  MODE 0: both sweeps in one launch,
// M and L21 read once; all fronts (at most ST_MAXFRONTS) must be resident at once, so it is an option
// (HQPKKT_SOLVE_TOP_FUSED), not the default: several systems in flight on one GPU could starve each other.
// A poll gives up after ~2^20 tries (about a second) and raises flags[ST_GAVE_UP]: hqpkkt_solve then rebuilds the
// exchange arrays and reports HQPKKT_E_DEVICE instead of hanging the device.
// Arithmetic: plain FMA sums in a fixed order (thread-local over k, then the 16 / 4 partial sums in index order; the
// backward column sums by the wavefront reduction) - reproducible from run to run and the same in all three modes, not
// bit-identical to the per-level kernels (different order of summation).
#pragma once

namespace kktdev {

static const int ST_THREADS = 1024, ST_MAXFRONTS = 128, ST_MAXSPLIT = 1 << 15;  // fronts of one fused launch / of the split form
static const int ST_XS = 192, ST_CS = 256;  // words per front in the exchange arrays: solution (pivots), contribution (border rows)
static const int ST_GAVE_UP = XW_GAVE_UP;    // index into the handle's flags buffer
static const unsigned long long ST_SENTINEL = XW_SENTINEL;

// the two instances: <3, 11> fronts of up to 176 pivots and 192 border rows, <4, 10> up to 160 pivots and 256 border rows
static inline bool st_top_fits(int p, int b, int ns, int nu) { return p <= 16 * nu && b <= 64 * ns; }
static inline size_t st_top_lds_bytes(int maxp, int ns) {
  const size_t mlen = ((size_t)maxp * (maxp + 1) / 2 + 1) & ~(size_t)1;
  return sizeof(double) * (mlen + 8 * 64 * ns + 16 * 64 * ns);
}
__device__ __forceinline__ void st_post(double *p, double v) { xw_post(p, v); }
__device__ __forceinline__ double st_take(const double *p, int *flags) { return xw_take(p, flags); }

struct TopArgs {
  const int *nodes;    // fronts of the fused levels, root first
  const int *top_idx;  // supernode -> index into nodes, -1 below the fused levels
  const int *bpos;     // [front][border row] -> word of the solution exchange array
  double *xcb, *xx;    // exchange arrays: 2 x ntop x ST_CS contributions, 2 x ntop x ST_XS solution
  const int *epoch;    // solves so far (k_rhs_* counts): this solve uses copy epoch & 1 of the exchange arrays
  int ntop;
  unsigned long long *stamps;  // diagnostics (hqpkkt_debug_solve_top_stamps): 8 times per front, or null
};
#define ST_STAMP(k) \
  if (A.stamps && tid == 0) A.stamps[8 * me + (k)] = __builtin_amdgcn_s_memrealtime()

// MODE 0: both sweeps in one launch (all fronts resident at once: at most ST_MAXFRONTS); 1 / 2: the forward sweep
// (`nodes` leaves first) and the backward sweep (root first) as launches of their own, for any number of fronts - a
// waiting front only waits for fronts before it in the launch, which relies on workgroups being dispatched in index
// order when the chip does not hold them all.
template <int NS, int NU, int MODE>
__global__ void __launch_bounds__(ST_THREADS)
k_solve_top(DevTree T, TopArgs A, const double *__restrict__ panel, const double *__restrict__ linv,
            const long long *__restrict__ linv_off, const double *__restrict__ dinv, const int *__restrict__ ptype,
            const int *__restrict__ lperm, const double *__restrict__ rhs, double *__restrict__ xsol,
            const double *__restrict__ cb, int *__restrict__ flags) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int node = A.nodes[blockIdx.x], me = A.top_idx[node];  // me: the front's slot in the exchange arrays
  const int p = T.npiv[node], b = T.nbor[node];
  const long long F = p + b;
  const int e0 = T.piv_start[node];
  const double *P = panel + T.panel_off[node];
  const double *W = linv + linv_off[node];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  constexpr int VEC = 64 * NS;  // length of the vectors in LDS (pivots, border rows) and stride of the partial sums
  ST_STAMP(MODE == 2 ? 6 : 0);  // (slots 6, 7: start and static data of the backward launch of the split form)
  const int par = *A.epoch & 1;
  const size_t mlen = ((size_t)p * (p + 1) / 2 + 1) & ~(size_t)1;
  double *Ms = lds;  // column t of M from its diagonal down: Ms[t p - t (t - 1) / 2 + (i - t)]
  double *t1 = Ms + mlen, *tp = t1 + VEC, *y = tp + VEC, *xd = y + VEC, *vv = xd + VEC, *cbs = vv + VEC,
         *x2 = cbs + VEC, *part = x2 + 2 * VEC;  // part: 16 x VEC (4 x 256 in the products with M)

  // ---- everything that depends on the front alone
  double l[NS][NU];  // L21(64 s + lane, wave + 16 u)
#pragma unroll
  for (int s = 0; s < NS; s++)
#pragma unroll
    for (int u = 0; u < NU; u++) {
      const int i = 64 * s + lane, k = wave + 16 * u;
      const bool ok = i < b && k < p;
      const double x = P[ok ? (long long)k * F + p + i : 0];
      l[s][u] = ok ? x : 0.0;
    }
#pragma unroll
  for (int u = 0; u < NU; u++) {
    const int t = wave + 16 * u;
    if (t < p) {  // wave-uniform
      const int off = t * p - (t * (t - 1)) / 2 - t;
#pragma unroll
      for (int r = 0; r < 3; r++) {
        const int i = t + lane + 64 * r;
        if (i < p) Ms[off + i] = W[(long long)t * p + i];
      }
    }
  }
  int lpk = 0, pty = 0;
  double pd0 = 0.0, pd1 = 0.0, rv = 0.0;
  if (tid < p) {
    lpk = lperm[e0 + tid], pty = ptype[e0 + tid];
    pd0 = dinv[2 * (e0 + tid)], pd1 = dinv[2 * (e0 + tid) + 1];
    rv = rhs[e0 + tid];
  }
  const int bpos = tid < b ? A.bpos[me * ST_CS + tid] : 0;
  const int c0 = T.child_ptr[node], c1 = T.child_ptr[node + 1];
  const int parent = T.parent[node];
  const bool has_parent = parent >= 0;  // (inside this launch: the fused levels are closed upwards)
  if (tid < VEC) t1[tid] = rv, cbs[tid] = 0.0;
  __syncthreads();
  double *xcb = A.xcb + (size_t)par * A.ntop * ST_CS, *xx = A.xx + (size_t)par * A.ntop * ST_XS;
  {  // this front's words of the other copy: back to the sentinel for the next launch
    double *ocb = A.xcb + (size_t)(par ^ 1) * A.ntop * ST_CS, *ox = A.xx + (size_t)(par ^ 1) * A.ntop * ST_XS;
    if (MODE != 2 && tid < ST_CS) st_post(ocb + me * ST_CS + tid, __longlong_as_double((long long)ST_SENTINEL));
    if (MODE != 1 && tid < ST_XS) st_post(ox + me * ST_XS + tid, __longlong_as_double((long long)ST_SENTINEL));
  }

  ST_STAMP(MODE == 2 ? 7 : 1);  // static data requested / in LDS
  int nfc = 0;  // children inside the fused levels
  if constexpr (MODE != 2) {
  // ---- forward: t = rhs + children, y = M P t, yd = D^-1 y, contribution = c - L21 y
  for (int cc = c0; cc < c1; cc++) {
    const int c = T.child_idx[cc];
    const int bc = T.nbor[c], ci = A.top_idx[c];
    const int *rel = T.rel + T.bptr[c];
    const double *cbc = cb + T.cb_off[c];
    if (ci >= 0) {  // block-uniform
      nfc++;
      if (tid < bc) {
        const int ri = rel[tid];
        const double v = st_take(xcb + ci * ST_CS + tid, flags);
        if (ri < p)
          t1[ri] += v;
        else
          cbs[ri - p] += v;
      }
    } else {
      for (int j = tid; j < bc; j += ST_THREADS) {  // (a child below the fused levels: finished by the launches before)
        const int rj = rel[j];
        if (rj < p)
          t1[rj] += cbc[j];
        else
          cbs[rj - p] += cbc[j];
      }
    }
    __syncthreads();
  }
  ST_STAMP(2);  // children have arrived
  if (tid < p) tp[tid] = t1[lpk];
  __syncthreads();
  {
    // y = M tp (M lower): row r and row p - 1 - r in one thread (the triangle's work balances), eight column classes
    // t = c (mod 8); the packed index of (i, t) steps by 8 p - 8 t - 36 from t to t + 8
    const int r = tid & 127, c = tid >> 7;
    const int iA = r, iB = p - 1 - r;
    double aA = 0.0, aB = 0.0;
    if (2 * r < p) {
      int idx = c * p - (c * (c - 1)) / 2 - c, step = 8 * p - 8 * c - 36;
      for (int t = c; t <= iB; t += 8) {
        const double x = tp[t];
        if (t <= iA) aA = fma(Ms[idx + iA], x, aA);
        aB = fma(Ms[idx + iB], x, aB);
        idx += step, step -= 64;
      }
      part[256 * c + iA] = aA;
      if (iB != iA) part[256 * c + iB] = aB;
    }
  }
  __syncthreads();
  if (tid < p)
    y[tid] = ((part[tid] + part[256 + tid]) + (part[512 + tid] + part[768 + tid])) +
             ((part[1024 + tid] + part[1280 + tid]) + (part[1536 + tid] + part[1792 + tid]));
  __syncthreads();
  if (tid < p) {
    const int kp = pty == 2 ? tid - 1 : min(tid + 1, p - 1);  // partner of a 2x2 pivot
    xd[tid] = pty == 0 ? y[tid] * pd0 : y[tid] * pd0 + y[kp] * pd1;
  }
  if (b > 0) {  // block-uniform
    double yk[NU];
#pragma unroll
    for (int u = 0; u < NU; u++) yk[u] = wave + 16 * u < p ? y[wave + 16 * u] : 0.0;
#pragma unroll
    for (int s = 0; s < NS; s++) {
      double a = 0.0;
#pragma unroll
      for (int u = 0; u < NU; u++) a = fma(l[s][u], yk[u], a);
      part[VEC * wave + 64 * s + lane] = a;
    }
    __syncthreads();
    if (tid < b) {
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int w = 0; w < 16; w += 2) s0 += part[VEC * w + tid], s1 += part[VEC * (w + 1) + tid];
      st_post(xcb + me * ST_CS + tid, cbs[tid] - (s0 + s1));
    }
  }

  if constexpr (MODE == 1) {  // yd for the backward launch
    if (tid < p) xsol[e0 + tid] = xd[tid];
    return;
  }
  } else {  // (backward launch: yd from the forward one; the fused children are counted for the posting below)
    if (tid < p) xd[tid] = xsol[e0 + tid];
    for (int cc = c0; cc < c1; cc++) nfc += A.top_idx[T.child_idx[cc]] >= 0 ? 1 : 0;
  }
  ST_STAMP(3);  // forward step done
  // ---- backward: v = yd - L21' x(border), x = P' M' v
  if (tid < VEC) x2[tid] = tid < b && has_parent ? st_take(xx + bpos, flags) : 0.0;
  __syncthreads();
  ST_STAMP(4);  // solution at the border rows has arrived
  {
    double xs[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) xs[s] = x2[64 * s + lane];
#pragma unroll
    for (int u = 0; u < NU; u++) {
      double a = 0.0;
#pragma unroll
      for (int s = 0; s < NS; s++) a = fma(l[s][u], xs[s], a);
      a = wave_sum_dpp(a);  // (every lane has the sum)
      const int k = wave + 16 * u;
      if (lane == 0 && k < p) vv[k] = xd[k] - a;
    }
  }
  __syncthreads();
  {
    // z = M' v: columns r and p - 1 - r in one thread, eight row classes i = t + c (mod 8)
    const int r = tid & 127, c = tid >> 7;
    const int tA = r, tB = p - 1 - r;
    double aA = 0.0, aB = 0.0;
    if (2 * r < p) {
      const double *colA = Ms + (tA * p - (tA * (tA - 1)) / 2 - tA), *colB = Ms + (tB * p - (tB * (tB - 1)) / 2 - tB);
      for (int i = tA + c; i < p; i += 8) aA = fma(colA[i], vv[i], aA);
      for (int i = tB + c; i < p; i += 8) aB = fma(colB[i], vv[i], aB);
      part[256 * c + tA] = aA;
      if (tB != tA) part[256 * c + tB] = aB;
    }
  }
  __syncthreads();
  if (tid < p) {
    const double z = ((part[tid] + part[256 + tid]) + (part[512 + tid] + part[768 + tid])) +
                     ((part[1024 + tid] + part[1280 + tid]) + (part[1536 + tid] + part[1792 + tid]));
    if (nfc > 0) st_post(xx + me * ST_XS + lpk, z);
    xsol[e0 + lpk] = z;  // (for the levels below, after this launch)
  }
  ST_STAMP(5);
}

}  // namespace kktdev
