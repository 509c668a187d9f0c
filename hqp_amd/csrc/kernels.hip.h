// gfx950 kernels of the KKT path.  Included once by hqpkkt.hip.
//
// Data layout in HBM (all fp64 values, int32 indices, int64 arena offsets):
//   vals   [nq+na+nc+1]   Qx | Ax | Cx | 1.0            (hqpkkt_set_values)
//   wt     [m+1]          per-inequality weight | 1.0   (w/z FULL, z/w REDUCED)
//   sc     [dim]          symmetric scaling in QP numbering
//   panel  per supernode  F x p column-major (ld = F): rows 0..p-1 the pivot
//                         block (lower triangle used), rows p..F-1 the border
//   upd    per supernode  b x b column-major, lower triangle: Schur complement
//                         handed to the parent (extend-add)
//   xar    per supernode  b x p column-major: X = L21 * D (kept for the update)
//   dinv   [2*dim]        per elimination index: inverse pivot data
//   ptype  [dim]          0: 1x1, 1: first row of a 2x2, 2: second row of a 2x2
//   lperm  [dim]          pivot order inside each supernode (local row index)
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>
namespace kktdev {

typedef double double4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double4_t mfma_f64(double a, double b, double4_t c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

struct DevTree {
  const int *piv_start, *npiv, *nbor, *parent;
  const long long *bptr;
  const int *bidx, *rel;
  const long long *panel_off, *upd_off, *x_off, *cb_off;
  const int *child_ptr, *child_idx;
  const int *pinv;            // per child: parent front index -> child border index (-1: none)
  const long long *pinv_off;
};

// Sum of the children's update blocks at front position (fi, fj), fi >= fj, of `node`:
// what an extend-add pass would have added to that entry (children in slot order).
__device__ __forceinline__ double gather_children(const DevTree &T, const double *__restrict__ upd, int node,
                                                  int fi, int fj) {
  double s = 0.0;
  for (int cc = T.child_ptr[node]; cc < T.child_ptr[node + 1]; cc++) {
    const int c = T.child_idx[cc];
    const int *iv = T.pinv + T.pinv_off[c];
    const int ci = iv[fi], cj = iv[fj];
    if (ci >= 0 && cj >= 0) s += upd[T.upd_off[c] + (long long)cj * T.nbor[c] + ci];
  }
  return s;
}

// order-preserving max for non-negative doubles through their bit pattern
// (the read-modify-writes of a launch's workgroups on ONE word are served one after the other, ~6 ns each: launches that
// end in this keep their grids at a thousand or two workgroups; a look at the word first - an agent-scope load - cost
// k_assemble_simple more than it saved k_residual)
__device__ __forceinline__ void atomic_max_pos(unsigned long long *addr, double v) {
  atomicMax(addr, (unsigned long long)__double_as_longlong(v));
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}
// max over the 64 lanes of a non-negative value with DPP row operations (no LDS
// crossbar traffic); the result is broadcast to every lane
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move(double v) {
  const long long b = __double_as_longlong(v);
  int lo = (int)(b & 0xffffffffLL), hi = (int)(b >> 32);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_max_dpp(double v) {
  v = fmax(v, dpp_move<0xb1, 0xf>(v));   // quad_perm [1,0,3,2]
  v = fmax(v, dpp_move<0x4e, 0xf>(v));   // quad_perm [2,3,0,1]
  v = fmax(v, dpp_move<0x124, 0xf>(v));  // row_ror 4
  v = fmax(v, dpp_move<0x128, 0xf>(v));  // row_ror 8
  v = fmax(v, dpp_move<0x142, 0xa>(v));  // row_bcast 15
  v = fmax(v, dpp_move<0x143, 0xc>(v));  // row_bcast 31 -> lane 63 holds the max
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), 63);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// sum over the 64 lanes the same way (lanes a DPP step does not reach contribute zero); broadcast to every lane
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move0(double v) {
  const long long b = __double_as_longlong(v);
  int lo = (int)(b & 0xffffffffLL), hi = (int)(b >> 32);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
  v += dpp_move0<0xb1, 0xf>(v);   // quad_perm [1,0,3,2]
  v += dpp_move0<0x4e, 0xf>(v);   // quad_perm [2,3,0,1]
  v += dpp_move0<0x124, 0xf>(v);  // row_ror 4
  v += dpp_move0<0x128, 0xf>(v);  // row_ror 8: every lane of a row has the row's sum
  v += dpp_move0<0x142, 0xa>(v);  // row_bcast 15: rows 1 and 3 add the sum of the row before
  v += dpp_move0<0x143, 0xc>(v);  // row_bcast 31: rows 2 and 3 add lane 31 -> lane 63 holds the total
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), 63);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// same for a float (the arg-max search of the pivot column runs in fp32: an fp64
// max costs ~40 cycles of latency per step on gfx950, an fp32 one a few)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL,
                                                    ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_max_dpp_f(float v) {
  v = fmaxf(v, dpp_move_f<0xb1, 0xf>(v));
  v = fmaxf(v, dpp_move_f<0x4e, 0xf>(v));
  v = fmaxf(v, dpp_move_f<0x124, 0xf>(v));
  v = fmaxf(v, dpp_move_f<0x128, 0xf>(v));
  v = fmaxf(v, dpp_move_f<0x142, 0xa>(v));
  v = fmaxf(v, dpp_move_f<0x143, 0xc>(v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// 1/d by the hardware estimate and two Newton steps (error ~1 ulp; the pivot is
// bounded away from zero and from overflow by the perturbation test)
__device__ __forceinline__ double fast_rcp(double d) {
  double x = __builtin_amdgcn_rcp(d);
  x = fma(fma(-d, x, 1.0), x, x);
  x = fma(fma(-d, x, 1.0), x, x);
  return x;
}


__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---------------------------------------------------------------- assembly
// wt / scaling from (z, w).  FULL: Hqp_IpSpBKP::factor, hqp/Hqp_IpSpBKP.C:152-160;
// REDUCED: v_slash(w, z, _zw), hqp/Hqp_IpRedSpBKP.C:296.
__global__ void k_weights(int mode, int m, int nme, const double *__restrict__ z,
                          const double *__restrict__ w, double *__restrict__ wt,
                          double *__restrict__ sc, int *__restrict__ status) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j == 0) wt[m] = 1.0;
  if (j >= m) return;
  double zj = z[j], wj = w[j];
  if (zj == 0.0 || wj == 0.0) {  // v_slash raises E_SING (meschach/vecop.c:346-348)
    atomicExch(status, 4);
    wt[j] = 1.0;
    if (mode == 0) sc[nme + j] = 1.0;
    return;
  }
  if (mode == 0) {
    double wz = wj / zj;
    wt[j] = wz;
    sc[nme + j] = fmin(1.0, sqrt(1.0 / wz));
  } else {
    wt[j] = zj / wj;
  }
}

struct TermDev {
  int s1, s2, wi;
  double sgn;
};

// An exactly zero pivot is E_SING for the reference (hqp/spBKP.C:699-700), whose pivot search
// sees the whole remaining column.  Here the search ends at the supernode's pivot block: in a
// root front, or on a variable without a diagonal of its own (esign +-2: an equality multiplier
// or an x without Q_ii - the zero then says that its rows are rank deficient), the zero is the
// reference's E_SING (status word counters[-1]).  Elsewhere (e.g. a slack row whose w/z = 1e-18
// was absorbed by c^2/Q_ii = 1 and cancelled again, its coupling to an ancestor's x still to
// come) the pivot is perturbed like a tiny one and only counters[3] is marked: hqpkkt_solve
// reports E_SING if the refinement then does not reach mat_eps.
// A rank-deficient equality block does not always end in an EXACT zero here (the reference's search over
// the whole remaining column finds one, hqp/spBKP.C:699-700: E_SING): with the search restricted to the
// pivot block the second of two identical rows can be left with a pivot of 1e-17.  A pivot of a variable
// without a diagonal of its own (equality multiplier, x without Q_ii) that is below SOFT_PIVOT_REL times the scale of its row (below) marks the
// factorisation (counters[4]): hqpkkt_solve reports E_SING if its refinement then ends with a residual above 1e-4 (a
// solution that is garbage, not one that is a few digits short of mat_eps as in the last iterations of an
// interior-point run).  Until round 5 the pivot itself was used as it was; since then it is replaced (soft_pivot_pert
// below): used as it was, a cancelled pivot of 1e-17 turns the factors into garbage whenever the rows were NOT
// dependent - the last iterations of hqpkkt_franke on the double-integrator QPs, where the reference's search over
// the whole column finds a proper pivot; all ten finds of the interior-point campaigns went back to this.
// ... relative to the largest entry of the pivot's row in the block as it was assembled: a pivot that
// cancelled (-c + c for the second of two identical equality rows) is 1e-16 of it, the multiplier pivots
// of a late interior-point iteration are of its order whatever max|K| = z/w has grown to
#define SOFT_PIVOT_REL 1e-13
// > 0: such a pivot is REPLACED by this multiple of its row's scale (static pivoting as the reference's PARDISO plugin
// configures it, hqp/Hqp_IpPARDISO.C: perturbed pivots + iterative refinement) instead of being used as it is (1e-6: with
// 1e-8 two of the ten finds of rounds 1-4 stayed - a smaller replacement amplifies the rounding errors of its row more
// than the refinement gains from the smaller change)
__device__ double soft_pivot_pert = 1e-6;
// ... but only where the caller has said that the matrix is known to be regular: a cancelled multiplier pivot is ALSO what
// a rank-deficient equality block leaves behind (two identical rows), which the reference reports as E_SING / "degenerate"
// at the first factorisation - a replaced pivot would solve the consistent singular system and go on.  Word
// TINY_REPLACE_WORD of the handle's flags (k_clear leaves it alone) switches the replacement on: the device-resident
// interior-point loops set it once their first factorisation + solve has succeeded (hqpkkt_mehrotra: behind the cold
// start; hqpkkt_franke: from the second iteration on) and clear it when they return; the plugin entry points
// (hqpkkt_factor on its own, the reference's solvers through the shim) never replace.  An EXACTLY zero pivot is replaced
// only where the loop's first factorisation of the run met NO cancelled multiplier pivot at all (the loops then set the
// word to TINY_REPLACE_ZEROS): a rank-deficient equality block shows its cancelled pivot in every factorisation, the first
// included - tiny there, exactly zero in the next (tests/test_reference_host.py, the duplicated equality row) - and stays
// the reference's E_SING; a zero that turns up at the END of a run whose first factorisations were clean is cancellation
// (w / z of 1e-21 beside 1e+9 at a gap of 4e-8: campaign case 187 of round 6) and is replaced like any other cancelled
// pivot.  Without the replacement a zero stays the reference's E_SING (zero_pivot_slot above).
static const int TINY_REPLACE_WORD = 112;
// (the word: 0 off; TINY_REPLACE_ON: cancelled pivots that are not exactly zero; TINY_REPLACE_ZEROS: exactly zero ones as well)
static const int TINY_REPLACE_ON = 0x01010101, TINY_REPLACE_ZEROS = 0x02020202;  // (set by hipMemsetAsync: a byte value)
__device__ __forceinline__ bool tiny_replace(const int *counters, double d) {
  const int t = counters[TINY_REPLACE_WORD - 1];
  return t != 0 && (d != 0.0 || t == TINY_REPLACE_ZEROS) && soft_pivot_pert > 0.0;
}
// (The replacement cures the runs that ended early on garbage factors - all ten finds of the campaigns of rounds 1-4 - and
// breaks about as many others, where the pivot used as it was had been good enough: six of the 12 000 cases of
// profiles/r05_fuzz_tree.txt against seven without it.  Trying both treatments per solve and keeping the better one was
// measured as well: the same six.  Neither is right everywhere; what is, is a pivot from outside the block.)
__device__ __forceinline__ int zero_pivot_slot(int sg1, int sg2, int b) {
  return (b == 0 || sg1 == 2 || sg1 == -2 || sg2 == 2 || sg2 == -2) ? -1 : 3;
}

// unscaled value of every stored entry of the matrix to factor
__global__ void k_entry_values(int nent, const int *__restrict__ term_ptr,
                               const TermDev *__restrict__ terms,
                               const double *__restrict__ vals, const double *__restrict__ wt,
                               double *__restrict__ ent_val, int *__restrict__ epoch) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e == 0 && epoch) *epoch += 1;  // the factorisation counter of the whole-tree launch (k_factor_diag_small<true, true>)
  if (e >= nent) return;
  double v = 0.0;
  for (int t = term_ptr[e]; t < term_ptr[e + 1]; t++) {
    TermDev tm = terms[t];
    v += tm.sgn * vals[tm.s1] * vals[tm.s2] * wt[tm.wi];
  }
  ent_val[e] = v;
}

// k_weights + k_entry_values as ONE launch (round 6, as k_rhs_red_t below: a launch inside a replayed graph is 4.6 us): the
// first nb_first workgroups the entries, with the weight evaluated where a term needs it (the same quotient: the same
// bits), the rest store the weights (the solves read them) and see zeros in z / w.
__device__ __forceinline__ double weight_elem(int mode, int m, const double *__restrict__ z, const double *__restrict__ w, int j) {
  if (j >= m) return 1.0;  // (wt[m]: the constant weight)
  const double zj = z[j], wj = w[j];
  if (zj == 0.0 || wj == 0.0) return 1.0;
  return mode == 0 ? wj / zj : zj / wj;
}
__global__ void __launch_bounds__(256)
k_wt_entry(int mode, int m, int nme, int nent, int nb_first, const double *__restrict__ z, const double *__restrict__ w, double *__restrict__ wt,
           double *__restrict__ sc, int *__restrict__ status, const int *__restrict__ term_ptr, const TermDev *__restrict__ terms,
           const double *__restrict__ vals, double *__restrict__ ent_val, int *__restrict__ epoch) {
  if ((int)blockIdx.x >= nb_first) {
    const int j = ((int)blockIdx.x - nb_first) * blockDim.x + threadIdx.x;
    if (j == 0) wt[m] = 1.0;
    if (j >= m) return;
    const double zj = z[j], wj = w[j];
    if (zj == 0.0 || wj == 0.0) {  // v_slash raises E_SING (meschach/vecop.c:346-348)
      atomicExch(status, 4);
      wt[j] = 1.0;
      if (mode == 0) sc[nme + j] = 1.0;
      return;
    }
    if (mode == 0) {
      const double wz = wj / zj;
      wt[j] = wz;
      sc[nme + j] = fmin(1.0, sqrt(1.0 / wz));
    } else
      wt[j] = zj / wj;
    return;
  }
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e == 0 && epoch) *epoch += 1;  // the factorisation counter of the whole-tree launch (k_factor_diag_small<true, true>)
  if (e >= nent) return;
  double v = 0.0;
  for (int t = term_ptr[e]; t < term_ptr[e + 1]; t++) {
    const TermDev tm = terms[t];
    v += tm.sgn * vals[tm.s1] * vals[tm.s2] * weight_elem(mode, m, z, w, tm.wi);
  }
  ent_val[e] = v;
}

// REDUCED: scale_i = min(1, sqrt(-1/J_ii))  (hqp/Hqp_IpRedSpBKP.C:130,138)
__device__ __forceinline__ double red_scale_elem(const int *__restrict__ diag_ent, const double *__restrict__ ent_val, int i) {
  const int e = diag_ent[i];
  return e >= 0 ? fmin(1.0, sqrt(-1.0 / ent_val[e])) : 1.0;
}
__global__ void k_red_scale(int n, const int *__restrict__ diag_ent,
                            const double *__restrict__ ent_val, double *__restrict__ sc) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  sc[i] = red_scale_elem(diag_ent, ent_val, i);
}

// J <- S J S scattered into the supernode panels (hqp/Hqp_IpSpBKP.C:162-176);
// entries are stored in destination order, so consecutive threads write (mostly)
// consecutive addresses.  max|K_ij| for the pivot-perturbation threshold: one
// atomic per workgroup (one per wave on a single word serialises at ~12 ns each).
__global__ void __launch_bounds__(256)
k_scatter(int nent, const int *__restrict__ ent_a, const int *__restrict__ ent_b,
          const long long *__restrict__ ent_dst, const double *__restrict__ ent_val,
          const double *__restrict__ sc, double *__restrict__ panel,
          unsigned long long *__restrict__ kmax) {
  __shared__ double red[4];
  double mx = 0.0;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nent; e += gridDim.x * blockDim.x) {
    const double v = ent_val[e] * sc[ent_a[e]] * sc[ent_b[e]];
    panel[ent_dst[e]] = v;
    mx = fmax(mx, fabs(v));
  }
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    mx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    if (mx > 0.0) atomic_max_pos(kmax, mx);
  }
}

// k_red_scale + k_scatter as ONE launch (REDUCED): the first nb_first workgroups scatter, with the scale of a row of x
// evaluated where an entry needs it (rows of the equality multipliers: 1, from memory), the rest store the scales.
__global__ void __launch_bounds__(256)
k_scale_scatter(int n, int nent, int nb_first, const int *__restrict__ diag_ent, const int *__restrict__ ent_a, const int *__restrict__ ent_b,
                const long long *__restrict__ ent_dst, const double *__restrict__ ent_val, double *__restrict__ sc,
                double *__restrict__ panel, unsigned long long *__restrict__ kmax) {
  if ((int)blockIdx.x >= nb_first) {
    const int i = ((int)blockIdx.x - nb_first) * blockDim.x + threadIdx.x;
    if (i < n) sc[i] = red_scale_elem(diag_ent, ent_val, i);
    return;
  }
  __shared__ double red[4];
  double mx = 0.0;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nent; e += nb_first * blockDim.x) {
    const int a = ent_a[e], b = ent_b[e];
    const double sa = a < n ? red_scale_elem(diag_ent, ent_val, a) : sc[a], sb = b < n ? red_scale_elem(diag_ent, ent_val, b) : sc[b];
    const double v = ent_val[e] * sa * sb;
    panel[ent_dst[e]] = v;
    mx = fmax(mx, fabs(v));
  }
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    mx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    if (mx > 0.0) atomic_max_pos(kmax, mx);
  }
}

// FULL plugin: every stored entry is a single term sgn * vals[s1] * wt[wi] (an entry of
// -Q, A or C, or a slack diagonal w/z), so value, scaling and scatter are one pass over a
// compact record (source index with the sign in its top bit, weight index).
__global__ void __launch_bounds__(256)
k_assemble_simple(int nent, const int *__restrict__ src, const int *__restrict__ wi,
                  const int *__restrict__ ent_a, const int *__restrict__ ent_b,
                  const long long *__restrict__ ent_dst, const double *__restrict__ vals,
                  const double *__restrict__ wt, const double *__restrict__ sc, double *__restrict__ panel,
                  unsigned long long *__restrict__ kmax, int *__restrict__ epoch) {
  __shared__ double red[4];
  double mx = 0.0;
  if (blockIdx.x == 0 && threadIdx.x == 0 && epoch) *epoch += 1;  // (see k_entry_values)
  // four entries per thread and trip: their records travel together, then their gathers (one entry per trip made every
  // trip two dependent round trips to memory: 77 us for the 4.9e6 entries of C2, round 6: see profiles/r06_c2_timeline.txt)
  const int stride = gridDim.x * blockDim.x;
  for (int e0 = blockIdx.x * blockDim.x + threadIdx.x; e0 < nent; e0 += 4 * stride) {
    int sg[4], wk[4], ea[4], eb[4];
    long long ed[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int e = min(e0 + k * stride, nent - 1);  // (behind the last entry: the last one again, not stored)
      sg[k] = src[e], wk[k] = wi[e], ea[k] = ent_a[e], eb[k] = ent_b[e], ed[k] = ent_dst[e];
    }
    double va[4], wa[4], sa[4], sb[4];
#pragma unroll
    for (int k = 0; k < 4; k++) va[k] = vals[sg[k] & 0x7fffffff], wa[k] = wt[wk[k]], sa[k] = sc[ea[k]], sb[k] = sc[eb[k]];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const double raw = va[k] * wa[k];
      const double v = (sg[k] < 0 ? -raw : raw) * sa[k] * sb[k];
      if (e0 + k * stride < nent) {
        panel[ed[k]] = v;
        mx = fmax(mx, fabs(v));
      }
    }
  }
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    mx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    if (mx > 0.0) atomic_max_pos(kmax, mx);
  }
}

// ------------------------------------------------- pivot block: dense BK LDL'
// k_factor_diag: one workgroup of 512 threads per supernode.  The p x p pivot
// block (p <= 128) lives in REGISTERS as a full symmetric matrix: thread
// (tx, ty) = (tid & 31, tid >> 5) owns rows ty+16m (m < 8) and columns tx+32n
// (n < 4), 32 values ("patch").  The elimination is blocked by panels of 16
// columns:
//   1. the owners publish the panel's columns to LDS;
//   2. ONE wavefront eliminates the panel's pivots with its rows in registers
//      (panel_wave below: no barrier and no LDS round trip per pivot);
//   3. all threads apply the panel's rank-16 update to their patches (sweep_patch,
//      restricted to the strips that are still alive).
// A pivot that fails the cheap test ends the panel early and goes through the
// slow path: one pivot with the complete Bunch-Kaufman decision, symmetric
// interchange (rows / columns are exchanged between threads through LDS) and 2x2
// pivots, one barrier per step.
//
// Bunch-Kaufman partial pivoting with the reference's threshold
// alpha = tol (1+sqrt 17)/8 and test order (hqp/spBKP.C:392, 431-438, 471, 480),
// restricted to the pivot block.  A pivot smaller than pert = pivot_eps * max|K|
// is replaced by +-pert (x rows negative, y / slack rows positive); the symbolic
// phase orders zero-diagonal variables so that this does not happen for
// structurally non-singular systems.  Eliminated columns stay unscaled (c = l d)
// until the write-back.  The tail inverts the unit lower L11 in place (MFMA) and
// leaves M = L11^-1 in its own arena: the panel solve and the tree solves are
// products with M.

// lower triangle (incl. diagonal) of a p x p column-major block -> LDS image with
// leading dimension ld, by the 8 wavefronts of k_factor_diag: all loads of a
// thread (up to 16 columns x 2 row halves) are in flight together - one memory
// latency
__device__ __forceinline__ void stage_lower8(const double *__restrict__ P, long long F, int p, int ld,
                                             double *a, int wave, int lane, const DevTree &T,
                                             const double *__restrict__ upd, int node) {
  double v[16][2];
#pragma unroll
  for (int u = 0; u < 16; u++) {
    const int j = wave + 8 * u;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int i = lane + 64 * h;
      v[u][h] = (j < p && i < p && i >= j) ? P[(long long)j * F + i] : 0.0;
    }
  }
  // + the children's update blocks (pulled through the inverse maps, slot order)
  for (int cc = T.child_ptr[node]; cc < T.child_ptr[node + 1]; cc++) {
    const int c = T.child_idx[cc], bc = T.nbor[c];
    const int *iv = T.pinv + T.pinv_off[c];
    const double *Uc = upd + T.upd_off[c];
    int ci[2], cj[16];
#pragma unroll
    for (int h = 0; h < 2; h++) ci[h] = (lane + 64 * h < p) ? iv[lane + 64 * h] : -1;
#pragma unroll
    for (int u = 0; u < 16; u++) cj[u] = (wave + 8 * u < p) ? iv[wave + 8 * u] : -1;
    double g[16][2];
#pragma unroll
    for (int u = 0; u < 16; u++)
#pragma unroll
      for (int h = 0; h < 2; h++)
        g[u][h] = (ci[h] >= 0 && cj[u] >= 0 && lane + 64 * h >= wave + 8 * u) ? Uc[(long long)cj[u] * bc + ci[h]] : 0.0;
#pragma unroll
    for (int u = 0; u < 16; u++)
#pragma unroll
      for (int h = 0; h < 2; h++) v[u][h] += g[u][h];
  }
#pragma unroll
  for (int u = 0; u < 16; u++) {
    const int j = wave + 8 * u;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int i = lane + 64 * h;
      if (j < p && i < p && i >= j) a[i + j * ld] = v[u][h];
    }
  }
}

#define DB 16
#define FD_THREADS 512
#ifndef FD_PANEL
#define FD_PANEL 16
#endif
#define FD_PLD 130  // leading dimension of the LDS panel: columns land in different banks
typedef double PatchT[8][4];  // [row strip m][column strip n]

// Dynamic strip selection is written as chains of selects on VALUES (uniform
// conditions): an if-chain over the strip index gets merged by the optimiser
// into a variably indexed access, which forces the whole patch into scratch.
#define PATCH_PUT_COL_CASE(N)                       \
  {                                                 \
    _Pragma("unroll") for (int m = 0; m < 8; m++) { \
      double v = A[m][N];                           \
      asm volatile("" : "+v"(v));                   \
      if (mine) buf[ty + 16 * m] = v;               \
    }                                               \
  }
__device__ __forceinline__ void patch_put_col(const PatchT &A, int c, int tx, int ty, int p,
                                              double *buf) {
  const int cn = c >> 5;  // wave-uniform
  const bool mine = tx == (c & 31);
  // The empty asm pins each value in a VGPR: without it the optimiser merges the
  // four branches into one variably indexed access and the patch lands in scratch.
  if (cn == 0) PATCH_PUT_COL_CASE(0)
  else if (cn == 1) PATCH_PUT_COL_CASE(1)
  else if (cn == 2) PATCH_PUT_COL_CASE(2)
  else PATCH_PUT_COL_CASE(3)
}
__device__ __forceinline__ void patch_get_col(PatchT &A, int c, int tx, int ty, int p,
                                              const double *buf) {
  const int cn = c >> 5;
  const bool mine = tx == (c & 31);
#pragma unroll
  for (int m = 0; m < 8; m++) {
    const double v = buf[ty + 16 * m];
#pragma unroll
    for (int n = 0; n < 4; n++) A[m][n] = (mine && n == cn) ? v : A[m][n];
  }
}
__device__ __forceinline__ void patch_put_row(const PatchT &A, int r, int tx, int ty, int p,
                                              double *buf) {
  const int rm = r >> 4;
  const bool mine = ty == (r & 15);
#pragma unroll
  for (int n = 0; n < 4; n++) {
    double v = A[0][n];
#pragma unroll
    for (int m = 1; m < 8; m++) {
      const double am = A[m][n];
      v = rm == m ? am : v;
    }
    if (mine) buf[tx + 32 * n] = v;
  }
}
__device__ __forceinline__ void patch_get_row(PatchT &A, int r, int tx, int ty, int p,
                                              const double *buf) {
  const int rm = r >> 4;
  const bool mine = ty == (r & 15);
#pragma unroll
  for (int n = 0; n < 4; n++) {
    const double v = buf[tx + 32 * n];
#pragma unroll
    for (int m = 0; m < 8; m++) A[m][n] = (mine && m == rm) ? v : A[m][n];
  }
}

#ifdef HQPKKT_STAMPS
// instrumented build (tools/): 32-bit s_memtime stamps of block 0 into the spare
// words of the flags buffer (counters = flags + 1)
#define STAMP(slot)                                                                              \
  do {                                                                                           \
    if (blockIdx.x == 0 && threadIdx.x == 0 && p > 64 && (slot) < 50)                           \
      counters[8 + (slot)] = (int)__builtin_amdgcn_s_memtime(), counters[8 + 50] = p;            \
  } while (0)
#ifndef FSTAMP_GRID
#define FSTAMP_GRID 8
#define FSTAMP_BLOCK 0
#endif
#define FSTAMP(slot)                                                                             \
  do {                                                                                           \
    if (FRONT && gridDim.x == FSTAMP_GRID && blockIdx.x == FSTAMP_BLOCK && threadIdx.x == 0)     \
      counters[8 + (slot)] = (int)__builtin_amdgcn_s_memtime(), counters[8 + 50] = p, counters[8 + 51] = b; \
  } while (0)
#else
#define STAMP(slot)
#define FSTAMP(slot)
#endif

// broadcast of one lane's double to the wavefront through SGPRs (src is uniform)
__device__ __forceinline__ double bcast_lane(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// Fast path of k_factor_diag, run by ONE wavefront: up to FD_PANEL consecutive 1x1
// pivots of the LDS panel Pc (16 columns x 128 rows, column-major, leading
// dimension FD_PLD).  Each lane keeps its row(s) of the panel in registers (rows
// lane and lane+64) and the pivot row is broadcast with v_readlane, so a pivot
// costs no LDS round trip and no barrier.  The body is one basic block without
// branches: from the first pivot that fails |a_kk| >= alpha max|column|
// (hqp/spBKP.C:431-438) or is smaller than the perturbation threshold on, the
// eliminations are masked out (multipliers 0), so the panel comes back in the
// state in front of that pivot.  Per pivot the next diagonal entry is updated and
// broadcast first, so that its reciprocal (the dependent chain) overlaps the rest
// of the rank-1 update.
// HI: the pivots are rows >= 64 (the caller never lets a panel straddle row 64);
// rows 0..63 are eliminated already and only rows 64.. are updated.
// Returns the number of pivots done; the panel columns (unscaled) go back to Pc,
// the multipliers to Pl.
template <bool TWO, bool HI>
__device__ __forceinline__ int panel_wave(double *Pc, double *Pl, int k, int kb, int lane, double alpha,
                                          double pert, double *dv, int *pt) {
  // one quarter of the rank-1 update that is still pending from the previous pivot
  // (columns j0, j0+4, ... of the panel): issued between the dependent steps of the
  // current pivot's reciprocal
#define PW_PENDING(Q)                                            \
  _Pragma("unroll") for (int j = kk + 1 + (Q); j < FD_PANEL; j += 4) { \
    if (!HI) R0[j] = fma(-lp0, cp[j], R0[j]);                    \
    if (TWO) R1[j] = fma(-lp1, cp[j], R1[j]);                    \
  }                                                              \
  __builtin_amdgcn_sched_barrier(0)
  double R0[FD_PANEL], R1[FD_PANEL];
#pragma unroll
  for (int j = 0; j < FD_PANEL; j++) {
    R0[j] = Pc[lane + FD_PLD * j];
    R1[j] = TWO ? Pc[lane + 64 + FD_PLD * j] : 0.0;
  }
  int done = kb;        // pivots in front of the first one that failed the test (uniform)
  double mydi = 0.0;
  double cp[FD_PANEL];  // the previous pivot's row (broadcast through SGPRs) and multipliers
  double lp0 = 0.0, lp1 = 0.0;
#pragma unroll
  for (int j = 0; j < FD_PANEL; j++) cp[j] = 0.0;
  double d = bcast_lane(HI ? R1[0] : R0[0], k & 63);
  // one pivot; kk is a compile-time constant so that every register index is static
  auto pivot = [&](auto KK) __attribute__((always_inline)) {
    constexpr int kk = decltype(KK)::value;
    const int kc = k + kk, src = kc & 63;
    // 1/d: hardware estimate (2^-24) and one cubic step x (1 + e + e^2), e = 1 - d x
    double x = __builtin_amdgcn_rcp(d);
    __builtin_amdgcn_sched_barrier(0);
    PW_PENDING(0);
    const double e = fma(-d, x, 1.0);
    __builtin_amdgcn_sched_barrier(0);
    PW_PENDING(1);
    const double e2 = fma(e, e, e);
    __builtin_amdgcn_sched_barrier(0);
    PW_PENDING(2);
    const double di = fma(x, e2, x);
    __builtin_amdgcn_sched_barrier(0);
    PW_PENDING(3);
    // this pivot's row (columns behind the pivot), now that they are up to date
    double c[FD_PANEL];
#pragma unroll
    for (int j = kk + 1; j < FD_PANEL; j++) c[j] = bcast_lane(HI ? R1[j] : R0[j], src);
    // the test (off the dependent chain; no short-circuit operators: they would
    // become branches).  From the first failing pivot on nothing is eliminated.
    const bool r0 = !HI & (lane > kc), r1 = TWO & (HI ? lane + 64 > kc : true);
    const double cmax = fmax(r0 ? fabs(R0[kk]) : 0.0, r1 ? fabs(R1[kk]) : 0.0);
    const bool bad = (int)!(fabs(d) >= alpha * cmax) | (int)!(fabs(d) >= pert);
    done = (__any(bad) && kk < done) ? kk : done;
    const bool act = kk < done;  // uniform
    const double l0 = (act & r0) ? R0[kk] * di : 0.0;
    const double l1 = (act & r1) ? R1[kk] * di : 0.0;
    if constexpr (kk + 1 < FD_PANEL) {  // the next diagonal entry first: it starts the next chain
      if (!HI) R0[kk + 1] = fma(-l0, c[kk + 1], R0[kk + 1]);
      if (TWO) R1[kk + 1] = fma(-l1, c[kk + 1], R1[kk + 1]);
      d = bcast_lane(HI ? R1[kk + 1] : R0[kk + 1], (kc + 1) & 63);
    }
    Pl[lane + FD_PLD * kk] = l0;
    Pl[lane + 64 + FD_PLD * kk] = l1;
    mydi = (lane == kk) ? di : mydi;  // lane kk keeps the inverse pivot until the panel is done
    lp0 = l0, lp1 = l1;
#pragma unroll
    for (int j = kk + 2; j < FD_PANEL; j++) cp[j] = c[j];
    __builtin_amdgcn_sched_barrier(0);
  };
#define PW4(B)                                                   \
  pivot(std::integral_constant<int, (B)>()), pivot(std::integral_constant<int, (B) + 1>()), \
      pivot(std::integral_constant<int, (B) + 2>()), pivot(std::integral_constant<int, (B) + 3>())
  // groups of four pivots: a group is skipped (uniform branch) once a pivot has failed or the
  // panel is shorter - after a slow-path pivot the next panel often fails again soon
  PW4(0);
  if (done > 3 && kb > 4) {
    PW4(4);
    if (done > 7 && kb > 8) {
      PW4(8);
      if (done > 11 && kb > 12) PW4(12);
    }
  }
#undef PW4
#pragma unroll
  for (int j = 0; j < FD_PANEL; j++) {
    if (!HI) Pc[lane + FD_PLD * j] = R0[j];
    if (TWO) Pc[lane + 64 + FD_PLD * j] = R1[j];
  }
  if (lane < done) dv[2 * (k + lane)] = mydi, dv[2 * (k + lane) + 1] = 0.0, pt[k + lane] = 0;
#undef PW_PENDING
  return done;
}

// registers <- registers - sum_{t < done} l_t c_t' for the pivots k..k+done-1 of the
// panel.  Only the strips MLO <= m < MHI of 16 rows (and the matching strips of 32
// columns) take part: below MLO everything is eliminated already, from MHI on the
// patch is the zero padding behind row / column p.
template <int MLO, int MHI>
__device__ __forceinline__ void sweep_patch(PatchT &A, const double *Pc, const double *Pl, int k, int done,
                                            int tx, int ty) {
  constexpr int NLO = MLO >> 1, NHI = (MHI + 1) >> 1;
#pragma unroll 2
  for (int t = 0; t < done; t++) {
    double lt[8], ct[4];
#pragma unroll
    for (int m = MLO; m < MHI; m++) lt[m] = Pl[ty + 16 * m + FD_PLD * t];
#pragma unroll
    for (int n = NLO; n < NHI; n++) ct[n] = Pc[tx + 32 * n + FD_PLD * t];
#pragma unroll
    for (int n = NLO; n < NHI; n++) ct[n] = (tx + 32 * n > k + t) ? ct[n] : 0.0;
#pragma unroll
    for (int m = MLO; m < MHI; m++)
#pragma unroll
      for (int n = NLO; n < NHI; n++) A[m][n] = fma(-lt[m], ct[n], A[m][n]);
  }
}
template <int MHI>
__device__ __forceinline__ void sweep_rows(PatchT &A, const double *Pc, const double *Pl, int k, int done,
                                           int tx, int ty) {
  switch ((k + 1) >> 4) {  // uniform: strips of 16 rows that are completely eliminated
    case 0: sweep_patch<0, MHI>(A, Pc, Pl, k, done, tx, ty); break;
    case 1: if constexpr (MHI > 1) sweep_patch<1, MHI>(A, Pc, Pl, k, done, tx, ty); break;
    case 2: if constexpr (MHI > 2) sweep_patch<2, MHI>(A, Pc, Pl, k, done, tx, ty); break;
    case 3: if constexpr (MHI > 3) sweep_patch<3, MHI>(A, Pc, Pl, k, done, tx, ty); break;
    case 4: if constexpr (MHI > 4) sweep_patch<4, MHI>(A, Pc, Pl, k, done, tx, ty); break;
    case 5: if constexpr (MHI > 5) sweep_patch<5, MHI>(A, Pc, Pl, k, done, tx, ty); break;
    case 6: if constexpr (MHI > 6) sweep_patch<6, MHI>(A, Pc, Pl, k, done, tx, ty); break;
    default: if constexpr (MHI > 7) sweep_patch<7, MHI>(A, Pc, Pl, k, done, tx, ty); break;
  }
}

__global__ void __launch_bounds__(FD_THREADS)
k_factor_diag(DevTree T, const int *__restrict__ level_nodes, double *__restrict__ panel,
              double *__restrict__ dinv, int *__restrict__ ptype, int *__restrict__ lperm,
              const signed char *__restrict__ esign, double *__restrict__ linv,
              const long long *__restrict__ linv_off, double alpha, double pivot_eps,
              const unsigned long long *__restrict__ kmax_bits, int *__restrict__ counters,
              const double *__restrict__ upd) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int node = level_nodes[blockIdx.x];
  const int p = T.npiv[node], b = T.nbor[node];
  const long long F = p + b;
  const int e0 = T.piv_start[node];
  double *P = panel + T.panel_off[node];
  const int ld = p | 1;
  double *a = lds;                  // ld * p: staging at load and write-back
  // 128-entry vectors (the patch is zero padded beyond p, so they are written and
  // read without bounds predicates)
  double *cbuf0 = a + max(ld * p, 2 * FD_PLD * FD_PANEL);  // behind the staging image / panel
  double *cbuf1 = cbuf0 + 128;      // row maxima of the block as assembled (rm0 below)
  double *cbr = cbuf1 + 128;        // column r (second column of a 2x2)
  double *xb0 = cbr + 128, *xb1 = xb0 + 128;  // row / column exchange
  double *dv = xb1 + 128;           // 2p: inverse pivot data per position
  int *lp = (int *)(dv + 2 * p);    // p   local pivot order
  int *pt = lp + p;                 // p   pivot type per position
  int *pdone = pt + p;              // 1   pivots done by the fast path of the current panel
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tx = tid & 31, ty = tid >> 5;
  const double pert = fmax(pivot_eps * __longlong_as_double((long long)*kmax_bits), 1e-300);

  STAMP(0);
  stage_lower8(P, F, p, ld, a, wave, lane, T, upd, node);
  for (int i = tid; i < p; i += blockDim.x) lp[i] = i;
  __syncthreads();
  PatchT A;
#pragma unroll
  for (int m = 0; m < 8; m++)
#pragma unroll
    for (int n = 0; n < 4; n++) {
      const int i = ty + 16 * m, j = tx + 32 * n;
      A[m][n] = (i < p && j < p) ? (i >= j ? a[i + j * ld] : a[j + i * ld]) : 0.0;
    }
  patch_put_col(A, 0, tx, ty, p, cbuf0);
  // largest entry of every row of the block as assembled (children included): what a pivot of a row without a
  // diagonal of its own is measured against (SOFT_PIVOT_REL).  Non-negative doubles order like their bit patterns.
  double *rm0 = cbuf1;
#pragma unroll
  for (int m = 0; m < 8; m++) {  // a row's 128 columns sit in the 32 lanes of one half of a wavefront: DPP max over them
    double v = fmax(fmax(fabs(A[m][0]), fabs(A[m][1])), fmax(fabs(A[m][2]), fabs(A[m][3])));
    v = fmax(v, dpp_move<0xb1, 0xf>(v));   // quad_perm [1,0,3,2]
    v = fmax(v, dpp_move<0x4e, 0xf>(v));   // quad_perm [2,3,0,1]
    v = fmax(v, dpp_move<0x124, 0xf>(v));  // row_ror 4
    v = fmax(v, dpp_move<0x128, 0xf>(v));  // row_ror 8: every lane of a row of 16 holds the row's maximum
    v = fmax(v, dpp_move<0x142, 0xa>(v));  // row_bcast 15 into rows 1 and 3: maximum of the half
    if (tx == 16) rm0[ty + 16 * m] = v;
  }
  __syncthreads();

  // LDS panel of the fast path (aliases the staging image, which is idle in the loop):
  // Pc[i + 128 jj] = current column k+jj (unscaled), Pl[i + 128 jj] = its multipliers
  double *Pc = a, *Pl = a + FD_PLD * FD_PANEL;
  int k = 0;
  bool had2x2 = false;  // block-uniform
  STAMP(1);
  [[maybe_unused]] int npan = 0;  // index of the STAMP slots (instrumented builds, -DHQPKKT_STAMPS)
  while (k < p) {
    k = __builtin_amdgcn_readfirstlane(k);  // wave-uniform: keep it scalar
    // ======== fast path: up to FD_PANEL consecutive 1x1 pivots without interchange ========
    // The panel columns are published to LDS once; each pivot then costs one arg-max
    // search, one rank-1 update of the (<= 16-column) LDS panel and one barrier.  The
    // register patches receive the pivots' rank-1 updates afterwards in one sweep.
    {
      // (a panel does not straddle row 64: the panel wave keeps rows lane / lane+64 apart)
      const int kb = min(min(FD_PANEL, p - k), k < 64 ? 64 - k : FD_PANEL);
#pragma unroll
      for (int n = 0; n < 4; n++) {
        const int jj = tx + 32 * n - k;  // panel slot of this thread's column strip n
        if (jj >= 0 && jj < kb) {
#pragma unroll
          for (int m = 0; m < 8; m++) Pc[ty + 16 * m + FD_PLD * jj] = A[m][n];
        }
      }
      __syncthreads();
      STAMP(2 + 4 * npan);
      if (wave == 0) {
int dn;
        if (p <= 64)
          dn = panel_wave<false, false>(Pc, Pl, k, kb, lane, alpha, pert, dv, pt);
        else if (k < 64)
          dn = panel_wave<true, false>(Pc, Pl, k, kb, lane, alpha, pert, dv, pt);
        else
          dn = panel_wave<true, true>(Pc, Pl, k, kb, lane, alpha, pert, dv, pt);
        if (lane == 0) *pdone = dn;
      }
      STAMP(3 + 4 * npan);
      __syncthreads();
      STAMP(4 + 4 * npan);
      const int done = *pdone;
      const bool slow = done < kb;
      // registers <- registers - sum_t l_t c_t'  (rows / columns <= k+t are masked out;
      // the panel's own columns end up equal to their LDS images)
      switch ((p + 15) >> 4) {  // uniform: strips of 16 rows that hold rows < p
        case 1: sweep_rows<1>(A, Pc, Pl, k, done, tx, ty); break;
        case 2: sweep_rows<2>(A, Pc, Pl, k, done, tx, ty); break;
        case 3: sweep_rows<3>(A, Pc, Pl, k, done, tx, ty); break;
        case 4: sweep_rows<4>(A, Pc, Pl, k, done, tx, ty); break;
        case 5: sweep_rows<5>(A, Pc, Pl, k, done, tx, ty); break;
        case 6: sweep_rows<6>(A, Pc, Pl, k, done, tx, ty); break;
        case 7: sweep_rows<7>(A, Pc, Pl, k, done, tx, ty); break;
        default: sweep_rows<8>(A, Pc, Pl, k, done, tx, ty); break;
      }
      k += done;
      STAMP(5 + 4 * npan);
      npan++;
      __syncthreads();  // the panel is re-used by the next publish
      if (!slow) continue;
    }
    // ======== slow path: one pivot with the complete test, interchanges, 2x2 pivots ========
    if (tid == 0) atomicAdd(&counters[2], 1);  // statistics: pivots that needed the complete test
    double *cur = cbuf0;
    patch_put_col(A, k, tx, ty, p, cur);
    __syncthreads();
    // ---- decision, redundantly per wave --------------------------------------
    // column max and its first row index r (hqp/spBKP.C:431-437): two candidates
    // per lane, the arg-max search in fp32 (DPP max + ballot), lambda re-read in fp64
    const int i1 = k + 1 + lane, i2 = i1 + 64;
    const double v1 = cur[i1 & 127], v2 = cur[i2 & 127], vkk = cur[k];
    const float t1 = i1 < p ? fabsf((float)v1) : -1.0f;
    const float t2 = i2 < p ? fabsf((float)v2) : -1.0f;
    const float tmax = wave_max_dpp_f(fmaxf(fmaxf(t1, t2), 0.0f));
    int r = p;
    {
      const unsigned long long m1 = __ballot(t1 == tmax), m2 = __ballot(t2 == tmax);
      if (m1)
        r = k + 1 + __builtin_ctzll(m1);
      else if (m2)
        r = k + 65 + __builtin_ctzll(m2);
    }
    r = __builtin_amdgcn_readfirstlane(r);
    const double akk = fabs(vkk);
    const double lambda = r < p ? fabs(cur[r]) : 0.0;
    int kind = 0;
    bool have_r = false;
    if (r < p && !(akk >= alpha * lambda)) {  // block-uniform
      patch_put_col(A, r, tx, ty, p, cbr);
      __syncthreads();
      have_r = true;
      double s = 0.0;
      for (int t = k + lane; t < p; t += 64)
        if (t != r) s = fmax(s, fabs(cbr[t]));
      const double sigma = wave_max_dpp(s);
      if (sigma * akk >= alpha * lambda * lambda)
        kind = 0;
      else if (fabs(cbr[r]) >= alpha * sigma)
        kind = 1;
      else
        kind = 2;
    }
    kind = __builtin_amdgcn_readfirstlane(kind);
    // ---- symmetric interchange p1 <-> r of the register matrix ------------------
    const int p1 = (kind == 2) ? k + 1 : k;
    if (kind != 0 && r != p1) {
      patch_put_row(A, p1, tx, ty, p, xb0);
      patch_put_row(A, r, tx, ty, p, xb1);
      __syncthreads();
      patch_get_row(A, p1, tx, ty, p, xb1);
      patch_get_row(A, r, tx, ty, p, xb0);
      __syncthreads();
      patch_put_col(A, p1, tx, ty, p, xb0);
      patch_put_col(A, r, tx, ty, p, xb1);
      __syncthreads();
      patch_get_col(A, p1, tx, ty, p, xb1);
      patch_get_col(A, r, tx, ty, p, xb0);
      if (tid == 0) {
        const int t = lp[p1];
        lp[p1] = lp[r];
        lp[r] = t;
      }
      __syncthreads();
      // refresh the published pivot column(s)
      patch_put_col(A, k, tx, ty, p, cur);
      if (kind == 2) patch_put_col(A, k + 1, tx, ty, p, cbr);
      __syncthreads();
    } else if (kind == 2 && !have_r) {
      patch_put_col(A, k + 1, tx, ty, p, cbr);
      __syncthreads();
    }
    // ---- pivot inverse (identical in every thread) + update of the patch --------
    // row strip m holds rows ty+16m (all <= 16m+15), column strip n columns tx+32n
    if (kind != 2) {
      // all LDS reads first (12 independent ds_reads), then straight-line math
      double cv[8], cj[4];
#pragma unroll
      for (int m = 0; m < 8; m++) cv[m] = cur[ty + 16 * m];
#pragma unroll
      for (int n = 0; n < 4; n++) cj[n] = cur[tx + 32 * n];
      double d = cur[k];
      bool pertd = false;
      if (fabs(d) < SOFT_PIVOT_REL * rm0[lp[k]]) {
        const int sgs = esign[e0 + lp[k]];
        if (sgs == 2 || sgs == -2) {
          counters[4] = 1;  // see SOFT_PIVOT_REL
          if (tiny_replace(counters, d)) d = (sgs < 0 ? -1.0 : 1.0) * fmax(soft_pivot_pert * rm0[lp[k]], pert), pertd = true;
        }
      }
      if (!(fabs(d) >= pert)) {
        const int sg = esign[e0 + lp[k]];
        if (d == 0.0) counters[zero_pivot_slot(sg, sg, b)] = 4;
        d = sg < 0 ? -pert : pert;
        pertd = true;
      }
      const double di = fast_rcp(d);
      if (tid == 0) {
        dv[2 * k] = di, dv[2 * k + 1] = 0.0, pt[k] = 0;
        if (pertd) atomicAdd(&counters[1], 1);
      }
      double lj[4];
#pragma unroll
      for (int n = 0; n < 4; n++) lj[n] = (tx + 32 * n > k) ? cj[n] * di : 0.0;
#pragma unroll
      for (int m = 0; m < 8; m++) {
        const double ck = (ty + 16 * m > k) ? cv[m] : 0.0;
#pragma unroll
        for (int n = 0; n < 4; n++) A[m][n] = fma(-ck, lj[n], A[m][n]);
      }
      k += 1;
    } else {
      double cv1[8], cv2[8], cj1[4], cj2[4];
#pragma unroll
      for (int m = 0; m < 8; m++) cv1[m] = cur[ty + 16 * m], cv2[m] = cbr[ty + 16 * m];
#pragma unroll
      for (int n = 0; n < 4; n++) cj1[n] = cur[tx + 32 * n], cj2[n] = cbr[tx + 32 * n];
      double d11 = cur[k], d21 = cur[k + 1], d22 = cbr[k + 1];
      double det = d11 * d22 - d21 * d21;
      bool pertd = false;
      if (!(fabs(det) >= pert * pert)) {  // degenerate 2x2: perturbed diagonal pair
        const int sg1 = esign[e0 + lp[k]], sg2 = esign[e0 + lp[k + 1]];
        if (det == 0.0) counters[zero_pivot_slot(sg1, sg2, b)] = 4;  // hqp/spBKP.C:731-732
        d11 = sg1 < 0 ? -pert : pert;
        d22 = sg2 < 0 ? -pert : pert;
        d21 = 0.0;
        det = d11 * d22;
        pertd = true;
      }
      had2x2 = true;
      const double rdet = fast_rcp(det);
      const double i11 = d22 * rdet, i21 = -d21 * rdet, i22 = d11 * rdet;
      if (tid == 0) {
        dv[2 * k] = i11, dv[2 * k + 1] = i21, dv[2 * k + 2] = i22, dv[2 * k + 3] = i21;
        pt[k] = 1, pt[k + 1] = 2;
        atomicAdd(&counters[0], 1);
        if (pertd) atomicAdd(&counters[1], 2);
      }
      double l1[4], l2[4];
#pragma unroll
      for (int n = 0; n < 4; n++) {
        const bool on = tx + 32 * n > k + 1;
        const double u1 = on ? cj1[n] : 0.0, u2 = on ? cj2[n] : 0.0;
        l1[n] = u1 * i11 + u2 * i21;
        l2[n] = u1 * i21 + u2 * i22;
      }
#pragma unroll
      for (int m = 0; m < 8; m++) {
        const bool on = ty + 16 * m > k + 1;
        const double c1 = on ? cv1[m] : 0.0, c2 = on ? cv2[m] : 0.0;
#pragma unroll
        for (int n = 0; n < 4; n++) A[m][n] = fma(-c2, l2[n], fma(-c1, l1[n], A[m][n]));
      }
      k += 2;
    }
    __syncthreads();
  }
  STAMP(44);
  // ---- registers -> LDS (lower triangle), scaling the columns on the way: L = C D^-1 ---
  // (columns of 1x1 pivots are scaled here; the rare 2x2 pairs need both columns and
  // are finished in LDS)
  {
    double dj[4];
    int tj[4];
#pragma unroll
    for (int n = 0; n < 4; n++) {
      const int j = min(tx + 32 * n, p - 1);
      dj[n] = dv[2 * j], tj[n] = pt[j];
    }
#pragma unroll
    for (int m = 0; m < 8; m++)
#pragma unroll
      for (int n = 0; n < 4; n++) {
        const int i = ty + 16 * m, j = tx + 32 * n;
        const double v = (i > j && tj[n] == 0) ? A[m][n] * dj[n] : A[m][n];
        if (i < p && j < p && i >= j) a[i + j * ld] = v;
      }
  }
  __syncthreads();
  if (had2x2) {  // block-uniform
    for (int j = wave; j < p; j += FD_THREADS / 64) {
      if (pt[j] != 1) continue;
      const double i11 = dv[2 * j], i21 = dv[2 * j + 1], i22 = dv[2 * j + 2];
      for (int i = j + 2 + lane; i < p; i += 64) {
        const double c1 = a[i + j * ld], c2 = a[i + (j + 1) * ld];
        a[i + j * ld] = c1 * i11 + c2 * i21;
        a[i + (j + 1) * ld] = c1 * i21 + c2 * i22;
      }
      if (lane == 0) a[j + 1 + j * ld] = 0.0;
    }
    __syncthreads();
  }
  STAMP(45);
  // waves 2..7 write L11 and the pivot data back while waves 0..1 invert the 16x16
  // diagonal blocks of L11 (unit lower): one 16-lane group per block, lane c solves
  // L x = e_c column-oriented (15 dependent steps of independent FMAs)
  const int nb = (p + DB - 1) / DB;
  double xinv[DB];
  {
    const int blk = tid >> 4, c = tid & 15;
    if (wave >= 2) {
      for (int j = wave - 2; j < p; j += FD_THREADS / 64 - 2)
        for (int i = j + lane; i < p; i += 64) P[(long long)j * F + i] = a[i + j * ld];
      for (int i = tid - 128; i < p; i += blockDim.x - 128) {
        lperm[e0 + i] = lp[i];
        ptype[e0 + i] = pt[i];
        dinv[2 * (e0 + i)] = dv[2 * i];
        dinv[2 * (e0 + i) + 1] = dv[2 * i + 1];
      }
    } else if (blk < nb) {
      const int kb = blk * DB, kw = min(DB, p - kb);
#pragma unroll
      for (int rr = 0; rr < DB; rr++) xinv[rr] = (rr == c) ? 1.0 : 0.0;
#pragma unroll
      for (int t = 0; t < DB - 1; t++) {
        const double xt = xinv[t];
#pragma unroll
        for (int rr = t + 1; rr < DB; rr++) {
          const double l = (rr < kw) ? a[kb + rr + (kb + t) * ld] : 0.0;
          xinv[rr] = fma(-l, xt, xinv[rr]);
        }
      }
    }
    __syncthreads();  // the write-back has read a[]
    STAMP(46);
    // ---- M = L11^-1 in place (hqp/spBKP.C has no counterpart: it substitutes row by
    // row; the explicit inverse turns the panel solve and the tree solves into
    // products).  The diagonal blocks first ...
    if (wave < 2 && blk < nb) {
      const int kb = blk * DB, kw = min(DB, p - kb);
#pragma unroll
      for (int rr = 0; rr < DB; rr++)
        if (rr < kw && c < kw) a[kb + rr + (kb + c) * ld] = xinv[rr];  // zeros above the diagonal
    }
    __syncthreads();
  }
  // ... then row block i from the rows above it: M_ij = -M_ii sum_{k=j}^{i-1} L_ik M_kj,
  // wave j owns block (i, j): 16x16x4 MFMAs with both operands from the LDS image.
  // The first product's result, in the MFMA C/D layout (lane: column l&15, rows
  // (l>>4)+4q), is exactly the B operand of the second product's four k-slices, so
  // it never leaves the registers; the row block is overwritten after a barrier.
  double *W = linv + linv_off[node];  // p x p, column-major; whole diagonal blocks + below
  for (int i = 1; i < nb; i++) {
    const int ri = DB * i, j = wave;
    const bool have = j < i;  // wave-uniform
    const int ml = lane & 15, kl = lane >> 4;
    double4_t res = {0.0, 0.0, 0.0, 0.0};
    if (!have) {
      // the waves without a block write the previous row block (final) to memory:
      // 16 rows x 16 i columns, four columns of 128 bytes per instruction
      for (int c = 4 * (wave - i) + kl; c < ri; c += 4 * (FD_THREADS / 64 - i))
        W[(long long)c * p + ri - DB + ml] = a[ri - DB + ml + c * ld];
    }
    if (have) {
      double4_t tacc = {0.0, 0.0, 0.0, 0.0};
      const bool rowon = ri + ml < p;
      for (int k = j; k < i; k++) {
        double av[4], bv[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) {
          const int kc = DB * k + 4 * s4 + kl;
          av[s4] = rowon ? a[ri + ml + kc * ld] : 0.0;  // L_ik
          bv[s4] = a[kc + (DB * j + ml) * ld];           // M_kj (k == j: zeros above the diagonal)
        }
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) tacc = mfma_f64(av[s4], bv[s4], tacc);
      }
#pragma unroll
      for (int s4 = 0; s4 < 4; s4++) {
        const int kc = ri + 4 * s4 + kl;
        const double av = (rowon && kc < p) ? a[ri + ml + kc * ld] : 0.0;  // M_ii
        res = mfma_f64(av, tacc[s4], res);
      }
    }
    __syncthreads();
    if (have) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int row = ri + kl + 4 * q;
        if (row < p) a[row + (DB * j + ml) * ld] = -res[q];
      }
    }
    __syncthreads();
  }
  STAMP(47);
  {  // the last row block
    const int rl = DB * (nb - 1), ml = lane & 15, kl = lane >> 4;
    if (rl + ml < p)
      for (int c = 4 * wave + kl; c < p; c += 4 * (FD_THREADS / 64)) W[(long long)c * p + rl + ml] = a[rl + ml + c * ld];
  }
  STAMP(48);
}

// ---- data words as signals (used by the whole-tree sweeps below and by k_solve_top, solve_top.hip.h): a producer
// writes its results with agent-scope atomic stores into an exchange array whose words hold a sentinel (a NaN payload
// no arithmetic produces) until then; the consumer's lanes poll exactly the words they need.  No flags, no cache
// maintenance: one memory round trip per tree level.  The arrays exist twice: solve e uses copy e & 1 and every front
// puts the sentinel back into ITS words of the other copy, which nobody reads during this solve.
static const unsigned long long XW_SENTINEL = 0x7ff8dead0badc0deULL;
static const int XW_GAVE_UP = 110;  // index into the handle's flags buffer: a poll gave up (~2^20 tries)
// tries before a poll gives up.  A device global so that a test can shorten it (HQPKKT_POLL_LIMIT, read when a handle
// uploads its tree): with 0 every poll that has to wait gives up, which forces the fall-back to the per-level launches.
__device__ int xw_poll_limit = 1 << 20;
__device__ __forceinline__ unsigned long long xw_peek(const double *p) {
  return __hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void xw_post(double *p, double v) {
  __hip_atomic_store((unsigned long long *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void xw_clear(double *p) {
  __hip_atomic_store((unsigned long long *)p, XW_SENTINEL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double xw_take(const double *p, int *flags) {
  unsigned long long v = xw_peek(p);
  for (int n = 0; v == XW_SENTINEL; n++) {
    if (n >= xw_poll_limit) {
      __hip_atomic_store(flags + XW_GAVE_UP, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return 0.0;
    }
    __builtin_amdgcn_s_sleep(1);
    v = xw_peek(p);
  }
  return __longlong_as_double((long long)v);
}
// exchange arrays of the whole-tree sweeps of small fronts: contribution vectors (the layout of cb) and solution
// (elimination indices), two copies each; epoch: solves so far (k_rhs_* counts)
struct TreeXchg {
  double *cb, *x;
  long long cb_elems;
  int dim;
  const int *epoch;
  int *flags;
};
// ... and of the whole-tree factorisation of small fronts: the update blocks (layout of the update arena), two
// copies; epoch: factorisations so far (the assembly kernel counts)
struct TreeXchgF {
  double *u;
  long long upd_elems;
  const int *epoch;
};

// ---------------------------------------- pivot block of a small supernode
// Narrow-band systems (Prg_DID: sbw 5) have thousands of supernodes with a
// handful of pivots each; the 512-thread kernel above would spend its time in
// barriers and leave the CU to one node at a time.  For p <= FS_MAXP one
// wavefront factors the block in LDS (full symmetric image, so interchanges are
// plain row + column swaps) and ~18 nodes are resident per CU.  Same pivoting
// rules, same outputs as k_factor_diag.
//
// FRONT = true: the whole front of a "small front" (also at most FS_MAXB border rows):
// the same wavefront first gathers the children's update blocks (extend-add, children
// in slot order), and after the pivot block it does the panel
// solve X = A21 P' M', L21 = X D^-1 and the update U = (gathered) - L21 X' - five
// launches per tree level become one.
#define FS_MAXP 32
#define FS_MAXB 16
// LDS bytes of one front for the leading dimensions of a launch: ldp odd and >= the largest p,
// ldb >= the largest b of the fronts in the launch (a level of a narrow-band tree has
// fronts of 5..20 pivots, a quarter of the room the limits would need)
__host__ __device__ inline size_t fs_lds_bytes(bool front, int ldp, int ldb) {
  const size_t d = (size_t)ldp * ldp + 2 * (size_t)ldp + (front ? 2 * (size_t)ldb * ldp + (size_t)ldb * ldb : 0);
  return 8 * d + 8 * (size_t)ldp;
}
// value of a lane's double in a uniform lane
__device__ __forceinline__ double rdlane(double v, int l) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), l);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// One wavefront per front and no other parallelism: what counts is the length of the
// dependent chains.  Global loads are issued in predicated, fully unrolled batches
// (one memory latency per batch, not one per loop trip), the LDS updates of a pivot
// step likewise, and the registers are capped at 128 so that 16 fronts share a CU.
#ifndef HQPKKT_FDS_WAVES
#define HQPKKT_FDS_WAVES 4
#endif
// TREE (with FRONT): ALL levels of a tree of small fronts in one launch, leaves first; a front's lanes wait for
// the words of its children's update blocks in the exchange array and post their own (see "data words as signals").
template <bool FRONT, bool TREE = false>
__global__ void __launch_bounds__(64, HQPKKT_FDS_WAVES)
k_factor_diag_small(DevTree T, const int *__restrict__ level_nodes, double *__restrict__ panel,
                    double *__restrict__ dinv, int *__restrict__ ptype, int *__restrict__ lperm,
                    const signed char *__restrict__ esign, double *__restrict__ linv,
                    const long long *__restrict__ linv_off, double alpha, double pivot_eps,
                    const unsigned long long *__restrict__ kmax_bits, int *__restrict__ counters,
                    double *__restrict__ upd, double *__restrict__ xar, int ldp, int ldb, TreeXchgF XF) {
  static_assert(FRONT || !TREE, "whole-tree launches are for small fronts");
  if constexpr (TREE) {
    const int par = *XF.epoch & 1;
    upd = XF.u + (long long)par * XF.upd_elems;
    double *other = XF.u + (long long)(par ^ 1) * XF.upd_elems + T.upd_off[level_nodes[blockIdx.x]];
    const int bb = T.nbor[level_nodes[blockIdx.x]];
    for (int t = threadIdx.x; t < bb * bb; t += 64) xw_clear(other + t);
  }
  extern __shared__ __attribute__((aligned(16))) double fsm[];
  double *a = fsm;               // ldp x ldp: full symmetric image of the pivot block
  double *dv = a + ldp * ldp;    // 2 ldp
  // FRONT: border rows of the pivot columns (later L21), X, the update block
  double *s21 = dv + 2 * ldp, *xs = s21 + (FRONT ? ldb * ldp : 0), *ub = xs + (FRONT ? ldb * ldp : 0);
  int *lp = (int *)(ub + (FRONT ? ldb * ldb : 0)), *pt = lp + ldp;
  const int node = level_nodes[blockIdx.x];
  const int p = T.npiv[node], b = T.nbor[node];
  const long long F = p + b;
  const int e0 = T.piv_start[node];
  double *P = panel + T.panel_off[node];
  const int lane = threadIdx.x, i = lane & 31, h = lane >> 5, r16 = lane & 15, cq = lane >> 4;
  const bool row_on = i < p;
  // lanes of the border rows' share of a pivot step (round 6): border row br, and bnq columns at a time - with the two
  // border rows of a leaf sixteen columns per trip, with up to eight rows eight (a trip is two dependent LDS round trips)
  const int brsh = b <= 4 ? 2 : b <= 8 ? 3 : 4, br = lane & ((1 << brsh) - 1), bq = lane >> brsh, bnq = 64 >> brsh;
  FSTAMP(0);
  const double pert = fmax(pivot_eps * __longlong_as_double((long long)*kmax_bits), 1e-300);
  // The columns in groups of 16 (most fronts have one), eight loads per lane and group in
  // flight.  Every unrolled slot sits behind a uniform test of p or b: a front of four
  // pivots executes a quarter of the instructions of one with sixteen (one wavefront per
  // front: the instruction count IS the run time).
  const int ng = p > FS_MAXP / 2 ? 2 : 1;
#pragma unroll 1
  for (int g = 0; g < ng; g++) {
    const int j0 = FS_MAXP / 2 * g;
    double v[FS_MAXP / 4], sv[FRONT ? FS_MAXP / 8 : 1];
#pragma unroll
    for (int u = 0; u < FS_MAXP / 4; u++)
      if (j0 + 2 * u < p) {
        const int j = j0 + 2 * u + h;  // masked lanes read P[0]: no branch around the load
        v[u] = P[(row_on && j < p) ? (long long)min(i, j) * F + max(i, j) : 0];
      }
    if constexpr (FRONT) {
#pragma unroll
      for (int u = 0; u < FS_MAXP / 8; u++)
        if (j0 + 4 * u < p && b > 0) {
          const int c = j0 + 4 * u + cq;
          sv[u] = P[(r16 < b && c < p) ? (long long)c * F + p + r16 : 0];
        }
    }
#pragma unroll
    for (int u = 0; u < FS_MAXP / 4; u++)
      if (j0 + 2 * u < p) {
        const int j = j0 + 2 * u + h;
        if (row_on && j < p) a[i + j * ldp] = v[u];
      }
    if constexpr (FRONT) {
#pragma unroll
      for (int u = 0; u < FS_MAXP / 8; u++)
        if (j0 + 4 * u < p && b > 0) {
          const int c = j0 + 4 * u + cq;
          if (r16 < b && c < p) s21[r16 + ldb * c] = sv[u];
        }
    }
  }
  if constexpr (FRONT)
    for (int t = lane; t < ldb * b; t += 64) ub[t] = 0.0;
  FSTAMP(1);
  if (lane < p) lp[lane] = lane;
  if constexpr (!FRONT) {  // pivot block += children's update blocks (pulled; the border parts are
                           // pulled by the panel solve and the Schur update)
    for (int cc = T.child_ptr[node]; cc < T.child_ptr[node + 1]; cc++) {
      const int c = T.child_idx[cc], bc = T.nbor[c];
      const int *iv = T.pinv + T.pinv_off[c];
      const double *Uc = upd + T.upd_off[c];
      const int civ = lane < p ? iv[lane] : -1;  // the pivot rows in the child's numbering
      const int ci = __shfl(civ, i);
#pragma unroll 1
      for (int g2 = 0; g2 < ng; g2++) {
        const int j0 = FS_MAXP / 2 * g2;
        double g[FS_MAXP / 4];
#pragma unroll
        for (int u = 0; u < FS_MAXP / 4; u++)
          if (j0 + 2 * u < p) {
            const int j = j0 + 2 * u + h, cj = __shfl(civ, j);
            const bool ok = row_on && j < p && ci >= 0 && cj >= 0;
            const double *src = ok ? Uc + ((long long)min(ci, cj) * bc + max(ci, cj)) : P;
            const double t = *src;
            g[u] = ok ? t : 0.0;
          }
#pragma unroll
        for (int u = 0; u < FS_MAXP / 4; u++)
          if (j0 + 2 * u < p) {
            const int j = j0 + 2 * u + h;
            if (row_on && j < p) a[i + j * ldp] += g[u];  // entry (i, j) belongs to this lane alone
          }
      }
    }
  }
  if constexpr (FRONT) {
    // extend-add: the lower triangle of each child's update block in 16 x 16 tiles (lane =
    // row of the tile + 16 * column mod 4), two children's loads in flight; a child's entries
    // go to distinct places (rel is one-to-one), children follow each other in slot order
    const int c0 = T.child_ptr[node], c1 = T.child_ptr[node + 1];
    __syncthreads();
    for (int cc = c0; cc < c1; cc += 2) {
      const bool two = cc + 1 < c1;
      const int ch0 = T.child_idx[cc], ch1 = T.child_idx[two ? cc + 1 : cc];
      const int bc0 = T.nbor[ch0], bc1 = two ? T.nbor[ch1] : 0;  // <= p + b <= 48
      const int *rel0 = T.rel + T.bptr[ch0], *rel1 = T.rel + T.bptr[ch1];
      const double *U0 = upd + T.upd_off[ch0], *U1 = upd + T.upd_off[ch1];
      const int relv0 = lane < bc0 ? rel0[lane] : 0, relv1 = lane < bc1 ? rel1[lane] : 0;
      const int bcm = max(bc0, bc1);
      for (int jb = 0; jb < bcm; jb += 16)
        for (int ib = jb; ib < bcm; ib += 16) {
          const int ii = ib + r16;
          double v0[4], v1[4];
          if constexpr (TREE) {
            // all words of the tile requested together, again while one this lane needs still holds the sentinel
            unsigned long long w0[4], w1[4];
            bool wait = true;
            for (int tries = 0; wait; tries++) {
#pragma unroll
              for (int u = 0; u < 4; u++)
                if (jb + 4 * u < bcm) {
                  const int j = jb + 4 * u + cq;
                  w0[u] = xw_peek(U0 + ((j < bc0 && ii < bc0 && ii >= j) ? (long long)j * bc0 + ii : 0));
                  w1[u] = xw_peek(U1 + ((j < bc1 && ii < bc1 && ii >= j) ? (long long)j * bc1 + ii : 0));
                }
              wait = false;
#pragma unroll
              for (int u = 0; u < 4; u++)
                if (jb + 4 * u < bcm) {
                  const int j = jb + 4 * u + cq;
                  wait = wait || (j < bc0 && ii < bc0 && ii >= j && w0[u] == XW_SENTINEL) ||
                         (j < bc1 && ii < bc1 && ii >= j && w1[u] == XW_SENTINEL);
                }
              if (wait && tries >= xw_poll_limit) {
                __hip_atomic_store(counters - 1 + XW_GAVE_UP, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                wait = false;
              }
              if (wait) __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) v0[u] = __longlong_as_double((long long)w0[u]), v1[u] = __longlong_as_double((long long)w1[u]);
          } else {
#pragma unroll
          for (int u = 0; u < 4; u++)
            if (jb + 4 * u < bcm) {
              const int j = jb + 4 * u + cq;
              v0[u] = U0[(j < bc0 && ii < bc0 && ii >= j) ? (long long)j * bc0 + ii : 0];
              v1[u] = U1[(j < bc1 && ii < bc1 && ii >= j) ? (long long)j * bc1 + ii : 0];
            }
          }
#pragma unroll
          for (int w = 0; w < 2; w++) {
            const int bc = w ? bc1 : bc0;
            const int relv = w ? relv1 : relv0;
            const int ri = __shfl(relv, ii);
#pragma unroll
            for (int u = 0; u < 4; u++)
              if (jb + 4 * u < bc) {
                const int j = jb + 4 * u + cq;
                const int rj = __shfl(relv, j);
                const double v = w ? v1[u] : v0[u];
                if (j < bc && ii < bc && ii >= j) {
                  if (rj >= p)
                    ub[(ri - p) + ldb * (rj - p)] += v;
                  else if (ri >= p)
                    s21[(ri - p) + ldb * rj] += v;
                  else {
                    a[ri + rj * ldp] += v;
                    if (ri != rj) a[rj + ri * ldp] += v;  // the image is the full symmetric matrix
                  }
                }
              }
            __syncthreads();
          }
        }
    }
  }
  __syncthreads();
  // largest entry of every row of the block as assembled (children included): the scale a pivot of a row without
  // a diagonal of its own is measured against (SOFT_PIVOT_REL)
  double rowmax0 = 0.0;
  if (row_on)
    for (int j = h; j < p; j += 2) rowmax0 = fmax(rowmax0, fabs(a[i + j * ldp]));
  rowmax0 = fmax(rowmax0, __shfl_xor(rowmax0, 32));
  FSTAMP(2);
  int k = 0, n2x2 = 0, npert = 0;  // statistics: one atomic per front, not one per pivot (thousands of
                                   // fronts of a level would queue up on the same address)
  while (k < p) {
    FSTAMP(10 + k);
    // column k, row i in both halves; its max below the diagonal and the first row that
    // attains it (hqp/spBKP.C:431-437)
    double ck = row_on ? a[i + k * ldp] : 0.0;
    const bool below = h == 0 && row_on && i > k;
    const float t1 = below ? fabsf((float)ck) : -1.0f;
    const float tmax = wave_max_dpp_f(fmaxf(t1, 0.0f));
    const unsigned long long m1 = __ballot(t1 == tmax);
    int r = m1 ? (int)__builtin_ctzll(m1) : p;
    r = __builtin_amdgcn_readfirstlane(r);
    const double akk = fabs(rdlane(ck, k));
    const double lambda = r < p ? fabs(rdlane(ck, r)) : 0.0;
    int kind = 0;
    if (r < p && !(akk >= alpha * lambda)) {
      const double sv = (h == 0 && row_on && i >= k && i != r) ? fabs(a[i + r * ldp]) : 0.0;
      const double sigma = wave_max_dpp(sv);
      if (sigma * akk >= alpha * lambda * lambda)
        kind = 0;
      else if (fabs(a[r + r * ldp]) >= alpha * sigma)
        kind = 1;
      else
        kind = 2;
    }
    kind = __builtin_amdgcn_readfirstlane(kind);
    const int p1 = (kind == 2) ? k + 1 : k;
    if (kind != 0 && r != p1) {  // symmetric interchange p1 <-> r
      if (lane < p) {
        const double x = a[p1 + lane * ldp], y = a[r + lane * ldp];
        a[p1 + lane * ldp] = y, a[r + lane * ldp] = x;
      }
      __syncthreads();
      if (lane < p) {
        const double x = a[lane + p1 * ldp], y = a[lane + r * ldp];
        a[lane + p1 * ldp] = y, a[lane + r * ldp] = x;
      }
      if (lane == 0) {
        const int t = lp[p1];
        lp[p1] = lp[r], lp[r] = t;
      }
      __syncthreads();
      ck = row_on ? a[i + k * ldp] : 0.0;
    }
    if (kind != 2) {
      double d = rdlane(ck, k);
      bool pertd = false;
      if (fabs(d) < SOFT_PIVOT_REL * rdlane(rowmax0, lp[k])) {
        const int sgs = esign[e0 + lp[k]];
        if (sgs == 2 || sgs == -2) {
          counters[4] = 1;  // see SOFT_PIVOT_REL
          if (tiny_replace(counters, d)) d = (sgs < 0 ? -1.0 : 1.0) * fmax(soft_pivot_pert * rdlane(rowmax0, lp[k]), pert), pertd = true;
        }
      }
      if (!(fabs(d) >= pert)) {
        const int sg = esign[e0 + lp[k]];
        if (d == 0.0) counters[zero_pivot_slot(sg, sg, b)] = 4;
        d = sg < 0 ? -pert : pert;
        pertd = true;
      }
      const double di = fast_rcp(d);
      if (lane == 0) {
        dv[2 * k] = di, dv[2 * k + 1] = 0.0, pt[k] = 0;
        if (pertd) a[k + k * ldp] = d;
      }
      npert += pertd;
      if (row_on && i > k) {
        const double li = ck * di;
        for (int jb = k + 1; jb < p; jb += 8) {
          double akj[4], aij[4];
#pragma unroll
          for (int u = 0; u < 4; u++)
            if (jb + 2 * u < p) {
              const int j = min(jb + 2 * u + h, p - 1);
              akj[u] = a[j + k * ldp], aij[u] = a[i + j * ldp];
            }
#pragma unroll
          for (int u = 0; u < 4; u++)
            if (jb + 2 * u < p) {
              const int j = jb + 2 * u + h;
              if (j < p) a[i + j * ldp] = fma(-li, akj[u], aij[u]);
            }
        }
      }
      if constexpr (FRONT) {
        // the border rows take part in the step like the pivot rows below k (their columns in s21 stay in the order of
        // the front: column lp[j] is the j-th pivot's), and the update block with them: when the last pivot is done it
        // is complete and goes to the parent before anything else (M, L21, the stores) is computed
        if (b > 0) {
          const int pk = lp[k], rc = min(br, b - 1);
          const bool ron = br < b;
          const double lr = s21[rc + ldb * pk] * di;
          for (int jb = k + 1; jb < p; jb += bnq) {
            const int j = min(jb + bq, p - 1), pj = lp[j];
            const double akj = a[j + k * ldp], sv = s21[rc + ldb * pj];
            if (ron && jb + bq < p) s21[br + ldb * pj] = fma(-lr, akj, sv);
          }
          for (int cb = 0; cb < b; cb += bnq) {
            const int c = min(cb + bq, b - 1);
            const double cc = s21[c + ldb * pk], uv = ub[rc + ldb * c];
            if (ron && cb + bq < b && br >= cb + bq) ub[br + ldb * c] = fma(-lr, cc, uv);
          }
        }
      }
      k += 1;
    } else {
      double d11 = rdlane(ck, k), d21 = rdlane(ck, k + 1), d22 = a[k + 1 + (k + 1) * ldp];
      double det = d11 * d22 - d21 * d21;
      bool pertd = false;
      if (!(fabs(det) >= pert * pert)) {  // degenerate 2x2: perturbed diagonal pair
        const int sg1 = esign[e0 + lp[k]], sg2 = esign[e0 + lp[k + 1]];
        if (det == 0.0) counters[zero_pivot_slot(sg1, sg2, b)] = 4;  // hqp/spBKP.C:731-732
        d11 = sg1 < 0 ? -pert : pert;
        d22 = sg2 < 0 ? -pert : pert;
        d21 = 0.0;
        det = d11 * d22;
        pertd = true;
      }
      const double rdet = fast_rcp(det);
      const double i11 = d22 * rdet, i21 = -d21 * rdet, i22 = d11 * rdet;
      if (lane == 0) {
        dv[2 * k] = i11, dv[2 * k + 1] = i21, dv[2 * k + 2] = i22, dv[2 * k + 3] = i21;
        pt[k] = 1, pt[k + 1] = 2;
      }
      n2x2 += 1, npert += pertd ? 2 : 0;
      if (row_on && i > k + 1) {
        const double c1 = ck, c2 = a[i + (k + 1) * ldp];
        const double l1 = c1 * i11 + c2 * i21, l2 = c1 * i21 + c2 * i22;
        for (int jb = k + 2; jb < p; jb += 8) {
          double ak0[4], ak1[4], aij[4];
#pragma unroll
          for (int u = 0; u < 4; u++)
            if (jb + 2 * u < p) {
              const int j = min(jb + 2 * u + h, p - 1);
              ak0[u] = a[j + k * ldp], ak1[u] = a[j + (k + 1) * ldp], aij[u] = a[i + j * ldp];
            }
#pragma unroll
          for (int u = 0; u < 4; u++)
            if (jb + 2 * u < p) {
              const int j = jb + 2 * u + h;
              if (j < p) a[i + j * ldp] = fma(-l2, ak1[u], fma(-l1, ak0[u], aij[u]));
            }
        }
      }
      if constexpr (FRONT) {
        if (b > 0) {
          const int pk0 = lp[k], pk1 = lp[k + 1], rc = min(br, b - 1);
          const bool ron = br < b;
          const double c1 = s21[rc + ldb * pk0], c2 = s21[rc + ldb * pk1];
          const double l1 = c1 * i11 + c2 * i21, l2 = c1 * i21 + c2 * i22;
          for (int jb = k + 2; jb < p; jb += bnq) {
            const int j = min(jb + bq, p - 1), pj = lp[j];
            const double ak0 = a[j + k * ldp], ak1 = a[j + (k + 1) * ldp], sv = s21[rc + ldb * pj];
            if (ron && jb + bq < p) s21[br + ldb * pj] = fma(-l2, ak1, fma(-l1, ak0, sv));
          }
          for (int cb = 0; cb < b; cb += bnq) {
            const int c = min(cb + bq, b - 1);
            const double cc0 = s21[c + ldb * pk0], cc1 = s21[c + ldb * pk1], uv = ub[rc + ldb * c];
            if (ron && cb + bq < b && br >= cb + bq) ub[br + ldb * c] = fma(-l2, cc1, fma(-l1, cc0, uv));
          }
        }
      }
      k += 2;
    }
    __syncthreads();
  }
  FSTAMP(3);
  if constexpr (FRONT) {
    if (b > 0) {  // the update block is complete: to the parent (its lower triangle, column by column)
      double *U = upd + T.upd_off[node];
      for (int cb = 0; cb < b; cb += 4) {
        const int c2 = cb + cq;
        if (r16 < b && c2 < b && r16 >= c2) {
          const double acc = ub[r16 + ldb * c2];
          if constexpr (TREE)
            xw_post(U + (long long)c2 * b + r16, acc);
          else
            U[(long long)c2 * b + r16] = acc;
        }
      }
    }
  }
  if (lane == 0) {
    if (n2x2) atomicAdd(&counters[0], n2x2);
    if (npert) atomicAdd(&counters[1], npert);
  }
  // L = C D^-1 (columns were kept unscaled): every entry below the diagonal on its own, all
  // reads before the first write (a 2x2 pivot mixes two columns)
  {
    const int ic = row_on ? i : 0;
    // a 2x2 pivot on the columns 15 | 16 straddles the two column groups: column 15 unscaled
    const double c15 = a[ic + min(15, p - 1) * ldp];
#pragma unroll 1
    for (int g = 0; g < ng; g++) {
      double nv[FS_MAXP / 4];
#pragma unroll
      for (int u = 0; u < FS_MAXP / 4; u++)
        if (FS_MAXP / 2 * g + 2 * u < p - 1) {  // the last column has nothing below the diagonal
          const int j = min(FS_MAXP / 2 * g + 2 * u + h, p - 1), jn = min(j + 1, p - 1), jp = max(j - 1, 0);
          const int ty = pt[j];
          const double c0 = a[ic + j * ldp], cn = a[ic + jn * ldp], cq0 = a[ic + jp * ldp];
          const double cp = jp == 15 ? c15 : cq0;
          const double d0 = dv[2 * j], d1 = dv[2 * j + 1];
          nv[u] = ty == 0 ? c0 * d0 : ty == 1 ? (i > j + 1 ? c0 * d0 + cn * d1 : 0.0) : cp * d1 + c0 * d0;
        }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < FS_MAXP / 4; u++)
        if (FS_MAXP / 2 * g + 2 * u < p - 1) {
          const int j = FS_MAXP / 2 * g + 2 * u + h;
          if (row_on && j < i) a[i + j * ldp] = nv[u];
        }
      __syncthreads();
    }
  }
#pragma unroll 1
  for (int g = 0; g < ng; g++) {
#pragma unroll
    for (int u = 0; u < FS_MAXP / 4; u++)
      if (FS_MAXP / 2 * g + 2 * u < p) {
        const int j = FS_MAXP / 2 * g + 2 * u + h;
        if (row_on && j <= i) P[(long long)j * F + i] = a[i + j * ldp];
      }
  }
  FSTAMP(4);
  if (lane < p) {
    lperm[e0 + lane] = lp[lane];
    ptype[e0 + lane] = pt[lane];
    dinv[2 * (e0 + lane)] = dv[2 * lane];
    dinv[2 * (e0 + lane) + 1] = dv[2 * lane + 1];
  }
  // inverses of the (at most two) 16x16 diagonal blocks of L11, then M = L11^-1 in
  // place (M_10 = -M_11 L_10 M_00) for the product-form panel solve / tree solves
  const int nb = (p + DB - 1) / DB;
  {
    // lane (blk, c): column c of the inverse of diagonal block blk by substitution; no
    // divergent control flow (clamped reads + selects), lanes of blocks >= nb repeat the last
    const int blk = min(cq, nb - 1), c = r16;
    const int kb = blk * DB, kw = min(DB, p - kb);
    double x[DB];  // row by row: x_rr = [rr == c] - sum_{t < rr} L(rr, t) x_t, two partial sums
    const int kwu = min(DB, p);  // uniform bound of the rows any block has
    x[0] = c == 0 ? 1.0 : 0.0;
#pragma unroll
    for (int rr = 1; rr < DB; rr++) {
      x[rr] = 0.0;
      if (rr < kwu) {
        const int rc = min(rr, kw - 1);
        double s0 = rr == c ? 1.0 : 0.0, s1 = 0.0;
#pragma unroll
        for (int t = 0; t < rr; t++) {
          const double l = a[kb + rc + (kb + min(t, kw - 1)) * ldp];
          if (t & 1)
            s1 = fma(-l, x[t], s1);
          else
            s0 = fma(-l, x[t], s0);
        }
        x[rr] = rr < kw ? s0 + s1 : 0.0;
      }
      // a few rows' reads of L in flight at a time (all 120 hoisted would not fit the registers)
      if (rr == 5 || rr == 8 || rr == 10 || rr == 12 || rr == 14) __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < DB; rr++)
      if (rr < kwu) {
        if (cq < nb && rr < kw && c < kw) a[kb + rr + (kb + c) * ldp] = x[rr];
      }
  }
  __syncthreads();
  FSTAMP(5);
  if (nb == 2) {
    const int kw1 = p - DB, r = r16, c4 = cq * 4;
    const bool rowon = r < kw1;
    double t4[4] = {0.0, 0.0, 0.0, 0.0};
    for (int tt = 0; tt < DB; tt++) {
      const double l = rowon ? a[DB + r + tt * ldp] : 0.0;
#pragma unroll
      for (int q = 0; q < 4; q++) t4[q] = fma(l, a[tt + (c4 + q) * ldp], t4[q]);
    }
    __syncthreads();
    if (rowon) {
#pragma unroll
      for (int q = 0; q < 4; q++) a[DB + r + (c4 + q) * ldp] = t4[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; q++) t4[q] = 0.0;
    for (int sft = 0; sft < kw1; sft++) {
      const double mii = rowon ? a[DB + r + (DB + sft) * ldp] : 0.0;
#pragma unroll
      for (int q = 0; q < 4; q++) t4[q] = fma(mii, a[DB + sft + (c4 + q) * ldp], t4[q]);
    }
    __syncthreads();
    if (rowon) {
#pragma unroll
      for (int q = 0; q < 4; q++) a[DB + r + (c4 + q) * ldp] = -t4[q];
    }
    __syncthreads();
  }
  {
    double *W = linv + linv_off[node];  // p x p, column-major; whole diagonal blocks + below
#pragma unroll 1
    for (int g = 0; g < ng; g++) {
#pragma unroll
      for (int u = 0; u < FS_MAXP / 4; u++)
        if (FS_MAXP / 2 * g + 2 * u < p) {
          const int j = FS_MAXP / 2 * g + 2 * u + h;
          if (row_on && j < p && i >= (j & ~(DB - 1))) W[(long long)j * p + i] = a[i + j * ldp];
        }
    }
  }
  FSTAMP(6);
  if constexpr (FRONT) {
    if (b > 0) {
      const bool ron = r16 < b;
      const int rc = min(r16, b - 1);
      {  // X = A21 P' M' is what the elimination has left in the border rows' columns (unscaled): into pivot order
        double t[FS_MAXP / 4];
#pragma unroll
        for (int u = 0; u < FS_MAXP / 4; u++)
          if (4 * u < p) t[u] = s21[rc + ldb * lp[min(4 * u + cq, p - 1)]];
#pragma unroll
        for (int u = 0; u < FS_MAXP / 4; u++)
          if (4 * u < p) {
            if (ron && 4 * u + cq < p) xs[r16 + ldb * (4 * u + cq)] = t[u];
          }
        __syncthreads();
      }
      FSTAMP(7);
      // L21 = X D^-1, both to memory
      double *X = xar + T.x_off[node];
      {
        double lv[FS_MAXP / 4], xv[FS_MAXP / 4];
#pragma unroll
        for (int u = 0; u < FS_MAXP / 4; u++)
          if (4 * u < p) {
            const int c2 = min(4 * u + cq, p - 1), ty = pt[c2];
            const int kp = ty == 2 ? c2 - 1 : min(c2 + 1, p - 1);
            xv[u] = xs[rc + ldb * c2];
            const double xk = xs[rc + ldb * kp], d0 = dv[2 * c2], d1 = dv[2 * c2 + 1];
            lv[u] = ty == 0 ? xv[u] * d0 : xv[u] * d0 + xk * d1;
          }
#pragma unroll
        for (int u = 0; u < FS_MAXP / 4; u++)
          if (4 * u < p) {
            const int c2 = 4 * u + cq;
            if (ron && c2 < p) {
              X[(long long)c2 * b + r16] = xv[u];
              P[(long long)c2 * F + p + r16] = lv[u];
            }
          }
      }
      __syncthreads();
      FSTAMP(8);
    }
  }
  FSTAMP(9);
}

// ---- tree solves of the small fronts: one wavefront per supernode does what the
// A and B kernels below do for the general fronts (two launches per level, not four).
// Like the factorisation above they are latency chains: every load that depends on the
// node alone (M, L21) is issued before anything else.
// TREE: ALL levels in one launch (trees of small fronts only: the double-integrator structure has a dozen levels
// of thousands of fronts of a few pivots, and a level's launch costs more than its work).  The fronts come in the
// order of the levels, leaves first; a front's lanes wait for the words of its children's contribution vectors in
// the exchange array and post their own.  With more fronts than the chip holds this relies on workgroups being
// dispatched in the order of their index (a waiting front only waits for fronts before it).
template <bool TREE>
__global__ void __launch_bounds__(64)
k_solve_fwd_small(DevTree T, const int *__restrict__ level_nodes, const double *__restrict__ panel,
                  const double *__restrict__ linv, const long long *__restrict__ linv_off,
                  const double *__restrict__ dinv, const int *__restrict__ ptype,
                  const int *__restrict__ lperm, const double *__restrict__ rhs, double *__restrict__ xsol,
                  double *__restrict__ ytmp, double *__restrict__ cb, TreeXchg X) {
  __shared__ double t1[FS_MAXP], tp[FS_MAXP], y[FS_MAXP], cbs[FS_MAXB];
  const int node = level_nodes[blockIdx.x];
  const int par = TREE ? (*X.epoch & 1) : 0;
  if constexpr (TREE) {
    cb = X.cb + (long long)par * X.cb_elems;
    if ((int)threadIdx.x < T.nbor[node]) xw_clear(X.cb + (long long)(par ^ 1) * X.cb_elems + T.cb_off[node] + threadIdx.x);
  }
  const int p = T.npiv[node], b = T.nbor[node];
  const long long F = p + b;
  const int e0 = T.piv_start[node];
  const double *P = panel + T.panel_off[node];
  const double *W = linv + linv_off[node];
  const int lane = threadIdx.x, i = lane & 31, h = lane >> 5, r16 = lane & 15, cq = lane >> 4;
  double wv[FS_MAXP / 2], lv[FS_MAXP / 4];  // M(i, t) for the t of parity h; L21(r16, 4u + cq)
#pragma unroll
  for (int u = 0; u < FS_MAXP / 2; u++) {
    const int t = 2 * u + h;
    const bool ok = i < p && t <= i;
    const double w = W[ok ? (long long)t * p + i : 0];  // masked lanes read W[0]: no branch around the load
    wv[u] = ok ? w : 0.0;
  }
#pragma unroll
  for (int u = 0; u < FS_MAXP / 4; u++) {
    const int c2 = 4 * u + cq;
    const bool ok = r16 < b && c2 < p;
    const double l = P[ok ? (long long)c2 * F + p + r16 : 0];
    lv[u] = ok ? l : 0.0;
  }
  int lpk = 0, pty = 0;
  double pd0 = 0.0, pd1 = 0.0;
  if (lane < p) {
    lpk = lperm[e0 + lane], pty = ptype[e0 + lane];
    pd0 = dinv[2 * (e0 + lane)], pd1 = dinv[2 * (e0 + lane) + 1];
    t1[lane] = rhs[e0 + lane];
  }
  if (lane < FS_MAXB) cbs[lane] = 0.0;
  __syncthreads();
  const int c0 = T.child_ptr[node], c1 = T.child_ptr[node + 1];
  for (int cc = c0; cc < c1; cc += 2) {  // two children's loads in flight (their borders have <= 48 rows)
    const bool two = cc + 1 < c1;
    const int ch0 = T.child_idx[cc], ch1 = T.child_idx[two ? cc + 1 : cc];
    const int bc0 = T.nbor[ch0], bc1 = two ? T.nbor[ch1] : 0;
    const int *rel0 = T.rel + T.bptr[ch0], *rel1 = T.rel + T.bptr[ch1];
    const double *cb0 = cb + T.cb_off[ch0], *cb1 = cb + T.cb_off[ch1];
    const int ri0 = lane < bc0 ? rel0[lane] : -1, ri1 = lane < bc1 ? rel1[lane] : -1;
    double v0, v1;
    if constexpr (TREE) {
      v0 = lane < bc0 ? xw_take(cb0 + lane, X.flags) : 0.0, v1 = lane < bc1 ? xw_take(cb1 + lane, X.flags) : 0.0;
    } else {
      v0 = lane < bc0 ? cb0[lane] : 0.0, v1 = lane < bc1 ? cb1[lane] : 0.0;
    }
    if (ri0 >= 0) {
      if (ri0 < p)
        t1[ri0] += v0;
      else
        cbs[ri0 - p] += v0;
    }
    __syncthreads();
    if (ri1 >= 0) {
      if (ri1 < p)
        t1[ri1] += v1;
      else
        cbs[ri1 - p] += v1;
    }
    __syncthreads();
  }
  if (lane < FS_MAXP) tp[lane] = lane < p ? t1[lpk] : 0.0;
  __syncthreads();
  double acc = 0.0;  // y = M tp: the columns of parity h, then both halves
#pragma unroll
  for (int u = 0; u < FS_MAXP / 2; u++) acc = fma(wv[u], tp[2 * u + h], acc);
  acc += __shfl_xor(acc, 32);
  if (lane < FS_MAXP) y[lane] = lane < p ? acc : 0.0;
  __syncthreads();
  if (lane < p) {
    const int kp = pty == 2 ? lane - 1 : min(lane + 1, p - 1);
    xsol[e0 + lane] = pty == 0 ? y[lane] * pd0 : y[lane] * pd0 + y[kp] * pd1;
    ytmp[e0 + lane] = y[lane];
  }
  double s = 0.0;  // contribution -= L21 y
#pragma unroll
  for (int u = 0; u < FS_MAXP / 4; u++) s = fma(lv[u], y[4 * u + cq], s);
  s += __shfl_xor(s, 16);
  s += __shfl_xor(s, 32);
  if constexpr (TREE) {
    if (lane < b) xw_post(cb + T.cb_off[node] + lane, cbs[lane] - s);
  } else {
    if (lane < b) cb[T.cb_off[node] + lane] = cbs[lane] - s;
  }
}

// TREE: all levels in one launch, root first; the lanes wait for the solution at their border rows (pivots of
// ancestors) in the exchange array and post their own part of it
template <bool TREE>
__global__ void __launch_bounds__(64)
k_solve_bwd_small(DevTree T, const int *__restrict__ level_nodes, const double *__restrict__ panel,
                  const double *__restrict__ linv, const long long *__restrict__ linv_off,
                  const int *__restrict__ lperm, double *__restrict__ xsol, TreeXchg X) {
  __shared__ double x2[FS_MAXB], v[FS_MAXP];
  const int node = level_nodes[blockIdx.x];
  const int par = TREE ? (*X.epoch & 1) : 0;
  if constexpr (TREE) {
    if ((int)threadIdx.x < T.npiv[node]) xw_clear(X.x + (long long)(par ^ 1) * X.dim + T.piv_start[node] + threadIdx.x);
  }
  const int p = T.npiv[node], b = T.nbor[node];
  const long long F = p + b;
  const int e0 = T.piv_start[node];
  const double *P = panel + T.panel_off[node];
  const double *W = linv + linv_off[node];
  const int *bi = T.bidx + T.bptr[node];
  const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
  double pv[FS_MAXB / 2], wv[FS_MAXP / 2];  // L21(2u + h, i); M(2u + h, i)
#pragma unroll
  for (int u = 0; u < FS_MAXB / 2; u++) {
    const int r = 2 * u + h;
    const bool ok = i < p && r < b;
    const double l = P[ok ? (long long)i * F + p + r : 0];
    pv[u] = ok ? l : 0.0;
  }
#pragma unroll
  for (int u = 0; u < FS_MAXP / 2; u++) {
    const int r = 2 * u + h;
    const bool ok = i < p && r >= i && r < p;
    const double w = W[ok ? (long long)i * p + r : 0];
    wv[u] = ok ? w : 0.0;
  }
  const int bil = lane < b ? bi[lane] : 0;
  const int lpk = i < p ? lperm[e0 + i] : 0;
  const double xv = i < p ? xsol[e0 + i] : 0.0;
  if constexpr (TREE) {
    if (lane < FS_MAXB) x2[lane] = lane < b ? xw_take(X.x + (long long)par * X.dim + bil, X.flags) : 0.0;
  } else {
    if (lane < FS_MAXB) x2[lane] = lane < b ? xsol[bil] : 0.0;
  }
  __syncthreads();
  double acc = 0.0;  // v = yd - L21' x(border)
#pragma unroll
  for (int u = 0; u < FS_MAXB / 2; u++) acc = fma(pv[u], x2[2 * u + h], acc);
  acc += __shfl_xor(acc, 32);
  if (lane < FS_MAXP) v[lane] = lane < p ? xv - acc : 0.0;
  __syncthreads();
  double z = 0.0;  // z = M' v, x1 = P' z
#pragma unroll
  for (int u = 0; u < FS_MAXP / 2; u++) z = fma(wv[u], v[2 * u + h], z);
  z += __shfl_xor(z, 32);
  if (lane < p) xsol[e0 + lpk] = z;
  if constexpr (TREE) {
    if (lane < p && T.child_ptr[node + 1] > T.child_ptr[node]) xw_post(X.x + (long long)par * X.dim + e0 + lpk, z);
  }
}

// --------------------------------------------------- panel solve (border rows)
// X = A21 P' L11^-T,  L21 = X D^-1, as a product with the explicit inverse
// M = L11^-1 that k_factor_diag leaves behind: x(i,c) = sum_{t<=c} s(i,t) M(c,t).
// One workgroup per (supernode, 32-row slab); the permuted slab s is staged in LDS
// once, every wavefront owns column tiles of 16 (dealt so that the triangular work
// balances) for both 16-row halves and accumulates them with v_mfma_f64_16x16x4,
// reading M straight from L2 (128-byte rows, shared by all slabs of the node).
// No sequential sweep over column blocks and no barrier inside the product.
// MFMA operand layout (verified by hqpkkt_selftest_mfma): A: lane l holds
// A[l&15][l>>4]; B: B[l>>4][l&15]; C/D: col = l&15, row = (l>>4) + 4*reg.

static const int PS_LD = 33;  // doubles per column of the RESULT slab in LDS (what the launch sizes its LDS by)
#ifdef HQPKKT_STAMPS
// instrumented build: s_memtime of lane 0 of every wavefront of workgroup 0 (the last launch wins: the top level)
__device__ int g_ps_stamps[64];
#define PSSTAMP(j)                                                                                           \
  do {                                                                                                       \
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) g_ps_stamps[16 * (threadIdx.x >> 6) + (j)] = (int)__builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define PSSTAMP(j)
#endif
// Round 6: 16 border rows per workgroup (32 before: half the chain of matrix products per wavefront, twice the
// workgroups - a level of a few fronts costs the latency of ONE workgroup, 25 us per level on C2 before), and ONE compact
// loop over the wavefront's batches (column tile, 64 pivots of its k range):
//  - the operands of the NEXT batch out of M travel while the products of this one run (loads without a condition,
//    addresses clamped into the block, so that the compiler counts them; the sequence goes tile by tile, a tile's k
//    ranges ascending: same bits as before);
//  - a finished tile goes from the accumulator straight into the RESULT image in LDS - a region of its own beside the
//    INPUT image, so no barrier in between and ONE accumulator: the loop's body exists once, a few hundred
//    instructions.  That matters more than anything else here: a version with the products unrolled over the
//    wavefront's four tiles and two register sets (50 KB of code, every line of it executed once per wavefront) spent
//    25 us of its 35 in the products, waiting for INSTRUCTIONS (stamps: tools/stamps_ps.py, profiles/r06_ps_stamps.txt).
// Input image: 16 doubles per pivot (a ds_read_b64 of the MFMA A operand touches k-rows t .. t + 3: 64 consecutive
// doubles); result image: 17 per pivot (the waves store with the column index running over the lanes).
// Workgroups of a launch are dealt round-robin over the eight XCDs (each with an L2 of its own): with the list of work
// items in launch order, the 16 slabs (or the tiles) of ONE front would land on all eight and every XCD would pull
// every front's M (L21, X) through its 4 MB.  (Speed only - nothing depends on where a workgroup runs.  A contiguous
// range of the list per XCD was measured slower than the list order: a level's fronts are listed by falling size.)
// Chunks of `mode` consecutive list entries (about the work of one front) go to one XCD, consecutive chunks to consecutive
// XCDs: workgroup bid of a group of 8 ch takes entry (group, chunk bid % 8, position bid / 8 inside it); the entries
// behind the last whole group keep their number.  mode 0: list order, > 0: chunks of that size
__device__ __forceinline__ int xcd_order(int bid, int nwg, int mode) {
  if (mode <= 0) return bid;
  const int grp = 8 * mode, g = bid / grp;
  if ((g + 1) * grp > nwg) return bid;
  const int within = bid - g * grp;
  return g * grp + mode * (within & 7) + (within >> 3);
}
static const int PS_ROWS = 16;
__global__ void __launch_bounds__(256, 3)
k_panel_solve(DevTree T, const int *__restrict__ slabs, double *__restrict__ panel,
              double *__restrict__ xar, const double *__restrict__ dinv,
              const int *__restrict__ ptype, const int *__restrict__ lperm,
              const double *__restrict__ linv, const long long *__restrict__ linv_off,
              const double *__restrict__ upd, int xcd_mode) {
  constexpr int ROWS = PS_ROWS;
  constexpr int NG = 256 / ROWS;     // column groups of the staging pass
  constexpr int NU = 128 / NG;       // columns per thread and trip of 128
  constexpr int LDO = ROWS + 1;      // result image
  extern __shared__ __attribute__((aligned(16))) double lds[];
  PSSTAMP(0);
  const int bid = xcd_order(blockIdx.x, gridDim.x, xcd_mode);
  const int sidx = bid >> 1;  // (the host lists 32-row slabs: two workgroups each)
  const int node = slabs[2 * sidx], slab = slabs[2 * sidx + 1];
  const int p = T.npiv[node], b = T.nbor[node];
  const int r0 = slab * 32 + 16 * (bid & 1);
  if (r0 >= b || p == 0) return;  // (the second half of a slab of <= 16 rows; a front without pivots has no columns here)
  const long long F = p + b;
  const int e0 = T.piv_start[node];
  double *P = panel + T.panel_off[node];
  double *X = xar + T.x_off[node];
  const double *W = linv + linv_off[node];
  const int tid = threadIdx.x, r = tid & (ROWS - 1), g = tid / ROWS;
  const int wave = tid >> 6, lane = tid & 63;
  const bool live = (r0 + r) < b;
  double *s = lds;                 // input image: ROWS x p
  double *so = s + ROWS * p;       // result image: LDO x p
  // pivot data (type, D^-1) of all columns: to LDS up front, together with the
  // permutation, so that the epilogue has no dependent global loads
  double *pd = so + LDO * p + (p & 1);  // 2 p (16-byte aligned)
  int *pty = (int *)(pd + 2 * p);  // p
  // the children list and the first two children (a dependent chain of scalar loads: requested before everything else)
  const int cc0 = T.child_ptr[node], cc1 = T.child_ptr[node + 1];
  const int chA = T.child_idx[cc0 < cc1 ? cc0 : 0], chB = T.child_idx[cc0 + 1 < cc1 ? cc0 + 1 : 0];
  if (tid < p) {  // (256 threads: fronts of up to 256 pivots)
    pty[tid] = ptype[e0 + tid];
    pd[2 * tid] = dinv[2 * (e0 + tid)];
    pd[2 * tid + 1] = dinv[2 * (e0 + tid) + 1];
  }
  PSSTAMP(1);
  for (int c0 = 0; c0 < p; c0 += 128) {  // 128 columns per trip (fronts of up to 256 pivots: two trips)
    int lc[NU];  // NU columns per thread and trip, loads batched
#pragma unroll
    for (int u = 0; u < NU; u++) {
      const int kcol = c0 + g + NG * u;
      lc[u] = kcol < p ? lperm[e0 + kcol] : 0;
    }
    double v[NU];
#pragma unroll
    for (int u = 0; u < NU; u++) {
      const int kcol = c0 + g + NG * u;
      v[u] = (live && kcol < p) ? P[(long long)lc[u] * F + p + r0 + r] : 0.0;
    }
    // + the children's update blocks at (border row, pivot column): two children at a time - their index maps
    // travel together, then their values (a front of a dissection tree has two: one round trip each instead of two)
    for (int cc = cc0; cc < cc1; cc += 2) {
      const bool two = cc + 1 < cc1;
      const int cA = cc == cc0 ? chA : T.child_idx[cc], cB = two ? (cc == cc0 ? chB : T.child_idx[cc + 1]) : cA;
      const int bcA = T.nbor[cA], bcB = T.nbor[cB];
      const int *ivA = T.pinv + T.pinv_off[cA], *ivB = T.pinv + T.pinv_off[cB];
      const double *UA = upd + T.upd_off[cA], *UB = upd + T.upd_off[cB];
      const int ciA = live ? ivA[p + r0 + r] : -1, ciB = (live && two) ? ivB[p + r0 + r] : -1;
      int cjA[NU], cjB[NU];
#pragma unroll
      for (int u = 0; u < NU; u++) {
        const bool on = c0 + g + NG * u < p;
        cjA[u] = on ? ivA[lc[u]] : -1, cjB[u] = (on && two) ? ivB[lc[u]] : -1;
      }
      double gA[NU], gB[NU];
#pragma unroll
      for (int u = 0; u < NU; u++) {
        gA[u] = (ciA >= 0 && cjA[u] >= 0) ? UA[(long long)cjA[u] * bcA + ciA] : 0.0;
        gB[u] = (ciB >= 0 && cjB[u] >= 0) ? UB[(long long)cjB[u] * bcB + ciB] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < NU; u++) v[u] = (v[u] + gA[u]) + gB[u];  // (children in slot order, as one by one)
    }
#pragma unroll
    for (int u = 0; u < NU; u++) {
      const int kcol = c0 + g + NG * u;
      if (kcol < p) s[r + ROWS * kcol] = v[u];
    }
  }
  PSSTAMP(2);
  __syncthreads();
  PSSTAMP(3);
  const int nbc = (p + 15) >> 4;
  const int ml = lane & 15, kl = lane >> 4;
  // column tiles from the most expensive down, dealt 0 1 2 3 3 2 1 0 0 1 2 3 3 2 1 0 to the waves: slot u of this wave
  auto ct_of = [&](int u) { return nbc - 1 - (8 * (u >> 1) + ((u & 1) == 0 ? wave : 7 - wave)); };
  // the M operands of (column tile ct, k-steps 16 hb .. 16 hb + 15: 64 pivots): M(c, t) = W[t p + c].  NO select on the
  // loaded value (hipcc turns "condition ? loaded : 0" into a branch around the load and waits for every pair of
  // loads): addresses clamped into the tile's rows of M - finite numbers, multiplied by an A operand that is zero
  // for the k-steps behind the tile's range; columns behind the last pivot give results nobody stores.
  auto load_b = [&](double (&bv)[16], int ct, int hb) {
    const int c0 = 16 * ct, tend = min(p, c0 + 16);
    const double *wc = W + min(c0 + ml, p - 1);
#pragma unroll
    for (int q = 0; q < 16; q++) bv[q] = wc[(long long)min(64 * hb + 4 * q + kl, tend - 1) * p];
  };
  {
    double bv[16], bvn[16];
    int u = 0, hb = 0, ct = ct_of(0);
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
    if (ct >= 0) load_b(bvn, ct, 0);
#pragma unroll 1
    while (ct >= 0) {  // (the tiles that exist are a prefix of the wave's slots; at most four: fronts of up to 256 pivots)
#pragma unroll
      for (int q = 0; q < 16; q++) bv[q] = bvn[q];
      const int tend = min(p, 16 * ct + 16);
      // the batch after this one (behind the last one: this one again, from L1, unused)
      const bool same = 64 * (hb + 1) < tend;
      const int un = same ? u : u + 1;
      const int ctn = same ? ct : (un < 4 ? ct_of(un) : -1), hn = same ? hb + 1 : 0;
      load_b(bvn, ctn < 0 ? ct : ctn, ctn < 0 ? hb : hn);
#pragma unroll
      for (int q4 = 0; q4 < 4; q4++) {
        if (64 * hb + 16 * q4 < tend) {  // wave-uniform: 16 pivots at a time (their four LDS reads in flight together)
          double a[4];
#pragma unroll
          for (int qq = 0; qq < 4; qq++) a[qq] = s[ml + ROWS * min(64 * hb + 16 * q4 + 4 * qq + kl, p - 1)];
#pragma unroll
          for (int qq = 0; qq < 4; qq++) {
            const int t = 64 * hb + 16 * q4 + 4 * qq + kl;
            acc = mfma_f64(t < tend ? a[qq] : 0.0, bv[4 * q4 + qq], acc);
          }
        }
      }
      if (!same) {  // the tile is complete: x(row kl + 4 q, column 16 ct + ml)
        const int c = 16 * ct + ml;
        if (c < p) {
#pragma unroll
          for (int q = 0; q < 4; q++) so[kl + 4 * q + LDO * c] = acc[q];
        }
        acc = double4_t{0.0, 0.0, 0.0, 0.0};
      }
      u = un, hb = hn, ct = ctn;
    }
  }
  PSSTAMP(4);
  __syncthreads();
  PSSTAMP(6);
  if (!live) return;
  for (int kcol = g; kcol < p; kcol += NG) {
    const double x = so[r + LDO * kcol];
    const int ty = pty[kcol];
    // partner column of a 2x2 pivot (kcol+1 / kcol-1)
    const int kp = ty == 2 ? kcol - 1 : min(kcol + 1, p - 1);
    const double l = ty == 0 ? x * pd[2 * kcol] : x * pd[2 * kcol] + so[r + LDO * kp] * pd[2 * kcol + 1];
    X[(long long)kcol * b + r0 + r] = x;
    P[(long long)kcol * F + p + r0 + r] = l;
  }
  PSSTAMP(7);
}

// ----------------------------------------------------- Schur update (MFMA f64)
// U(i,j) -= sum_k L21(i,k) X(j,k) on 64x64 tiles of the lower triangle, four
// waves per workgroup, each wave a 32x32 block = 2x2 v_mfma_f64_16x16x4_f64.
// A operand: lane l holds A[l&15][l>>4]; B operand: B[l>>4][l&15];
// C/D: 4 values per lane, col = l&15, row = (l>>4) + 4*reg  (f64 layout).

// WJ = 2: one workgroup per 64 x 64 tile (a wave: 32 x 32); WJ = 1: two workgroups per tile, 64 x 32 each (a wave:
// 32 x 16) - the thin upper levels of a tree, where a level costs the latency of one workgroup (half the chain of
// products per wavefront).  Round 6: the operands of the next 16 pivots travel while the products of these 16 run
// (two register sets, loads without a condition: clamped addresses, as in k_schur_update_big); until then every trip
// of 32 pivots waited for its own loads.  Same products in the same order: same bits.
template <int WJ>
__global__ void __launch_bounds__(256, 3)
k_schur_update(DevTree T, const int *__restrict__ tiles, const double *__restrict__ panel,
               const double *__restrict__ xar, double *__restrict__ upd, int xcd_mode) {
  static_assert(WJ == 1 || WJ == 2, "a wave holds 32 x 16 or 32 x 32");
  const int bid = xcd_order(blockIdx.x, gridDim.x, xcd_mode);
  const int bx = WJ == 2 ? bid : (bid >> 1);
  const int node = tiles[3 * bx], ti = tiles[3 * bx + 1], tj = tiles[3 * bx + 2];
  const int p = T.npiv[node], b = T.nbor[node];
  const long long F = p + b;
  const double *L = panel + T.panel_off[node] + p;  // L21(i,k) = L[k*F + i]
  const double *X = xar + T.x_off[node];            // X(j,k)   = X[k*b + j]
  double *U = upd + T.upd_off[node];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i0 = ti * 64 + (wave >> 1) * 32;
  const int j0 = tj * 64 + (WJ == 2 ? (wave & 1) * 32 : (bid & 1) * 32 + (wave & 1) * 16);
  if (i0 >= b || j0 >= b || i0 + 31 < j0) return;  // wave-uniform
  const int lr = lane & 15, lk = lane >> 4;
  double4_t acc[2][WJ];
#pragma unroll
  for (int x = 0; x < 2; x++)
#pragma unroll
    for (int y = 0; y < WJ; y++) acc[x][y] = (double4_t){0.0, 0.0, 0.0, 0.0};
  // rows / columns beyond the border read the last one (never written back), pivots beyond the last one read it again
  // (zeroed by a select; whole k-steps beyond it are skipped)
  const double *La[2], *Xb[WJ];
#pragma unroll
  for (int x = 0; x < 2; x++) La[x] = L + min(i0 + 16 * x + lr, b - 1);
#pragma unroll
  for (int y = 0; y < WJ; y++) Xb[y] = X + min(j0 + 16 * y + lr, b - 1);
  double av[2][4][2], bv[2][4][WJ];
  auto load = [&](int set, int k0) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int k = k0 + 4 * q + lk, kc = min(k, p - 1);
      const long long ka = (long long)kc * F, kb = (long long)kc * b;
#pragma unroll
      for (int x = 0; x < 2; x++) {
        const double v = La[x][ka];
        av[set][q][x] = k < p ? v : 0.0;
      }
#pragma unroll
      for (int y = 0; y < WJ; y++) {
        const double v = Xb[y][kb];
        bv[set][q][y] = k < p ? v : 0.0;
      }
    }
  };
  auto products = [&](int set, int k0) {
#pragma unroll
    for (int q = 0; q < 4; q++)
      if (k0 + 4 * q < p) {  // wave-uniform
        // D = X-tile * L-tile': the accumulator holds U(i,j) with i along the lanes, so that
        // the writes of U below move 128-byte segments
#pragma unroll
        for (int x = 0; x < 2; x++)
#pragma unroll
          for (int y = 0; y < WJ; y++) acc[x][y] = mfma_f64(bv[set][q][y], av[set][q][x], acc[x][y]);
      }
  };
  // the first operands are requested before the index chase of the children's blocks
  if (p > 0) load(0, 0);
  // the children's contributions to this wave's part of U (there is no extend-add pass and
  // U is not pre-zeroed: this kernel writes every entry once) travel while the products run
  double uold[2][WJ][4];
#pragma unroll
  for (int x = 0; x < 2; x++)
#pragma unroll
    for (int y = 0; y < WJ; y++)
#pragma unroll
      for (int rg = 0; rg < 4; rg++) uold[x][y][rg] = 0.0;
  for (int cc = T.child_ptr[node]; cc < T.child_ptr[node + 1]; cc++) {
    const int c = T.child_idx[cc], bc = T.nbor[c];
    const int *iv = T.pinv + T.pinv_off[c] + p;  // border part of the parent's front
    const double *Uc = upd + T.upd_off[c];
    int ci[2], cj[WJ][4];  // this lane's two rows and its columns in the child's numbering
#pragma unroll
    for (int x = 0; x < 2; x++) ci[x] = (i0 + 16 * x + lr < b) ? iv[i0 + 16 * x + lr] : -1;
#pragma unroll
    for (int y = 0; y < WJ; y++)
#pragma unroll
      for (int rg = 0; rg < 4; rg++) cj[y][rg] = (j0 + 16 * y + lk + 4 * rg < b) ? iv[j0 + 16 * y + lk + 4 * rg] : -1;
    double gv[2][WJ][4];
#pragma unroll
    for (int x = 0; x < 2; x++)
#pragma unroll
      for (int y = 0; y < WJ; y++)
#pragma unroll
        for (int rg = 0; rg < 4; rg++)
          gv[x][y][rg] = (ci[x] >= 0 && cj[y][rg] >= 0 && ci[x] >= cj[y][rg])
                             ? Uc[(long long)cj[y][rg] * bc + ci[x]] : 0.0;
#pragma unroll
    for (int x = 0; x < 2; x++)
#pragma unroll
      for (int y = 0; y < WJ; y++)
#pragma unroll
        for (int rg = 0; rg < 4; rg++) uold[x][y][rg] += gv[x][y][rg];
  }
  for (int k0 = 0; k0 < p; k0 += 32) {
    load(1, k0 + 16);
    products(0, k0);
    load(0, k0 + 32);
    products(1, k0 + 16);
  }
#pragma unroll
  for (int x = 0; x < 2; x++)
#pragma unroll
    for (int y = 0; y < WJ; y++)
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        const int i = i0 + 16 * x + lr, j = j0 + 16 * y + lk + 4 * rg;
        if (i < b && j < b && i >= j) U[(long long)j * b + i] = uold[x][y][rg] - acc[x][y][rg];
      }
}

// The same update on 128x128 tiles for fronts with borders in the thousands (the top of the tree of an
// irregular graph: the separators of a mesh with far couplings): each wave holds a 64x64 block = 4x4
// MFMA tiles, so an operand fragment fetched from L2 feeds four products instead of two - this kernel
// is bound by the operand traffic, not by HBM (the panel of 160 pivots gives 20 flop per byte of U).
// The operands of the next eight pivots travel while the products of these eight run (two register
// sets); the children's contributions are gathered after the products, straight into the accumulators.
// WX x WY wavefronts, each holding TX x TY MFMA tiles (16 x 16): a workgroup tile of 16 WX TX rows x 16 WY TY columns
// (128 x 128); KT k-steps (of four pivots) per register set; OCC wavefronts per SIMD.  The instance in use is
// <2, 2, 4, 4, 2, 2>: with eight wavefronts of 64 x 32 at four per SIMD the registers spill inside the loop (114 ms
// against 104 on the 1e6-cell mesh with far couplings; the products alone run at 66 TFLOP/s, the blocks' traffic -
// 197 GB there - at 3.9 TB/s, and two workgroups per CU overlap the two by about a third).
template <int WX, int WY, int TX, int TY, int KT, int OCC>
__global__ void __launch_bounds__(64 * WX * WY, OCC)
k_schur_update_big(DevTree T, const int *__restrict__ tiles, const double *__restrict__ panel,
                   const double *__restrict__ xar, double *__restrict__ upd) {
  static_assert(WX * TX == 8 && WY * TY == 8, "128 x 128 tiles (Analysis::run lists them)");
  const int node = tiles[3 * blockIdx.x], ti = tiles[3 * blockIdx.x + 1],
            tj = tiles[3 * blockIdx.x + 2];
  const int p = T.npiv[node], b = T.nbor[node];
  const long long F = p + b;
  const double *L = panel + T.panel_off[node] + p;  // L21(i,k) = L[k*F + i]
  const double *X = xar + T.x_off[node];            // X(j,k)   = X[k*b + j]
  double *U = upd + T.upd_off[node];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i0 = ti * 128 + (wave / WY) * (16 * TX), j0 = tj * 128 + (wave % WY) * (16 * TY);
  if (i0 >= b || j0 >= b || i0 + 16 * TX - 1 < j0) return;  // wave-uniform
  const int lr = lane & 15, lk = lane >> 4;
  double4_t acc[TX][TY];
#pragma unroll
  for (int x = 0; x < TX; x++)
#pragma unroll
    for (int y = 0; y < TY; y++) acc[x][y] = (double4_t){0.0, 0.0, 0.0, 0.0};
  // rows / columns beyond the border read the last one (never written back) and pivots beyond the last one read
  // it again (zeroed by a select; whole k-steps beyond it are skipped): no predicated loads in the loop
  const double *La[TX], *Xb[TY];
#pragma unroll
  for (int x = 0; x < TX; x++) La[x] = L + min(i0 + 16 * x + lr, b - 1);
#pragma unroll
  for (int y = 0; y < TY; y++) Xb[y] = X + min(j0 + 16 * y + lr, b - 1);
  double av[2][KT][TX], bv[2][KT][TY];
  auto load = [&](int set, int k0) {
#pragma unroll
    for (int q = 0; q < KT; q++) {
      const int k = k0 + 4 * q + lk, kc = min(k, p - 1);
      const long long ka = (long long)kc * F, kb = (long long)kc * b;
#pragma unroll
      for (int x = 0; x < TX; x++) {
        const double v = La[x][ka];
        av[set][q][x] = k < p ? v : 0.0;
      }
#pragma unroll
      for (int y = 0; y < TY; y++) {
        const double v = Xb[y][kb];
        bv[set][q][y] = k < p ? v : 0.0;
      }
    }
  };
  auto products = [&](int set, int k0) {
#pragma unroll
    for (int q = 0; q < KT; q++)
      if (k0 + 4 * q < p) {  // wave-uniform
#pragma unroll
        for (int x = 0; x < TX; x++)
#pragma unroll
          for (int y = 0; y < TY; y++) acc[x][y] = mfma_f64(bv[set][q][y], av[set][q][x], acc[x][y]);
      }
  };
  if (p > 0) load(0, 0);  // (an empty supernode - the test hook HQPKKT_SCHUR_BIG_B=1 sends those here too - has no pivot to clamp to)
  for (int k0 = 0; k0 < p; k0 += 8 * KT) {
    load(1, k0 + 4 * KT);
    products(0, k0);
    load(0, k0 + 8 * KT);
    products(1, k0 + 4 * KT);
  }
  // acc <- acc - (children's contributions); U = -acc.  One row block of the wave's tile at a time: its 16 x TY values
  // of every child, then its part of U (the index maps are re-read per row block: they are small and cached)
#pragma unroll
  for (int x = 0; x < TX; x++) {
    const int i = i0 + 16 * x + lr;
    for (int cc = T.child_ptr[node]; cc < T.child_ptr[node + 1]; cc++) {
      const int c = T.child_idx[cc], bc = T.nbor[c];
      const int *iv = T.pinv + T.pinv_off[c] + p;  // border part of the parent's front
      const double *Uc = upd + T.upd_off[c];
      const int ci = i < b ? iv[i] : -1;
      int cj[TY][4];
#pragma unroll
      for (int y = 0; y < TY; y++)
#pragma unroll
        for (int rg = 0; rg < 4; rg++) cj[y][rg] = (j0 + 16 * y + lk + 4 * rg < b) ? iv[j0 + 16 * y + lk + 4 * rg] : -1;
      double gv[TY][4];
#pragma unroll
      for (int y = 0; y < TY; y++)
#pragma unroll
        for (int rg = 0; rg < 4; rg++)
          gv[y][rg] = (ci >= 0 && cj[y][rg] >= 0 && ci >= cj[y][rg]) ? Uc[(long long)cj[y][rg] * bc + ci] : 0.0;
#pragma unroll
      for (int y = 0; y < TY; y++)
#pragma unroll
        for (int rg = 0; rg < 4; rg++) acc[x][y][rg] -= gv[y][rg];
    }
#pragma unroll
    for (int y = 0; y < TY; y++)
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        const int j = j0 + 16 * y + lk + 4 * rg;
        if (i < b && j < b && i >= j) U[(long long)j * b + i] = -acc[x][y][rg];
      }
  }
}

// MFMA layout self-test: C(16x16) = A(16x16) * B(16x16), all row-major
__global__ void k_mfma_selftest(const double *A, const double *B, double *C) {
  const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < 16; k0 += 4)
    acc = mfma_f64(A[lr * 16 + k0 + lk], B[(k0 + lk) * 16 + lr], acc);
  for (int rg = 0; rg < 4; rg++) C[(lk + 4 * rg) * 16 + lr] = acc[rg];
}

// ------------------------------------------------------------------ solves
// The sweeps over the assembly tree run level by level:
//   forward    per (supernode, 64-row slab): t = rhs(pivots) + children contributions,
//              y = M P t with M = L11^-1 (every slab on its own), xsol = D^-1 y (slab 0),
//              contribution(slab) = gathered - L21(slab,:) y            -- one launch
//   backward B per 16 columns  v = yd - L21' x(border)
//            A per supernode   x1 = P' M' v                             -- two launches
// The L21 products are spread over many workgroups because they carry the bytes
// (nnz(L) is streamed once per sweep).
// y(16-row block ib) = sum_{tb <= ib} M(ib, tb) t(tb) with M = L11^-1 read straight
// from L2: the matrix-vector product runs on the MFMA pipe with the vector in
// column 0 of the B operand (the flops are free, the loads are exactly the lower
// triangle of M, all in flight together).  Row blocks are dealt 0 1 2 3 3 2 1 0.
__device__ __forceinline__ void mfma_lower_times_vec(const double *__restrict__ W, int p, const double *tp,
                                                     double *y, int wave, int lane) {
  const int nb = (p + DB - 1) / DB, ml = lane & 15, kl = lane >> 4;
#pragma unroll 1
  for (int u = 0; u < 4; u++) {  // (fronts of up to 256 pivots: 16 row blocks, four per wave)
    const int ib = nb - 1 - (8 * (u >> 1) + ((u & 1) == 0 ? wave : 7 - wave));
    if (ib < 0) continue;  // wave-uniform
    const int i0 = DB * ib, tend = min(p, i0 + DB);
    const bool ron = i0 + ml < p;
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
    for (int t0 = 0; t0 < tend; t0 += 128) {  // 32 k-steps (128 pivots) of operands in flight together
      double av[32];
#pragma unroll
      for (int q = 0; q < 32; q++) {
        const int t = t0 + 4 * q + kl;
        av[q] = (ron && t < tend) ? W[(long long)t * p + i0 + ml] : 0.0;
      }
#pragma unroll
      for (int q = 0; q < 32; q++) {
        if (t0 + 4 * q < tend) {  // wave-uniform
          const int t = t0 + 4 * q + kl;
          const double bv = (ml == 0 && t < tend) ? tp[t] : 0.0;
          acc = mfma_f64(av[q], bv, acc);
        }
      }
    }
    if (ml == 0) {
#pragma unroll
      for (int q = 0; q < 4; q++)
        if (i0 + kl + 4 * q < p) y[i0 + kl + 4 * q] = acc[q];
    }
  }
}

// z(16-column block tb) = sum_{ib >= tb} M(ib, tb)' v(ib): the transposed product, the
// vector in row 0 of the A operand
__device__ __forceinline__ void mfma_lower_trans_times_vec(const double *__restrict__ W, int p,
                                                           const double *v, double *z, int wave, int lane) {
  const int nb = (p + DB - 1) / DB, ml = lane & 15, kl = lane >> 4;
#pragma unroll 1
  for (int u = 0; u < 4; u++) {
    const int tb = 8 * (u >> 1) + ((u & 1) == 0 ? wave : 7 - wave);  // column block 0 is the longest
    if (tb >= nb) continue;                                            // wave-uniform
    const int t0 = DB * tb;
    const bool con = t0 + ml < p;
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
    for (int i0 = t0; i0 < p; i0 += 128) {
      double bv[32];
#pragma unroll
      for (int q = 0; q < 32; q++) {
        const int i = i0 + 4 * q + kl;
        bv[q] = (con && i < p) ? W[(long long)(t0 + ml) * p + i] : 0.0;
      }
#pragma unroll
      for (int q = 0; q < 32; q++) {
        if (i0 + 4 * q < p) {  // wave-uniform
          const int i = i0 + 4 * q + kl;
          const double av = (ml == 0 && i < p) ? v[i] : 0.0;
          acc = mfma_f64(av, bv[q], acc);
        }
      }
    }
    if (kl == 0 && con) z[t0 + ml] = acc[0];
  }
}

// forward step in two launches (levels with thousands of slabs: M is read once per front,
// not once per slab): A gathers and computes y, B the slabs' L21 y
__global__ void __launch_bounds__(256)
k_solve_fwd_a(DevTree T, const int *__restrict__ level_nodes, const double *__restrict__ linv,
              const long long *__restrict__ linv_off, const double *__restrict__ dinv,
              const int *__restrict__ ptype, const int *__restrict__ lperm,
              const double *__restrict__ rhs, double *__restrict__ xsol, double *__restrict__ ytmp,
              double *__restrict__ cb) {
  __shared__ double t1[256], tp[256], y[256];
  const int node = level_nodes[blockIdx.x];
  const int p = T.npiv[node], b = T.nbor[node];
  const int e0 = T.piv_start[node];
  const double *W = linv + linv_off[node];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  double *cbn = cb + T.cb_off[node];
  // this thread's pivot (tid < p): permutation and pivot data, loaded up front
  int lpk = 0, pty = 0;
  double pd0 = 0.0, pd1 = 0.0;
  if (tid < p) {
    lpk = lperm[e0 + tid], pty = ptype[e0 + tid];
    pd0 = dinv[2 * (e0 + tid)], pd1 = dinv[2 * (e0 + tid) + 1];
    t1[tid] = rhs[e0 + tid];
  }
  for (int i = tid; i < b; i += blockDim.x) cbn[i] = 0.0;
  __syncthreads();
  for (int cc = T.child_ptr[node]; cc < T.child_ptr[node + 1]; cc++) {
    const int c = T.child_idx[cc];
    const int bc = T.nbor[c];
    const int *rel = T.rel + T.bptr[c];
    const double *cbc = cb + T.cb_off[c];
    for (int i = tid; i < bc; i += blockDim.x) {
      const int ri = rel[i];
      if (ri < p)
        t1[ri] += cbc[i];
      else
        cbn[ri - p] += cbc[i];
    }
    __syncthreads();
  }
  if (tid < p) tp[tid] = t1[lpk];
  __syncthreads();
  mfma_lower_times_vec(W, p, tp, y, wave, lane);
  __syncthreads();
  if (tid < p) {
    const int kp = pty == 2 ? tid - 1 : min(tid + 1, p - 1);  // partner of a 2x2 pivot
    xsol[e0 + tid] = pty == 0 ? y[tid] * pd0 : y[tid] * pd0 + y[kp] * pd1;
    ytmp[e0 + tid] = y[tid];
  }
}

// contribution(slab) -= L21(slab,:) y ; 64 rows per workgroup, the four waves
// split the columns and their partial sums meet in LDS
__global__ void __launch_bounds__(256)
k_solve_fwd_b(DevTree T, const int *__restrict__ gslabs, const double *__restrict__ panel,
              const double *__restrict__ ytmp, double *__restrict__ cb) {
  __shared__ double part[4][64];
  __shared__ double ysh[256];
  const int node = gslabs[2 * blockIdx.x], slab = gslabs[2 * blockIdx.x + 1];
  const int p = T.npiv[node], b = T.nbor[node];
  const long long F = p + b;
  const int e0 = T.piv_start[node];
  const double *L = panel + T.panel_off[node] + p;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i = slab * 64 + lane;
  for (int k = tid; k < p; k += blockDim.x) ysh[k] = ytmp[e0 + k];
  __syncthreads();
  double acc = 0.0;
  if (i < b) {
    double a8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int k0 = wave; k0 < p; k0 += 32) {  // 8 columns of this wave per trip, loads in flight together
      double l[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int k = k0 + 4 * u;
        l[u] = k < p ? L[(long long)k * F + i] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int k = k0 + 4 * u;
        a8[u] += k < p ? l[u] * ysh[k] : 0.0;
      }
    }
    acc = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
  }
  part[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && i < b)
    cb[T.cb_off[node] + i] -= (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

// v(16 columns) = yd - L21(:, cols)' x(border)
__global__ void __launch_bounds__(256)
k_solve_bwd_b(DevTree T, const int *__restrict__ cblks, const double *__restrict__ panel,
              const double *__restrict__ xsol, double *__restrict__ vtmp) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int node = cblks[2 * blockIdx.x], cblk = cblks[2 * blockIdx.x + 1];
  const int p = T.npiv[node], b = T.nbor[node];
  const long long F = p + b;
  const int e0 = T.piv_start[node];
  const double *P = panel + T.panel_off[node];
  const int *bi = T.bidx + T.bptr[node];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  double *x2 = lds;  // b
  for (int i = tid; i < b; i += blockDim.x) x2[i] = xsol[bi[i]];
  __syncthreads();
  {
    // this wave's four columns at once; per trip 4 x 4 independent loads / sums
    double acc[4][4];
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
      for (int u = 0; u < 4; u++) acc[c][u] = 0.0;
    for (int i0 = lane; i0 < b; i0 += 256) {
      double l[4][4], xv[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int i = i0 + 64 * u;
        xv[u] = i < b ? x2[i] : 0.0;
#pragma unroll
        for (int c = 0; c < 4; c++) {
          const int k = cblk * 16 + wave + 4 * c;
          l[c][u] = (i < b && k < p) ? P[(long long)k * F + p + i] : 0.0;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int c = 0; c < 4; c++) acc[c][u] += l[c][u] * xv[u];
    }
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const int k = cblk * 16 + wave + 4 * c;
      const double t = wave_sum((acc[c][0] + acc[c][1]) + (acc[c][2] + acc[c][3]));
      if (lane == 0 && k < p) vtmp[e0 + k] = xsol[e0 + k] - t;
    }
  }
}

// x1 = P' L11^-T v
__global__ void __launch_bounds__(256)
k_solve_bwd_a(DevTree T, const int *__restrict__ level_nodes, const double *__restrict__ linv,
              const long long *__restrict__ linv_off, const int *__restrict__ lperm,
              const double *__restrict__ vtmp, double *__restrict__ xsol) {
  __shared__ double v[256], z[256];
  const int node = level_nodes[blockIdx.x];
  const int p = T.npiv[node];
  const int e0 = T.piv_start[node];
  const double *W = linv + linv_off[node];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  int lpk = 0;
  if (tid < p) lpk = lperm[e0 + tid], v[tid] = vtmp[e0 + tid];
  __syncthreads();
  mfma_lower_trans_times_vec(W, p, v, z, wave, lane);
  __syncthreads();
  if (tid < p) xsol[e0 + lpk] = z[tid];
}

// ---- forward step of the general fronts, one launch per level: a tree level of a banded
// system is a handful of fronts whose cost IS the launches.  One workgroup per (front,
// 64-row slab); every slab gathers the children's contributions and recomputes
// y = M P t (the p x p product runs on the MFMA pipe, M comes from L2), then owns its
// rows of the contribution c(slab) - L21(slab,:) y; the slab's part of L21 is requested
// before anything else.  Slab 0 also stores D^-1 y.
__global__ void __launch_bounds__(256)
k_solve_fwd(DevTree T, const int *__restrict__ gslabs, const double *__restrict__ panel,
            const double *__restrict__ linv, const long long *__restrict__ linv_off,
            const double *__restrict__ dinv, const int *__restrict__ ptype, const int *__restrict__ lperm,
            const double *__restrict__ rhs, double *__restrict__ xsol, double *__restrict__ ytmp,
            double *__restrict__ cb) {
  __shared__ double t1[256], tp[256], y[256], cbs[64], part[4][64];
  const int node = gslabs[2 * blockIdx.x], slab = gslabs[2 * blockIdx.x + 1];
  const int p = T.npiv[node], b = T.nbor[node];
  const long long F = p + b;
  const int e0 = T.piv_start[node];
  const double *W = linv + linv_off[node];
  const double *L = panel + T.panel_off[node] + p;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i = slab * 64 + lane;
  double l[32];  // L21(i, wave + 4u) of the first 128 pivots (requested before anything else)
#pragma unroll
  for (int u = 0; u < 32; u++) {
    const int k = wave + 4 * u;
    l[u] = L[(i < b && k < p) ? (long long)k * F + i : -(long long)p];  // masked lanes read the panel's first entry
  }
  int lpk = 0, pty = 0;
  double pd0 = 0.0, pd1 = 0.0;
  if (tid < p) {
    lpk = lperm[e0 + tid], pty = ptype[e0 + tid];
    pd0 = dinv[2 * (e0 + tid)], pd1 = dinv[2 * (e0 + tid) + 1];
    t1[tid] = rhs[e0 + tid];
  }
  if (tid < 64) cbs[tid] = 0.0;
  __syncthreads();
  const int r0 = p + 64 * slab;
  for (int cc = T.child_ptr[node]; cc < T.child_ptr[node + 1]; cc++) {
    const int c = T.child_idx[cc];
    const int bc = T.nbor[c];
    const int *rel = T.rel + T.bptr[c];
    const double *cbc = cb + T.cb_off[c];
    for (int j = tid; j < bc; j += blockDim.x) {
      const int ri = rel[j];
      if (ri < p)
        t1[ri] += cbc[j];
      else if (ri >= r0 && ri < r0 + 64)
        cbs[ri - r0] += cbc[j];
    }
    __syncthreads();
  }
  if (tid < p) tp[tid] = t1[lpk];
  __syncthreads();
  mfma_lower_times_vec(W, p, tp, y, wave, lane);
  __syncthreads();
  if (slab == 0 && tid < p) {
    const int kp = pty == 2 ? tid - 1 : min(tid + 1, p - 1);  // partner of a 2x2 pivot
    xsol[e0 + tid] = pty == 0 ? y[tid] * pd0 : y[tid] * pd0 + y[kp] * pd1;
    ytmp[e0 + tid] = y[tid];
  }
  double a4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int u = 0; u < 32; u++) {
    const int k = wave + 4 * u;
    if (4 * u < p) a4[u & 3] += (i < b && k < p) ? l[u] * y[k] : 0.0;
  }
  if (p > 128) {  // block-uniform: the pivots behind the first 128
    double l2[32];
#pragma unroll
    for (int u = 0; u < 32; u++) {
      const int k = 128 + wave + 4 * u;
      l2[u] = L[(i < b && k < p) ? (long long)k * F + i : -(long long)p];
    }
#pragma unroll
    for (int u = 0; u < 32; u++) {
      const int k = 128 + wave + 4 * u;
      if (128 + 4 * u < p) a4[u & 3] += (i < b && k < p) ? l2[u] * y[k] : 0.0;
    }
  }
  part[wave][lane] = (a4[0] + a4[1]) + (a4[2] + a4[3]);
  __syncthreads();
  if (wave == 0 && i < b)
    cb[T.cb_off[node] + i] = cbs[lane] - ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]));
}

// ------------------------------------------------------------ step pre/post
// FULL: rhs = P [r1; r2; (r3 + r4./z) .* scale]   (hqp/Hqp_IpSpBKP.C:196-204)
__global__ void k_rhs_full(int n, int me, int m, const int *__restrict__ q2e,
                           const double *__restrict__ sc, const double *__restrict__ z,
                           const double *__restrict__ r1, const double *__restrict__ r2,
                           const double *__restrict__ r3, const double *__restrict__ r4,
                           double *__restrict__ rhs, int *__restrict__ epoch) {
  int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q == 0 && epoch) *epoch += 1;  // the solve counter of the whole-tree sweeps (solve_top.hip.h)
  if (q >= n + me + m) return;
  double v;
  if (q < n)
    v = r1[q];
  else if (q < n + me)
    v = r2[q - n];
  else {
    int j = q - n - me;
    v = (r4[j] / z[j] + r3[j]) * sc[q];
  }
  rhs[q2e[q]] = v;
}

// CSR row product with value indirection: sum_k vals[src[k]] * x[col[k]]
__device__ __forceinline__ double row_dot(const int *__restrict__ ptr, const int *__restrict__ col,
                                          const int *__restrict__ src,
                                          const double *__restrict__ vals,
                                          const double *__restrict__ x, int row) {
  double s = 0.0;
  for (int k = ptr[row]; k < ptr[row + 1]; k++) s += vals[src[k]] * x[col[k]];
  return s;
}

// FULL: dx, dy, dz = x3 .* scale from the permuted solution (hqp/Hqp_IpSpBKP.C:208-212)
__global__ void k_unpack_full(int n, int me, int m, const int *__restrict__ q2e,
                              const double *__restrict__ sc, const double *__restrict__ xsol,
                              double *__restrict__ dx, double *__restrict__ dy,
                              double *__restrict__ dz) {
  int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n + me + m) return;
  double v = xsol[q2e[q]];
  if (q < n)
    dx[q] = v;
  else if (q < n + me)
    dy[q - n] = v;
  else
    dz[q - n - me] = v * sc[q];
}

// dw = C dx - r3   (hqp/Hqp_IpSpBKP.C:216-217, hqp/Hqp_IpRedSpBKP.C:364-365).  That row carries the
// absolute error of dx; for an active constraint (w_j < z_j, w_j -> 0) the other row that defines
// dw_j, z_j dw_j + w_j dz_j = r4_j, gives it to a relative accuracy instead (the same number in
// exact arithmetic): the step length tests of the interior-point solvers compare dw_j with w_j.
__global__ void k_dw(int m, const int *__restrict__ Cp, const int *__restrict__ Cc,
                     const int *__restrict__ Cs, const double *__restrict__ vals,
                     const double *__restrict__ dx, const double *__restrict__ r3,
                     const double *__restrict__ z, const double *__restrict__ w,
                     const double *__restrict__ r4, const double *__restrict__ dz,
                     double *__restrict__ dw) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  if (w[j] < z[j])
    dw[j] = (r4[j] - w[j] * dz[j]) / z[j];
  else
    dw[j] = -1.0 * r3[j] + row_dot(Cp, Cc, Cs, vals, dx, j);
}

// REDUCED, part 1: tz = r4./w + (z/w).*r3   (hqp/Hqp_IpRedSpBKP.C:339-341)
__device__ __forceinline__ double red_t_elem(const double *__restrict__ w, const double *__restrict__ zw, const double *__restrict__ r3,
                                             const double *__restrict__ r4, int j) {
  return r4[j] / w[j] + zw[j] * r3[j];
}
__global__ void k_red_t(int m, const double *__restrict__ w, const double *__restrict__ zw,
                        const double *__restrict__ r3, const double *__restrict__ r4,
                        double *__restrict__ tz) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  tz[j] = red_t_elem(w, zw, r3, r4, j);
}
// REDUCED, part 2: rhs = P [(r1 - C' tz) .* scale; r2]   (:342-348)
__global__ void k_rhs_red(int n, int me, const int *__restrict__ q2e, const double *__restrict__ sc,
                          const int *__restrict__ CTp, const int *__restrict__ CTc,
                          const int *__restrict__ CTs, const double *__restrict__ vals,
                          const double *__restrict__ tz, const double *__restrict__ r1,
                          const double *__restrict__ r2, double *__restrict__ rhs, int *__restrict__ epoch) {
  int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q == 0 && epoch) *epoch += 1;  // the solve counter of the whole-tree sweeps (solve_top.hip.h)
  if (q >= n + me) return;
  double v;
  if (q < n)
    v = (r1[q] - row_dot(CTp, CTc, CTs, vals, tz, q)) * sc[q];
  else
    v = r2[q - n];
  rhs[q2e[q]] = v;
}
// REDUCED, part 3: dx = scale .* x1, dy   (:354-355)
__global__ void k_unpack_red(int n, int me, const int *__restrict__ q2e,
                             const double *__restrict__ sc, const double *__restrict__ xsol,
                             double *__restrict__ dx, double *__restrict__ dy) {
  int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n + me) return;
  double v = xsol[q2e[q]];
  if (q < n)
    dx[q] = v * sc[q];
  else
    dy[q - n] = v;
}
// REDUCED, part 4: dz = tz - (z/w).*(C dx); dw = C dx - r3   (:357-365)
__global__ void k_red_dzdw(int m, const int *__restrict__ Cp, const int *__restrict__ Cc,
                           const int *__restrict__ Cs, const double *__restrict__ vals,
                           const double *__restrict__ dx, const double *__restrict__ zw,
                           const double *__restrict__ tz, const double *__restrict__ r3,
                           double *__restrict__ dz, double *__restrict__ dw) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  double cdx = row_dot(Cp, Cc, Cs, vals, dx, j);
  dz[j] = tz[j] - zw[j] * cdx;
  dw[j] = -1.0 * r3[j] + cdx;
}

// REDUCED, parts 1 + 2 and parts 3 + 4 as ONE launch each (round 6: inside a replayed graph a launch takes 4.6 us whatever
// it does, and these do a microsecond's work on the systems whose iterations are counted in launches).  The first
// nb_first workgroups do the part that needs the other's result and evaluate it where they need it - tz_j where C' meets
// it, dx_i = scale_i x_i where C meets it: the same expressions, so the same bits -, the rest store it.
__global__ void __launch_bounds__(256)
k_rhs_red_t(int n, int me, int m, int nb_first, const int *__restrict__ q2e, const double *__restrict__ sc, const int *__restrict__ CTp,
            const int *__restrict__ CTc, const int *__restrict__ CTs, const double *__restrict__ vals, const double *__restrict__ w,
            const double *__restrict__ zw, const double *__restrict__ r3, const double *__restrict__ r4, double *__restrict__ tz,
            const double *__restrict__ r1, const double *__restrict__ r2, double *__restrict__ rhs, int *__restrict__ epoch) {
  if ((int)blockIdx.x >= nb_first) {
    const int j = ((int)blockIdx.x - nb_first) * blockDim.x + threadIdx.x;
    if (j < m) tz[j] = red_t_elem(w, zw, r3, r4, j);
    return;
  }
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q == 0 && epoch) *epoch += 1;  // the solve counter of the whole-tree sweeps (solve_top.hip.h)
  if (q >= n + me) return;
  double v;
  if (q < n) {
    double s = 0.0;
    for (int k = CTp[q]; k < CTp[q + 1]; k++) s += vals[CTs[k]] * red_t_elem(w, zw, r3, r4, CTc[k]);
    v = (r1[q] - s) * sc[q];
  } else
    v = r2[q - n];
  rhs[q2e[q]] = v;
}
__global__ void __launch_bounds__(256)
k_unpack_dzdw(int n, int me, int m, int nb_first, const int *__restrict__ q2e, const double *__restrict__ sc, const double *__restrict__ xsol,
              double *__restrict__ dx, double *__restrict__ dy, const int *__restrict__ Cp, const int *__restrict__ Cc,
              const int *__restrict__ Cs, const double *__restrict__ vals, const double *__restrict__ zw, const double *__restrict__ tz,
              const double *__restrict__ r3, double *__restrict__ dz, double *__restrict__ dw) {
  if ((int)blockIdx.x >= nb_first) {
    const int q = ((int)blockIdx.x - nb_first) * blockDim.x + threadIdx.x;
    if (q >= n + me) return;
    const double v = xsol[q2e[q]];
    if (q < n)
      dx[q] = v * sc[q];
    else
      dy[q - n] = v;
    return;
  }
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  double cdx = 0.0;
  for (int k = Cp[j]; k < Cp[j + 1]; k++) {
    const int c = Cc[k];
    const double dxc = xsol[q2e[c]] * sc[c];
    cdx += vals[Cs[k]] * dxc;
  }
  dz[j] = tz[j] - zw[j] * cdx;
  dw[j] = -1.0 * r3[j] + cdx;
}

// ---------------------------------------------------------------- residuum
// Hqp_IpMatrix::residuum (hqp/Hqp_IpMatrix.C:147-176) over the concatenated
// [n | me | m] row space:
//   rho1 = r1 + Q dx - A' dy - C' dz,  rho2 = r2 - A dx,
//   rho3 = r3 - (C dx - dw),           rho4 = r4 - (z.*dw + w.*dz)
// Sixteen lanes (one DPP row) share a matrix row: they stride its non-zeros with
// coalesced index / value loads and add up with DPP row rotations.
struct CsrDev {
  const int *ptr, *col, *src;
  const double *val;  // the values in CSR order (gathered through src once per update())
};
__global__ void k_gather_values(int nnz, const int *__restrict__ src, const double *__restrict__ vals,
                                double *__restrict__ out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < nnz) out[k] = vals[src[k]];
}
// zd_policy -1 (hqpkkt_opts): does some x have a Hessian diagonal that is weak against its coupling to an
// equality, |Q_ii| < 0.01 max_r |A_ri| ?  Evaluated on the device after every hqpkkt_set_values (the values
// of an SQP run change: a Hessian that starts as the identity may become weak later).
__global__ void k_zd_weak(int n, CsrDev Q, CsrDev AT, int *__restrict__ flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double qd = 0.0, am = 0.0;
  for (int k = Q.ptr[i]; k < Q.ptr[i + 1]; k++)
    if (Q.col[k] == i) qd = fabs(Q.val[k]);
  for (int k = AT.ptr[i]; k < AT.ptr[i + 1]; k++) am = fmax(am, fabs(AT.val[k]));
  if (am > 0.0 && qd < 0.01 * am) atomicOr(flag, 1);
}
// sum over the LPR (16 or 4) consecutive lanes that share a CSR row
template <int LPR>
__device__ __forceinline__ double row_sum(double v) {
  v += dpp_move<0xb1, 0xf>(v);   // quad_perm [1,0,3,2]
  v += dpp_move<0x4e, 0xf>(v);   // quad_perm [2,3,0,1]
  if (LPR == 16) {
    v += dpp_move<0x124, 0xf>(v);  // row_ror 4
    v += dpp_move<0x128, 0xf>(v);  // row_ror 8 -> every lane of the row holds the sum
  }
  return v;
}
template <int LPR>
__device__ __forceinline__ double row_dot(const CsrDev M, const double *__restrict__ vals,
                                          const double *__restrict__ x, int row, int sub) {
  // (two entries per lane in flight: the loop is a chain of index -> value round trips, 80 - 160 entries per row on
  // the banded systems)
  double s = 0.0, t = 0.0;
  const int e = M.ptr[row + 1];
  int k = M.ptr[row] + sub;
  for (; k + LPR < e; k += 2 * LPR) {
    const int c0 = M.col[k], c1 = M.col[k + LPR];
    const double v0 = M.val[k], v1 = M.val[k + LPR];
    s += v0 * x[c0], t += v1 * x[c1];
  }
  if (k < e) s += M.val[k] * x[M.col[k]];
  return row_sum<LPR>(s + t);
}
// LPR lanes per CSR row: 16 for the banded systems, 4 when the rows hold a handful of
// entries (DOCP / Prg_DID matrices: 1-3 per row)
template <int LPR>
__global__ void __launch_bounds__(256)
k_residual(int n, int me, int m, CsrDev Q, CsrDev AT, CsrDev CT, CsrDev A, CsrDev C,
           const double *__restrict__ vals, const double *__restrict__ z,
           const double *__restrict__ w, const double *__restrict__ r1,
           const double *__restrict__ r2, const double *__restrict__ r3,
           const double *__restrict__ r4, const double *__restrict__ dx,
           const double *__restrict__ dy, const double *__restrict__ dz,
           const double *__restrict__ dw, double *__restrict__ o1, double *__restrict__ o2,
           double *__restrict__ o3, double *__restrict__ o4,
           unsigned long long *__restrict__ resbits, unsigned long long *__restrict__ resbits_next,
           // (resbits_next: the word the NEXT residual accumulates into, zeroed here - no memset between residuals)
           // STAGED with dense dynamics: x1 = A_dyn' dy (n), x2 = A_dyn dx (first ndyn rows of A,
           // which are empty in the CSR block), computed by the dense kernels of staged.hip.h
           const double *__restrict__ x1 = nullptr, const double *__restrict__ x2 = nullptr, int ndyn = 0) {
  __shared__ double red[4];
  if (blockIdx.x == 0 && threadIdx.x == 0) *resbits_next = 0ULL;
  const int sub = threadIdx.x & (LPR - 1);
  const int total = n + me + m;
  double mag = 0.0;
  constexpr int RPB = 256 / LPR;  // rows per block and trip
  for (int q = blockIdx.x * RPB + threadIdx.x / LPR; q < total; q += gridDim.x * RPB) {
    if (q < n) {
      double s = row_dot<LPR>(Q, vals, dx, q, sub);
      s += -1.0 * row_dot<LPR>(AT, vals, dy, q, sub);
      s += -1.0 * row_dot<LPR>(CT, vals, dz, q, sub);
      if (x1) s -= x1[q];
      s = r1[q] + s;
      if (sub == 0) o1[q] = s;
      mag = fmax(mag, fabs(s) == fabs(s) ? fabs(s) : __longlong_as_double(0x7ff0000000000000LL));
    } else if (q < n + me) {
      const int i = q - n;
      const double s = r2[i] - row_dot<LPR>(A, vals, dx, i, sub) - (i < ndyn ? x2[i] : 0.0);
      if (sub == 0) o2[i] = s;
      mag = fmax(mag, fabs(s) == fabs(s) ? fabs(s) : __longlong_as_double(0x7ff0000000000000LL));
    } else {
      const int j = q - n - me;
      const double cdx = row_dot<LPR>(C, vals, dx, j, sub);
      const double s3 = r3[j] - (cdx - dw[j]);
      const double s4 = r4[j] - (z[j] * dw[j] + w[j] * dz[j]);
      if (sub == 0) o3[j] = s3, o4[j] = s4;
      const double t = fmax(fabs(s3), fabs(s4));
      // a NaN must not get lost in the max
      mag = fmax(mag, (s3 == s3 && s4 == s4) ? t : __longlong_as_double(0x7ff0000000000000LL));
    }
  }
  mag = wave_max(mag);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mag;
  __syncthreads();
  if (threadIdx.x == 0) {
    mag = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    if (mag > 0.0) atomic_max_pos(resbits, mag);
  }
}

// several vectors moved by one launch (staging of caller pointers into the
// handle's fixed buffers, so that the numeric sequences can be replayed as graphs)
// the panel arena and the 128 status words of a factorisation cleared by ONE kernel of the captured sequence (no memset
// nodes in the graphs: a graph whose memset nodes had run a few times was seen to clear with the arguments of a later
// copy of the caller's on another stream - status "6000", tests/test_gpu_parity.py::test_repeated_calls_...)
// Every workgroup clears one contiguous piece with ordinary 16-byte stores: 26.8 us for C2's 0.21 GB arena at any grid
// from 1024 workgroups on, as fast as the runtime's fill (27.2 us); a grid-stride loop 32 - 35 us, streaming (non-temporal)
// stores 40 - 46 us (tools/clear_probe.hip).
__global__ void __launch_bounds__(256) k_clear(double *__restrict__ p, long long n, int *__restrict__ words) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  if (blockIdx.x == 0 && threadIdx.x < 128 && threadIdx.x != TINY_REPLACE_WORD) words[threadIdx.x] = 0;  // (that word is the caller's)
  const long long n2 = n >> 1, per = (n2 + gridDim.x - 1) / gridDim.x;
  const long long b0 = (long long)blockIdx.x * per, b1 = b0 + per < n2 ? b0 + per : n2;
  d2 *q = (d2 *)p;
  const d2 zero{0.0, 0.0};
  long long i = b0 + threadIdx.x;
  for (; i + 768 < b1; i += 1024) q[i] = zero, q[i + 256] = zero, q[i + 512] = zero, q[i + 768] = zero;
  for (; i < b1; i += 256) q[i] = zero;
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) p[n - 1] = 0.0;
}
// Read-back through mapped, coherent host memory (hqpkkt::hpin_dev): ONE wavefront stores the 128 status words of
// `flags` (as 64 eight-byte words) and, with `out`, n_out <= 64 doubles behind double 64, then - behind a system-scope
// fence - the sequence number the host spins on.  (One wavefront: its stores are ordered by its own fence.)
// The number is counted on the device (*dev_seq: this launch is the only writer, launches of a stream follow each
// other), so that a launch replayed inside a captured graph posts the next number like any other; the host counts its
// posts along.  residual != 0 (the post behind a residual kernel): the residual maximum (word 61 of `flags`, k_residual)
// goes along and is cleared once it is on its way - the next residual kernel starts from zero whatever ran in between;
// any other post leaves the host's copy of that word as the last such post has written it.
#define HPIN_DOUBLES 256
#define HPIN_SEQ 200  // the double of hpin whose first four bytes hold the sequence number
#define HPIN_ZM 208   // two doubles the HOST writes for a kernel to read: zeta and mu of a step of the Franke loop (k_fr_rhs)
__global__ void __launch_bounds__(64) k_post_words(int *__restrict__ flags, const double *__restrict__ out, int n_out,
                                                   double *__restrict__ host, unsigned *__restrict__ dev_seq, int residual) {
  const int lane = threadIdx.x;
  const unsigned long long w = ((const unsigned long long *)flags)[lane];
  const bool rword = lane == 59 || lane == 61;
  if (!rword || residual) __hip_atomic_store((unsigned long long *)host + lane, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (rword && residual) ((unsigned long long *)flags)[lane] = 0ULL;
  if (out && lane < n_out)
    __hip_atomic_store((unsigned long long *)host + 64 + lane, (unsigned long long)__double_as_longlong(out[lane]), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  __threadfence_system();
  if (lane == 0) {
    const unsigned seq = *dev_seq + 1u;
    *dev_seq = seq;
    __hip_atomic_store((unsigned *)(host + HPIN_SEQ), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// sharded mode: clear the (offset, length) ranges of an arena this rank writes
__global__ void k_zero_ranges(double *__restrict__ base, const long long *__restrict__ ranges) {
  const long long off = ranges[2 * blockIdx.y], len = ranges[2 * blockIdx.y + 1];
  double *p = base + off;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < len;
       i += (long long)gridDim.x * blockDim.x)
    p[i] = 0.0;
}
// sharded mode: the status words of a factorisation as doubles for one all-reduce (sum): [0] ranks
// with E_SING (or an infinite / NaN max|K|), [1] ranks with another error, [2] ranks that perturbed an
// exactly zero pivot (soft singular), [3] ranks that met a tiny multiplier-type pivot (SOFT_PIVOT_REL, flags[5]:
// seen by the owner of the subtree only, and hqpkkt_solve turns it into E_SING when the refinement fails - on
// every rank or on none); and back: every rank ends with the same words
__global__ void k_status_pack(const int *__restrict__ flags, const unsigned long long *__restrict__ bits,
                              double *__restrict__ out) {
  if (threadIdx.x != 0) return;
  const double kmax = __longlong_as_double((long long)bits[0]);
  const bool badk = !(kmax == kmax) || kmax == __longlong_as_double(0x7ff0000000000000LL);
  out[0] = (flags[0] == 4 || badk) ? 1.0 : 0.0;
  out[1] = (flags[0] != 0 && flags[0] != 4) ? 1.0 : 0.0;
  out[2] = flags[4] != 0 ? 1.0 : 0.0;
  out[3] = flags[5] != 0 ? 1.0 : 0.0;
}
__global__ void k_status_unpack(const double *__restrict__ in, int *__restrict__ flags,
                                unsigned long long *__restrict__ bits) {
  if (threadIdx.x != 0) return;
  if (in[1] > 0.0)
    flags[0] = 17;
  else if (in[0] > 0.0)
    flags[0] = 4;
  if (in[2] > 0.0) flags[4] = 1;
  if (in[3] > 0.0) flags[5] = 1;
  (void)bits;
}
// sharded mode: keep only the entries this rank contributes to the all-reduce
__global__ void k_mask_vector(int n, const signed char *__restrict__ keep, double *__restrict__ x) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && !keep[i]) x[i] = 0.0;
}

struct CopyList {
  const double *src[6];
  double *dst[6];
  int len[6];
};
__global__ void k_copy_vectors(CopyList L, int nvec) {
  for (int v = 0; v < nvec; v++) {
    const double *__restrict__ s = L.src[v];
    double *__restrict__ d = L.dst[v];
    if (!s || !d) continue;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < L.len[v]; i += gridDim.x * blockDim.x) d[i] = s[i];
  }
}

// d <- d + alpha e over the four blocks (v_mltadd, hqp/Hqp_IpMatrix.C:103-106)
__global__ void k_axpy4(int n, int me, int m, double alpha, const double *__restrict__ e1,
                        const double *__restrict__ e2, const double *__restrict__ e3,
                        const double *__restrict__ e4, double *__restrict__ d1,
                        double *__restrict__ d2, double *__restrict__ d3,
                        double *__restrict__ d4) {
  int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n)
    d1[q] = d1[q] + alpha * e1[q];
  else if (q < n + me)
    d2[q - n] = d2[q - n] + alpha * e2[q - n];
  else if (q < n + me + m)
    d3[q - n - me] = d3[q - n - me] + alpha * e3[q - n - me];
  else if (q < n + me + 2 * m)
    d4[q - n - me - m] = d4[q - n - me - m] + alpha * e4[q - n - me - m];
}

}  // namespace kktdev
