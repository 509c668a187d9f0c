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

namespace kktdev {

typedef double double4_t __attribute__((ext_vector_type(4)));

struct DevTree {
  const int *piv_start, *npiv, *nbor, *parent;
  const long long *bptr;
  const int *bidx, *rel;
  const long long *panel_off, *upd_off, *x_off, *cb_off;
  const int *child_ptr, *child_idx;
};

// order-preserving max for non-negative doubles through their bit pattern
__device__ __forceinline__ void atomic_max_pos(unsigned long long *addr, double v) {
  atomicMax(addr, (unsigned long long)__double_as_longlong(v));
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
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

// unscaled value of every stored entry of the matrix to factor
__global__ void k_entry_values(int nent, const int *__restrict__ term_ptr,
                               const TermDev *__restrict__ terms,
                               const double *__restrict__ vals, const double *__restrict__ wt,
                               double *__restrict__ ent_val) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nent) return;
  double v = 0.0;
  for (int t = term_ptr[e]; t < term_ptr[e + 1]; t++) {
    TermDev tm = terms[t];
    v += tm.sgn * vals[tm.s1] * vals[tm.s2] * wt[tm.wi];
  }
  ent_val[e] = v;
}

// REDUCED: scale_i = min(1, sqrt(-1/J_ii))  (hqp/Hqp_IpRedSpBKP.C:130,138)
__global__ void k_red_scale(int n, const int *__restrict__ diag_ent,
                            const double *__restrict__ ent_val, double *__restrict__ sc) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int e = diag_ent[i];
  sc[i] = e >= 0 ? fmin(1.0, sqrt(-1.0 / ent_val[e])) : 1.0;
}

// J <- S J S scattered into the supernode panels (hqp/Hqp_IpSpBKP.C:162-176)
__global__ void k_scatter(int nent, const int *__restrict__ ent_a, const int *__restrict__ ent_b,
                          const long long *__restrict__ ent_dst,
                          const double *__restrict__ ent_val, const double *__restrict__ sc,
                          double *__restrict__ panel, unsigned long long *__restrict__ kmax) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  double v = 0.0;
  if (e < nent) {
    v = ent_val[e] * sc[ent_a[e]] * sc[ent_b[e]];
    panel[ent_dst[e]] = v;
  }
  double mx = wave_max(fabs(v));
  if ((threadIdx.x & 63) == 0 && mx > 0.0) atomic_max_pos(kmax, mx);
}

// ------------------------------------------------------------- extend-add
// child update matrix -> parent front (panel columns or parent update block)
__global__ void k_extend_add(DevTree T, const int *__restrict__ seg_nodes,
                             double *__restrict__ panel, double *__restrict__ upd) {
  const int c = seg_nodes[blockIdx.x];
  const int q = T.parent[c];
  const int b = T.nbor[c];
  const int pq = T.npiv[q], bq = T.nbor[q];
  const long long Fq = pq + bq;
  const int *rel = T.rel + T.bptr[c];
  const double *U = upd + T.upd_off[c];
  double *Pq = panel + T.panel_off[q];
  double *Uq = upd + T.upd_off[q];
  for (int j = blockIdx.y; j < b; j += gridDim.y) {
    const int rj = rel[j];
    for (int i = j + threadIdx.x; i < b; i += blockDim.x) {
      const int ri = rel[i];
      const double v = U[(long long)j * b + i];
      if (rj < pq)
        Pq[(long long)rj * Fq + ri] += v;
      else
        Uq[(long long)(rj - pq) * bq + (ri - pq)] += v;
    }
  }
}

// ------------------------------------------------- pivot block: dense BK LDL'
// One workgroup per supernode.  The p x p pivot block sits in LDS (lower
// triangle, leading dimension p|1).  Bunch-Kaufman partial pivoting with the
// reference's threshold alpha = tol (1+sqrt 17)/8 and test order
// (hqp/spBKP.C:392, 431-438, 471, 480), restricted to the pivot block; a pivot
// below pert = pivot_eps * max|K| is replaced by +-pert (sign from the block the
// row belongs to: x rows negative, y/slack rows positive).
struct DiagShared {
  int kind, r;
  double d0, d1, d2;
};

__global__ void __launch_bounds__(256)
k_factor_diag(DevTree T, const int *__restrict__ level_nodes, double *__restrict__ panel,
              double *__restrict__ dinv, int *__restrict__ ptype, int *__restrict__ lperm,
              const signed char *__restrict__ esign, double alpha, double pivot_eps,
              const unsigned long long *__restrict__ kmax_bits, int *__restrict__ counters) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int node = level_nodes[blockIdx.x];
  const int p = T.npiv[node], b = T.nbor[node];
  const long long F = p + b;
  const int e0 = T.piv_start[node];
  double *P = panel + T.panel_off[node];
  const int ld = p | 1;
  double *a = lds;                 // ld * p
  double *v1 = a + ld * p;         // p   scaled column(s)
  double *v2 = v1 + p;
  double *w1 = v2 + p;             // p   unscaled column(s)
  double *w2 = w1 + p;
  int *lp = (int *)(w2 + p);       // p   local pivot order
  int *pt = lp + p;                // p   pivot type per position
  __shared__ DiagShared sh;
  const int tid = threadIdx.x, lane = tid & 63;
  const double pert = pivot_eps * __longlong_as_double((long long)*kmax_bits);

  for (int idx = tid; idx < p * p; idx += blockDim.x) {
    int i = idx % p, j = idx / p;
    a[i + j * ld] = (i >= j) ? P[(long long)j * F + i] : 0.0;
  }
  for (int i = tid; i < p; i += blockDim.x) lp[i] = i;
  __syncthreads();

  int k = 0;
  while (k < p) {
    // ---- decision (wave 0) ------------------------------------------------
    if (tid < 64) {
      double best = -1.0;
      int bi = p;
      for (int i = k + 1 + lane; i < p; i += 64) {
        double t = fabs(a[i + k * ld]);
        if (t > best) best = t, bi = i;  // ascending i inside a lane: first max wins
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        double ob = __shfl_xor(best, o);
        int oi = __shfl_xor(bi, o);
        if (ob > best || (ob == best && oi < bi)) best = ob, bi = oi;
      }
      const double akk = fabs(a[k + k * ld]);
      const double lambda = best < 0.0 ? 0.0 : best;
      int kind = 0, r = bi;
      if (!(akk >= alpha * lambda)) {
        double s = 0.0;
        for (int t = k + lane; t < p; t += 64)
          if (t != r) s = fmax(s, fabs(t < r ? a[r + t * ld] : a[t + r * ld]));
        const double sigma = wave_max(s);
        if (sigma * akk >= alpha * lambda * lambda)
          kind = 0;
        else if (fabs(a[r + r * ld]) >= alpha * sigma)
          kind = 1;
        else
          kind = 2;
      }
      if (lane == 0) sh.kind = kind, sh.r = r;
    }
    __syncthreads();
    const int kind = sh.kind, r = sh.r;
    // ---- symmetric interchange -------------------------------------------
    const int p1 = (kind == 2) ? k + 1 : k;
    if (kind != 0 && r != p1) {
      for (int t = tid; t < p; t += blockDim.x) {
        double *x, *y;
        if (t < p1)
          x = &a[p1 + t * ld], y = &a[r + t * ld];
        else if (t == p1)
          x = &a[p1 + p1 * ld], y = &a[r + r * ld];
        else if (t < r)
          x = &a[t + p1 * ld], y = &a[r + t * ld];
        else if (t == r)
          continue;
        else
          x = &a[t + p1 * ld], y = &a[t + r * ld];
        double tmp = *x;
        *x = *y;
        *y = tmp;
      }
      if (tid == 0) {
        int t = lp[p1];
        lp[p1] = lp[r];
        lp[r] = t;
      }
      __syncthreads();
    }
    // ---- pivot inverse -----------------------------------------------------
    if (tid == 0) {
      if (kind != 2) {
        double d = a[k + k * ld];
        if (!(fabs(d) >= pert) || d == 0.0) {
          d = (double)esign[e0 + lp[k]] * fmax(pert, 1e-300);
          atomicAdd(&counters[1], 1);
        }
        sh.d0 = 1.0 / d;
        a[k + k * ld] = d;
        ptype[e0 + k] = 0, pt[k] = 0;
        dinv[2 * (e0 + k)] = 1.0 / d;
        dinv[2 * (e0 + k) + 1] = 0.0;
      } else {
        double d11 = a[k + k * ld], d21 = a[k + 1 + k * ld], d22 = a[k + 1 + (k + 1) * ld];
        double det = d11 * d22 - d21 * d21;
        if (!(fabs(det) >= pert * pert) || det == 0.0) {
          // degenerate 2x2: make it a perturbed diagonal pair
          d11 = (double)esign[e0 + lp[k]] * fmax(pert, 1e-300);
          d22 = (double)esign[e0 + lp[k + 1]] * fmax(pert, 1e-300);
          d21 = 0.0;
          det = d11 * d22;
          a[k + k * ld] = d11, a[k + 1 + k * ld] = 0.0, a[k + 1 + (k + 1) * ld] = d22;
          atomicAdd(&counters[1], 2);
        }
        sh.d0 = d22 / det, sh.d1 = -d21 / det, sh.d2 = d11 / det;
        ptype[e0 + k] = 1, ptype[e0 + k + 1] = 2, pt[k] = 1, pt[k + 1] = 2;
        dinv[2 * (e0 + k)] = sh.d0, dinv[2 * (e0 + k) + 1] = sh.d1;
        dinv[2 * (e0 + k + 1)] = sh.d2, dinv[2 * (e0 + k + 1) + 1] = sh.d1;
        atomicAdd(&counters[0], 1);
      }
    }
    __syncthreads();
    // ---- multipliers ---------------------------------------------------------
    const int kw = (kind == 2) ? 2 : 1;
    if (kw == 1) {
      const double di = sh.d0;
      for (int i = k + 1 + tid; i < p; i += blockDim.x) {
        double c = a[i + k * ld];
        w1[i] = c;
        c *= di;
        v1[i] = c;
        a[i + k * ld] = c;
      }
    } else {
      const double i11 = sh.d0, i21 = sh.d1, i22 = sh.d2;
      for (int i = k + 2 + tid; i < p; i += blockDim.x) {
        double c1 = a[i + k * ld], c2 = a[i + (k + 1) * ld];
        w1[i] = c1, w2[i] = c2;
        double l1 = c1 * i11 + c2 * i21, l2 = c1 * i21 + c2 * i22;
        v1[i] = l1, v2[i] = l2;
        a[i + k * ld] = l1, a[i + (k + 1) * ld] = l2;
      }
    }
    __syncthreads();
    // ---- trailing update (lower triangle) ------------------------------------
    {
      const int s = k + kw, nt = p - s;
      // thread owns column strips: element (i, j), i >= j
      for (int idx = tid; idx < nt * nt; idx += blockDim.x) {
        int i = s + idx % nt, j = s + idx / nt;
        if (i < j) continue;
        double u = v1[i] * w1[j];
        if (kw == 2) u += v2[i] * w2[j];
        a[i + j * ld] -= u;
      }
    }
    __syncthreads();
    k += kw;
  }
  // ---- write back: L11 (strict lower, zero under 2x2 diagonals), diag = D ----
  for (int idx = tid; idx < p * p; idx += blockDim.x) {
    int i = idx % p, j = idx / p;
    if (i < j) continue;
    double v = a[i + j * ld];
    if (i == j + 1 && pt[j] == 1) v = 0.0;
    P[(long long)j * F + i] = v;
  }
  for (int i = tid; i < p; i += blockDim.x) lperm[e0 + i] = lp[i];
}

// --------------------------------------------------- panel solve (border rows)
// X = A21 P' L11^-T,  L21 = X D^-1.  One workgroup per (supernode, 32-row slab).
// Column blocks of 16: in-block forward substitution, then a rank-16 update of
// the remaining columns.
#define PS_COLS 16
__global__ void __launch_bounds__(256)
k_panel_solve(DevTree T, const int *__restrict__ slabs, double *__restrict__ panel,
              double *__restrict__ xar, const double *__restrict__ dinv,
              const int *__restrict__ ptype, const int *__restrict__ lperm) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int node = slabs[2 * blockIdx.x], slab = slabs[2 * blockIdx.x + 1];
  const int p = T.npiv[node], b = T.nbor[node];
  const long long F = p + b;
  const int e0 = T.piv_start[node];
  double *P = panel + T.panel_off[node];
  double *X = xar + T.x_off[node];
  const int r0 = slab * 32;
  const int tid = threadIdx.x, r = tid & 31, g = tid >> 5;
  const bool live = (r0 + r) < b;
  double *s = lds;                  // 32 x p, s[r + 32*k]
  double *Lb = s + 32 * p;          // p x PS_COLS block of L11: Lb[j + p*kk]
  // load the slab with the pivot-block column permutation
  for (int kcol = g; kcol < p; kcol += 8)
    s[r + 32 * kcol] = live ? P[(long long)lperm[e0 + kcol] * F + p + r0 + r] : 0.0;
  for (int kb = 0; kb < p; kb += PS_COLS) {
    const int kw = min(PS_COLS, p - kb);
    __syncthreads();
    for (int idx = tid; idx < (p - kb) * kw; idx += blockDim.x) {
      int j = kb + idx % (p - kb), kk = idx / (p - kb);
      Lb[j + p * kk] = P[(long long)(kb + kk) * F + j];
    }
    __syncthreads();
    // in-block sequential part
    for (int kk = 0; kk < kw; kk++) {
      const double xk = s[r + 32 * (kb + kk)];
      for (int j = kb + kk + 1 + g; j < kb + kw; j += 8) s[r + 32 * j] -= xk * Lb[j + p * kk];
      __syncthreads();
    }
    // rank-kw update of the columns to the right of the block
    for (int j = kb + kw + g; j < p; j += 8) {
      double acc = 0.0;
      for (int kk = 0; kk < kw; kk++) acc += s[r + 32 * (kb + kk)] * Lb[j + p * kk];
      s[r + 32 * j] -= acc;
    }
  }
  __syncthreads();
  if (!live) return;
  for (int kcol = g; kcol < p; kcol += 8) {
    const int e = e0 + kcol, ty = ptype[e];
    const double x = s[r + 32 * kcol];
    double l;
    if (ty == 0)
      l = x * dinv[2 * e];
    else if (ty == 1)
      l = x * dinv[2 * e] + s[r + 32 * (kcol + 1)] * dinv[2 * e + 1];
    else
      l = s[r + 32 * (kcol - 1)] * dinv[2 * e + 1] + x * dinv[2 * e];
    X[(long long)kcol * b + r0 + r] = x;
    P[(long long)kcol * F + p + r0 + r] = l;
  }
}

// ----------------------------------------------------- Schur update (MFMA f64)
// U(i,j) -= sum_k L21(i,k) X(j,k) on 64x64 tiles of the lower triangle, four
// waves per workgroup, each wave a 32x32 block = 2x2 v_mfma_f64_16x16x4_f64.
// A operand: lane l holds A[l&15][l>>4]; B operand: B[l>>4][l&15];
// C/D: 4 values per lane, col = l&15, row = (l>>4) + 4*reg  (f64 layout).
__device__ __forceinline__ double4_t mfma_f64(double a, double b, double4_t c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

__global__ void __launch_bounds__(256)
k_schur_update(DevTree T, const int *__restrict__ tiles, const double *__restrict__ panel,
               const double *__restrict__ xar, double *__restrict__ upd) {
  const int node = tiles[3 * blockIdx.x], ti = tiles[3 * blockIdx.x + 1],
            tj = tiles[3 * blockIdx.x + 2];
  const int p = T.npiv[node], b = T.nbor[node];
  const long long F = p + b;
  const double *L = panel + T.panel_off[node] + p;  // L21(i,k) = L[k*F + i]
  const double *X = xar + T.x_off[node];            // X(j,k)   = X[k*b + j]
  double *U = upd + T.upd_off[node];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i0 = ti * 64 + (wave >> 1) * 32, j0 = tj * 64 + (wave & 1) * 32;
  if (i0 >= b || j0 >= b || i0 + 31 < j0) return;  // wave-uniform
  const int lr = lane & 15, lk = lane >> 4;
  double4_t acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; x++)
#pragma unroll
    for (int y = 0; y < 2; y++) acc[x][y] = (double4_t){0.0, 0.0, 0.0, 0.0};
  const int ia = i0 + lr, ib = i0 + 16 + lr, ja = j0 + lr, jb = j0 + 16 + lr;
  for (int k0 = 0; k0 < p; k0 += 4) {
    const int k = k0 + lk;
    const bool kin = k < p;
    const double a0 = (kin && ia < b) ? L[(long long)k * F + ia] : 0.0;
    const double a1 = (kin && ib < b) ? L[(long long)k * F + ib] : 0.0;
    const double b0 = (kin && ja < b) ? X[(long long)k * b + ja] : 0.0;
    const double b1 = (kin && jb < b) ? X[(long long)k * b + jb] : 0.0;
    acc[0][0] = mfma_f64(a0, b0, acc[0][0]);
    acc[0][1] = mfma_f64(a0, b1, acc[0][1]);
    acc[1][0] = mfma_f64(a1, b0, acc[1][0]);
    acc[1][1] = mfma_f64(a1, b1, acc[1][1]);
  }
#pragma unroll
  for (int x = 0; x < 2; x++)
#pragma unroll
    for (int y = 0; y < 2; y++)
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        const int i = i0 + 16 * x + lk + 4 * rg, j = j0 + 16 * y + lr;
        if (i < b && j < b && i >= j) U[(long long)j * b + i] -= acc[x][y][rg];
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
// Forward sweep of one level: t = [rhs(pivots); 0] + children contributions,
// y = L11^-1 P t1, contribution = t2 - L21 y, xsol(pivots) = D^-1 y.
#define SV_COLS 16
__global__ void __launch_bounds__(256)
k_solve_fwd(DevTree T, const int *__restrict__ level_nodes, const double *__restrict__ panel,
            const double *__restrict__ dinv, const int *__restrict__ ptype,
            const int *__restrict__ lperm, const double *__restrict__ rhs,
            double *__restrict__ xsol, double *__restrict__ cb) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int node = level_nodes[blockIdx.x];
  const int p = T.npiv[node], b = T.nbor[node];
  const long long F = p + b;
  const int e0 = T.piv_start[node];
  const double *P = panel + T.panel_off[node];
  const int tid = threadIdx.x;
  double *t = lds;            // F
  double *y = t + (p + b);    // p
  double *Lb = y + p;         // p x SV_COLS
  for (int i = tid; i < p + b; i += blockDim.x) t[i] = i < p ? rhs[e0 + i] : 0.0;
  __syncthreads();
  for (int cc = T.child_ptr[node]; cc < T.child_ptr[node + 1]; cc++) {
    const int c = T.child_idx[cc];
    const int bc = T.nbor[c];
    const int *rel = T.rel + T.bptr[c];
    const double *cbc = cb + T.cb_off[c];
    for (int i = tid; i < bc; i += blockDim.x) t[rel[i]] += cbc[i];
    __syncthreads();
  }
  for (int k = tid; k < p; k += blockDim.x) y[k] = t[lperm[e0 + k]];
  for (int kb = 0; kb < p; kb += SV_COLS) {
    const int kw = min(SV_COLS, p - kb);
    __syncthreads();
    for (int idx = tid; idx < (p - kb) * kw; idx += blockDim.x) {
      int j = kb + idx % (p - kb), kk = idx / (p - kb);
      Lb[j + p * kk] = P[(long long)(kb + kk) * F + j];
    }
    __syncthreads();
    if (tid < 64) {  // sequential part inside the block (kw <= 16 rows)
      for (int kk = 0; kk < kw; kk++) {
        const double yk = y[kb + kk];
        const int j = kb + kk + 1 + tid;
        if (j < kb + kw) y[j] -= Lb[j + p * kk] * yk;
        __builtin_amdgcn_wave_barrier();
      }
    }
    __syncthreads();
    for (int j = kb + kw + tid; j < p; j += blockDim.x) {
      double acc = 0.0;
      for (int kk = 0; kk < kw; kk++) acc += Lb[j + p * kk] * y[kb + kk];
      y[j] -= acc;
    }
  }
  __syncthreads();
  double *cbn = cb + T.cb_off[node];
  for (int i = tid; i < b; i += blockDim.x) {
    double acc = t[p + i];
    const double *Li = P + p + i;
    for (int k = 0; k < p; k++) acc -= Li[(long long)k * F] * y[k];
    cbn[i] = acc;
  }
  for (int k = tid; k < p; k += blockDim.x) {
    const int e = e0 + k, ty = ptype[e];
    double v;
    if (ty == 0)
      v = y[k] * dinv[2 * e];
    else if (ty == 1)
      v = y[k] * dinv[2 * e] + y[k + 1] * dinv[2 * e + 1];
    else
      v = y[k - 1] * dinv[2 * e + 1] + y[k] * dinv[2 * e];
    xsol[e] = v;
  }
}

// Backward sweep of one level: x1 = P' L11^-T (yd - L21' x2)
__global__ void __launch_bounds__(256)
k_solve_bwd(DevTree T, const int *__restrict__ level_nodes, const double *__restrict__ panel,
            const int *__restrict__ lperm, double *__restrict__ xsol) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int node = level_nodes[blockIdx.x];
  const int p = T.npiv[node], b = T.nbor[node];
  const long long F = p + b;
  const int e0 = T.piv_start[node];
  const double *P = panel + T.panel_off[node];
  const int *bi = T.bidx + T.bptr[node];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  double *x2 = lds;          // b
  double *v = x2 + b;        // p
  double *Lb = v + p;        // p x SV_COLS
  for (int i = tid; i < b; i += blockDim.x) x2[i] = xsol[bi[i]];
  for (int k = tid; k < p; k += blockDim.x) v[k] = xsol[e0 + k];
  __syncthreads();
  // v -= L21' x2 : one wave per column, lanes stride the rows
  for (int k = wave; k < p; k += 4) {
    const double *Lk = P + (long long)k * F + p;
    double acc = 0.0;
    for (int i = lane; i < b; i += 64) acc += Lk[i] * x2[i];
    acc = wave_sum(acc);
    if (lane == 0) v[k] -= acc;
  }
  // x = L11^-T v, column blocks from the right
  const int nblk = (p + SV_COLS - 1) / SV_COLS;
  for (int blk = nblk - 1; blk >= 0; blk--) {
    const int kb = blk * SV_COLS, kw = min(SV_COLS, p - kb);
    __syncthreads();
    for (int idx = tid; idx < (p - kb) * kw; idx += blockDim.x) {
      int j = kb + idx % (p - kb), kk = idx / (p - kb);
      Lb[j + p * kk] = P[(long long)(kb + kk) * F + j];
    }
    __syncthreads();
    // contributions of the already solved rows j >= kb+kw
    for (int kk = wave; kk < kw; kk += 4) {
      double acc = 0.0;
      for (int j = kb + kw + lane; j < p; j += 64) acc += Lb[j + p * kk] * v[j];
      acc = wave_sum(acc);
      if (lane == 0) v[kb + kk] -= acc;
    }
    __syncthreads();
    if (tid == 0) {
      for (int kk = kw - 1; kk >= 0; kk--) {
        double acc = v[kb + kk];
        for (int j = kb + kk + 1; j < kb + kw; j++) acc -= Lb[j + p * kk] * v[j];
        v[kb + kk] = acc;
      }
    }
  }
  __syncthreads();
  for (int k = tid; k < p; k += blockDim.x) xsol[e0 + lperm[e0 + k]] = v[k];
}

// ------------------------------------------------------------ step pre/post
// FULL: rhs = P [r1; r2; (r3 + r4./z) .* scale]   (hqp/Hqp_IpSpBKP.C:196-204)
__global__ void k_rhs_full(int n, int me, int m, const int *__restrict__ q2e,
                           const double *__restrict__ sc, const double *__restrict__ z,
                           const double *__restrict__ r1, const double *__restrict__ r2,
                           const double *__restrict__ r3, const double *__restrict__ r4,
                           double *__restrict__ rhs) {
  int q = blockIdx.x * blockDim.x + threadIdx.x;
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

// dw = C dx - r3   (hqp/Hqp_IpSpBKP.C:216-217, hqp/Hqp_IpRedSpBKP.C:364-365)
__global__ void k_dw(int m, const int *__restrict__ Cp, const int *__restrict__ Cc,
                     const int *__restrict__ Cs, const double *__restrict__ vals,
                     const double *__restrict__ dx, const double *__restrict__ r3,
                     double *__restrict__ dw) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  dw[j] = -1.0 * r3[j] + row_dot(Cp, Cc, Cs, vals, dx, j);
}

// REDUCED, part 1: tz = r4./w + (z/w).*r3   (hqp/Hqp_IpRedSpBKP.C:339-341)
__global__ void k_red_t(int m, const double *__restrict__ w, const double *__restrict__ zw,
                        const double *__restrict__ r3, const double *__restrict__ r4,
                        double *__restrict__ tz) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  tz[j] = r4[j] / w[j] + zw[j] * r3[j];
}
// REDUCED, part 2: rhs = P [(r1 - C' tz) .* scale; r2]   (:342-348)
__global__ void k_rhs_red(int n, int me, const int *__restrict__ q2e, const double *__restrict__ sc,
                          const int *__restrict__ CTp, const int *__restrict__ CTc,
                          const int *__restrict__ CTs, const double *__restrict__ vals,
                          const double *__restrict__ tz, const double *__restrict__ r1,
                          const double *__restrict__ r2, double *__restrict__ rhs) {
  int q = blockIdx.x * blockDim.x + threadIdx.x;
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

// ---------------------------------------------------------------- residuum
// Hqp_IpMatrix::residuum (hqp/Hqp_IpMatrix.C:147-176), one thread per row of
// the concatenated [n | me | m] index space:
//   rho1 = r1 + Q dx - A' dy - C' dz,  rho2 = r2 - A dx,
//   rho3 = r3 - (C dx - dw),           rho4 = r4 - (z.*dw + w.*dz)
struct CsrDev {
  const int *ptr, *col, *src;
};
__global__ void k_residual(int n, int me, int m, CsrDev Q, CsrDev AT, CsrDev CT, CsrDev A, CsrDev C,
                           const double *__restrict__ vals, const double *__restrict__ z,
                           const double *__restrict__ w, const double *__restrict__ r1,
                           const double *__restrict__ r2, const double *__restrict__ r3,
                           const double *__restrict__ r4, const double *__restrict__ dx,
                           const double *__restrict__ dy, const double *__restrict__ dz,
                           const double *__restrict__ dw, double *__restrict__ o1,
                           double *__restrict__ o2, double *__restrict__ o3,
                           double *__restrict__ o4, unsigned long long *__restrict__ resbits) {
  int q = blockIdx.x * blockDim.x + threadIdx.x;
  double mag = 0.0;
  if (q < n) {
    double s = row_dot(Q.ptr, Q.col, Q.src, vals, dx, q);
    s += -1.0 * row_dot(AT.ptr, AT.col, AT.src, vals, dy, q);
    s += -1.0 * row_dot(CT.ptr, CT.col, CT.src, vals, dz, q);
    s = r1[q] + s;
    o1[q] = s;
    mag = fabs(s);
  } else if (q < n + me) {
    int i = q - n;
    double s = r2[i] - row_dot(A.ptr, A.col, A.src, vals, dx, i);
    o2[i] = s;
    mag = fabs(s);
  } else if (q < n + me + m) {
    int j = q - n - me;
    double cdx = row_dot(C.ptr, C.col, C.src, vals, dx, j);
    double s3 = r3[j] - (cdx - dw[j]);
    double s4 = r4[j] - (z[j] * dw[j] + w[j] * dz[j]);
    o3[j] = s3, o4[j] = s4;
    mag = fmax(fabs(s3), fabs(s4));
  }
  // NaN must not be lost by the max
  if (mag != mag) mag = __longlong_as_double(0x7ff0000000000000LL);
  mag = wave_max(mag);
  if ((threadIdx.x & 63) == 0 && mag > 0.0) atomic_max_pos(resbits, mag);
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
