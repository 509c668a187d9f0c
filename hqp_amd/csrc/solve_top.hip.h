// ---- the top of the assembly tree in ONE launch: forward sweep up to the root and backward sweep down again
//
// Above the level where a tree level is a handful of fronts, a level of the solve is a latency chain (gather the
// children's contributions, y = M P t, c - L21 y; then v = yd - L21' x, x = P' M' v) and its cost is the launches and
// the first-touch memory latency of M and L21: 22 + 19 us per level and direction on the C2 tree against 1 - 2 us
// of arithmetic.  k_solve_top gives every front of the top levels (at most ST_MAXFRONTS, all resident at once) ONE
// workgroup of 16 wavefronts for the whole solve:
//   - before anything else it loads what depends on the front alone: the lower triangle of M = L11^-1 into LDS
//     (packed by columns), L21 into registers (lane = border row within a 64-row slab, wavefront w holds the columns
//     k = w (mod 16): 3 slabs x 11 columns per lane), permutation, pivot data, index lists;
//   - forward step when its children have arrived (per-front flag, agent-scope release / acquire; children below
//     the fused levels were finished by the launches before), backward step when its parent has;
//   - M and L21 stay on chip between the two steps: the top of the tree reads them once per solve.
// Flags reset themselves: a forward flag has one consumer (the parent, which clears it), a backward flag is set to
// the number of fused children and every child takes one.  A wait gives up after ~2^20 polls (a second or so) and
// raises flags[ST_GAVE_UP]: the grid fits the chip many times over, so this only happens when something else is wrong
// (hqpkkt_solve then clears the protocol flags and reports HQPKKT_E_DEVICE instead of hanging the device).
// Arithmetic: plain FMA sums in a fixed order (thread-local over k, then the 16 / 4 partial sums in index order; the
// backward column sums by the DPP wavefront reduction) - reproducible from run to run, not bit-identical to the
// per-level kernels (different order of summation).
#pragma once

namespace kktdev {

static const int ST_THREADS = 1024, ST_MAXP = 176, ST_MAXB = 192, ST_MAXFRONTS = 128, ST_VEC = 192;
static const int ST_GAVE_UP = 110;  // index into the handle's flags buffer
static const int ST_NU = (ST_MAXP + 15) / 16, ST_NS = ST_MAXB / 64;

static inline size_t st_top_lds_bytes(int maxp) {
  const size_t mlen = ((size_t)maxp * (maxp + 1) / 2 + 1) & ~(size_t)1;
  return sizeof(double) * (mlen + 8 * ST_VEC + 16 * ST_MAXB);
}

// thread 0 of the workgroup: wait until *f is non-zero (returns its value; 0 = gave up)
__device__ __forceinline__ int st_wait(int *f, int *flags) {
  int v = 0, n = 0;
  while ((v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) {
    __builtin_amdgcn_s_sleep(2);
    if (++n > (1 << 20)) {
      __hip_atomic_store(flags + ST_GAVE_UP, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
  }
  return v;
}

__global__ void __launch_bounds__(ST_THREADS)
k_solve_top(DevTree T, const int *__restrict__ nodes, const int *__restrict__ top_idx, int *__restrict__ sync, int ntop,
            const double *__restrict__ panel, const double *__restrict__ linv, const long long *__restrict__ linv_off,
            const double *__restrict__ dinv, const int *__restrict__ ptype, const int *__restrict__ lperm,
            const double *__restrict__ rhs, double *__restrict__ xsol, double *__restrict__ cb, int *__restrict__ flags) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int me = blockIdx.x, node = nodes[me];
  const int p = T.npiv[node], b = T.nbor[node];
  const long long F = p + b;
  const int e0 = T.piv_start[node];
  const double *P = panel + T.panel_off[node];
  const double *W = linv + linv_off[node];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const size_t mlen = ((size_t)p * (p + 1) / 2 + 1) & ~(size_t)1;
  double *Ms = lds;  // column t of M from its diagonal down: Ms[t p - t (t - 1) / 2 + (i - t)]
  double *t1 = Ms + mlen, *tp = t1 + ST_VEC, *y = tp + ST_VEC, *xd = y + ST_VEC, *vv = xd + ST_VEC, *cbs = vv + ST_VEC,
         *x2 = cbs + ST_VEC, *spare = x2 + ST_VEC, *part = spare + ST_VEC;  // part: 16 x 192 (4 x 256 in the products with M)
  (void)spare;
  int *fl_f = sync, *fl_b = sync + ntop;

  // ---- everything that depends on the front alone
  double l[ST_NS][ST_NU];  // L21(64 s + lane, wave + 16 u)
#pragma unroll
  for (int s = 0; s < ST_NS; s++)
#pragma unroll
    for (int u = 0; u < ST_NU; u++) {
      const int i = 64 * s + lane, k = wave + 16 * u;
      const bool ok = i < b && k < p;
      const double x = P[ok ? (long long)k * F + p + i : 0];
      l[s][u] = ok ? x : 0.0;
    }
#pragma unroll
  for (int u = 0; u < ST_NU; u++) {
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
  const int bix = tid < b ? T.bidx[T.bptr[node] + tid] : 0;
  const int c0 = T.child_ptr[node], c1 = T.child_ptr[node + 1];
  const int parent = T.parent[node];
  const int pidx = parent >= 0 ? top_idx[parent] : -1;
  if (tid < ST_VEC) t1[tid] = rv, cbs[tid] = 0.0;
  __syncthreads();

  // ---- forward: t = rhs + children, y = M P t, yd = D^-1 y, contribution = c - L21 y
  int nfc = 0;  // children inside the fused levels
  for (int cc = c0; cc < c1; cc++) {
    const int c = T.child_idx[cc];
    const int bc = T.nbor[c], ci = top_idx[c];
    const int *rel = T.rel + T.bptr[c];
    const double *cbc = cb + T.cb_off[c];
    const int ri = tid < bc ? rel[tid] : -1;  // (borders of fused fronts have at most ST_MAXB rows; others: loop below)
    if (ci >= 0) {
      nfc++;
      if (tid == 0) {
        st_wait(fl_f + ci, flags);
        __hip_atomic_store(fl_f + ci, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      __syncthreads();
    }
    if (ri >= 0) {
      const double v = cbc[tid];
      if (ri < p)
        t1[ri] += v;
      else
        cbs[ri - p] += v;
    }
    for (int j = tid + ST_THREADS; j < bc; j += ST_THREADS) {  // (a child below the fused levels with a long border)
      const int rj = rel[j];
      if (rj < p)
        t1[rj] += cbc[j];
      else
        cbs[rj - p] += cbc[j];
    }
    __syncthreads();
  }
  if (tid < p) tp[tid] = t1[lpk];
  __syncthreads();
  {
    const int i = tid & 255, c = tid >> 8;
    double a0 = 0.0, a1 = 0.0;
    if (i < p) {
      int t = c;
      for (; t + 4 <= i; t += 8) {
        a0 = fma(Ms[t * p - (t * (t - 1)) / 2 - t + i], tp[t], a0);
        a1 = fma(Ms[(t + 4) * p - ((t + 4) * (t + 3)) / 2 - (t + 4) + i], tp[t + 4], a1);
      }
      if (t <= i) a0 = fma(Ms[t * p - (t * (t - 1)) / 2 - t + i], tp[t], a0);
    }
    part[256 * c + i] = a0 + a1;
  }
  __syncthreads();
  if (tid < p) y[tid] = (part[tid] + part[256 + tid]) + (part[512 + tid] + part[768 + tid]);
  __syncthreads();
  if (tid < p) {
    const int kp = pty == 2 ? tid - 1 : min(tid + 1, p - 1);  // partner of a 2x2 pivot
    xd[tid] = pty == 0 ? y[tid] * pd0 : y[tid] * pd0 + y[kp] * pd1;
  }
  if (b > 0) {  // block-uniform
    double yk[ST_NU];
#pragma unroll
    for (int u = 0; u < ST_NU; u++) yk[u] = wave + 16 * u < p ? y[wave + 16 * u] : 0.0;
#pragma unroll
    for (int s = 0; s < ST_NS; s++) {
      double a = 0.0;
#pragma unroll
      for (int u = 0; u < ST_NU; u++) a = fma(l[s][u], yk[u], a);
      part[ST_MAXB * wave + 64 * s + lane] = a;
    }
    __syncthreads();
    if (tid < b) {
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int w = 0; w < 16; w += 2) s0 += part[ST_MAXB * w + tid], s1 += part[ST_MAXB * (w + 1) + tid];
      cb[T.cb_off[node] + tid] = cbs[tid] - (s0 + s1);
    }
    if (pidx >= 0) {  // (a front with a border has a parent; inside this launch when pidx >= 0)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_store(fl_f + me, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  } else if (pidx >= 0 && tid == 0) {
    __hip_atomic_store(fl_f + me, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }

  // ---- backward: v = yd - L21' x(border), x = P' M' v
  if (pidx >= 0) {
    if (tid == 0) {
      if (st_wait(fl_b + pidx, flags)) __hip_atomic_fetch_add(fl_b + pidx, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
  if (tid < ST_VEC) x2[tid] = tid < b ? xsol[bix] : 0.0;
  __syncthreads();
  {
    double xs[ST_NS];
#pragma unroll
    for (int s = 0; s < ST_NS; s++) xs[s] = x2[64 * s + lane];
#pragma unroll
    for (int u = 0; u < ST_NU; u++) {
      double a = 0.0;
#pragma unroll
      for (int s = 0; s < ST_NS; s++) a = fma(l[s][u], xs[s], a);
      a = wave_sum(a);
      const int k = wave + 16 * u;
      if (lane == 0 && k < p) vv[k] = xd[k] - a;
    }
  }
  __syncthreads();
  {
    const int t = tid & 255, c = tid >> 8;
    double a0 = 0.0, a1 = 0.0;
    if (t < p) {
      const double *col = Ms + (t * p - (t * (t - 1)) / 2 - t);
      int i = t + c;
      for (; i + 4 < p; i += 8) {
        a0 = fma(col[i], vv[i], a0);
        a1 = fma(col[i + 4], vv[i + 4], a1);
      }
      if (i < p) a0 = fma(col[i], vv[i], a0);
    }
    part[256 * c + t] = a0 + a1;
  }
  __syncthreads();
  if (tid < p) xsol[e0 + lpk] = (part[tid] + part[256 + tid]) + (part[512 + tid] + part[768 + tid]);
  if (nfc > 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_store(fl_b + me, nfc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace kktdev
