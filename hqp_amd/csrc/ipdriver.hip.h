// Device-resident interior-point loop around the KKT path: the vector work of the
// reference's Mehrotra predictor-corrector solver (hqp/Hqp_IpsMehrotra.C:209-327
// cold_start, :355-693 step, :696-735 solve), restated as kernels over the
// handle's CSR blocks so that x, y, z, w, the right-hand sides and the steps never
// leave the GPU between factor and solve.  The
// scalars that steer the iteration (gap, mu, step lengths, sigma) are reduced on
// the device in a FIXED order (per-block partials, then one block) and read back:
// the iteration is deterministic run to run.
#pragma once

namespace kktdev {

#define IP_BLOCKS 256
#define IP_SLOTS 8
enum { IP_SUM = 0, IP_MAX = 1, IP_MIN = 2 };

// block-level combine of one slot (256 threads), result in thread 0
__device__ __forceinline__ double ip_block_reduce(double v, int op, double *red) {
  v = op == IP_SUM ? wave_sum(v) : (op == IP_MAX ? wave_max(v) : -wave_max(-v));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = red[0];
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; k++) r = op == IP_SUM ? r + red[k] : (op == IP_MAX ? fmax(r, red[k]) : fmin(r, red[k]));
  }
  __syncthreads();
  return r;
}
__device__ __forceinline__ double nan_to_inf(double a) {
  return a == a ? a : __longlong_as_double(0x7ff0000000000000LL);
}

// element formulas shared by the separate launches and the one-workgroup kernels of small QPs (same expression, same
// contraction into multiply-adds: the two forms give the same bits)
__device__ __forceinline__ double ip_corr_elem(double z, double w, double dza, double dwa, double smm) {
  return -(z * w + (dza * dwa - smm));
}
__device__ __forceinline__ double ip_mupl_elem(double z, double w, double dz, double dw, double alpha) {
  return (z + alpha * dz) * (w + alpha * dw);
}
// partials[IP_BLOCKS][IP_SLOTS] -> out[IP_SLOTS], slot k combined with ops[k]
struct IpOps {
  int op[IP_SLOTS];
};
// ---- the scalar logic of a step on the device (thread 0 of the final reduction kernels, no
// launches of its own): the host reads the results
// together with the next iteration's convergence data instead of stopping the stream
// after every reduction.  S: 0 alpha_aff, 1 sigma mu, 2 alpha before damping, 3 damping
// needed (0/1), 4 second corrector needed (0/1), 5 alpha of the step (0 if 4), 6 sigma
#define IPS_ALPHA_AFF 0
#define IPS_SMM 1
#define IPS_ALPHA_PRE 2
#define IPS_DAMP 3
#define IPS_NEED2 4
#define IPS_ALPHA 5
#define IPS_SIGMA 6
// Terlaky's sigma (hqp/Hqp_IpsMehrotra.C:583-590); the safe one when the predictor step
// is short (:612-616, the first corrector is skipped then).  red: 0 min ratio, 1 t
__device__ __forceinline__ void ip_sigma(const double *__restrict__ red, double mu, double gamma, double *__restrict__ S) {
  const double alpha_aff = fmax(0.0, fmin(fmin(1.0, red[0]), 1.0));
  const double t = red[1];
  const double sigma = alpha_aff >= 0.1 ? gamma * (t + 1.0 - alpha_aff) / (1.0 - gamma) : gamma / (1.0 - gamma);
  S[IPS_ALPHA_AFF] = alpha_aff, S[IPS_SIGMA] = sigma, S[IPS_SMM] = sigma * mu;
}
// after the corrector: its own largest step (:604-611) decides about a second corrector
// (:612); Mehrotra's step rule up to the damping (:629-656).  B: k_ip_minratio_final's 12
__device__ __forceinline__ void ip_alpha_pre(const double *__restrict__ B, int m, double gamma, double *__restrict__ S) {
  const double zmin = B[0], wmin = B[6];
  const int izmin = (int)B[1], iwmin = (int)B[7];
  const double amin = fmin(izmin < 0 ? 1e300 : zmin, iwmin < 0 ? 1e300 : wmin);
  const double alpha_corr = fmax(0.0, fmin(fmin(1.0, amin), 1.0));
  S[IPS_NEED2] = (S[IPS_ALPHA_AFF] >= 0.1 && alpha_corr < gamma * gamma / 2.0 / m / m) ? 1.0 : 0.0;
  double alpha;
  if (izmin < 0 && iwmin < 0)
    alpha = 1.0, S[IPS_DAMP] = 0.0;
  else {
    alpha = izmin < 0 ? wmin : iwmin < 0 ? zmin : fmin(zmin, wmin);
    S[IPS_DAMP] = 1.0;
  }
  S[IPS_ALPHA_PRE] = alpha;
}
// the damped step (:657-672).  red: 0 (z + alpha dz)'(w + alpha dw)
__device__ __forceinline__ void ip_alpha_fin(const double *__restrict__ red, const double *__restrict__ B, int m, double gammaf,
                               double *__restrict__ S) {
  double alpha = S[IPS_ALPHA_PRE];
  if (S[IPS_DAMP] != 0.0) {
    const double zmin = B[0], wmin = B[6];
    const int izmin = (int)B[1], iwmin = (int)B[7];
    const double z_iz = B[2], dz_iz = B[3], w_iz = B[4], dw_iz = B[5];
    const double z_iw = B[8], dz_iw = B[9], w_iw = B[10], dw_iw = B[11];
    const double mu_pl = red[0] / m;
    double fpd;
    if (iwmin >= 0 && alpha == wmin && z_iw > -alpha * dz_iw)
      fpd = (gammaf * mu_pl / (z_iw + alpha * dz_iw) - w_iw) / (alpha * dw_iw);
    else if (izmin >= 0 && alpha == zmin && w_iz > -alpha * dw_iz)
      fpd = (gammaf * mu_pl / (w_iz + alpha * dw_iz) - z_iz) / (alpha * dz_iz);
    else
      fpd = 0.0;
    alpha = fmax(0.0, fmin(fmax(1.0 - gammaf, fpd) * alpha, 1.0));
  }
  S[IPS_ALPHA] = S[IPS_NEED2] != 0.0 ? 0.0 : alpha;  // a second corrector first: this step is not taken
}

// what thread 0 does with the reduced values before the kernel ends
struct IpEpi {
  int kind;  // 0 nothing, 1 sigma (k_ip_ratio's reduction), 2 damped step length (k_ip_mupl's),
             // 3 Franke's step length min(1, beta val1) (k_fr_ratio's reduction; beta in gammaf)
  int m;
  double mu, gamma, gammaf;
  const double *B;  // k_ip_minratio_final's 12 values
  double *S;
};
__global__ void __launch_bounds__(256)
k_ip_final(const double *__restrict__ part, IpOps ops, double *__restrict__ out, IpEpi epi) {
  __shared__ double red[4];
  double mine[IP_SLOTS];
  for (int k = 0; k < IP_SLOTS; k++) {
    const double v = part[threadIdx.x * IP_SLOTS + k];  // IP_BLOCKS == blockDim.x
    const double r = ip_block_reduce(v, ops.op[k], red);
    mine[k] = r;
    if (threadIdx.x == 0) out[k] = r;
  }
  if (threadIdx.x == 0) {
    if (epi.kind == 1) ip_sigma(mine, epi.mu, epi.gamma, epi.S);
    if (epi.kind == 2) ip_alpha_fin(mine, epi.B, epi.m, epi.gammaf, epi.S);
    if (epi.kind == 3) epi.S[IPS_ALPHA] = fmin(1.0, epi.gammaf * mine[0]);  // hqp/Hqp_IpsFranke.C:333-334
  }
}

// right-hand sides of an iteration (hqp/Hqp_IpsMehrotra.C:425-447) and the
// quantities of the convergence test:
//   r1 = Qx + c - A'y - C'z, r2 = -(Ax + b), r3 = -(Cx + d - w), r4 = -z.*w
//   slots: 0 gap = x'(Qx+c) + y'b + z'd, 1 pcost, 2 z'w, 3 max(|r1|,|r2|,|r3|), 4 min z, 5 min w
template <int LPR>
__global__ void __launch_bounds__(256)
k_ip_rhs(int n, int me, int m, CsrDev Q, CsrDev AT, CsrDev CT, CsrDev A, CsrDev C,
         const double *__restrict__ vals, const double *__restrict__ c, const double *__restrict__ b,
         const double *__restrict__ d, const double *__restrict__ x, const double *__restrict__ y,
         const double *__restrict__ z, const double *__restrict__ w, double *__restrict__ r1,
         double *__restrict__ r2, double *__restrict__ r3, double *__restrict__ r4,
         double *__restrict__ part,
         // STAGED with dense dynamics: x1 = A_dyn' y (n), x2 = A_dyn x (the first ndyn rows of A, empty in
         // the CSR block), from k_st_dyn_aty / k_st_dyn_ax
         const double *__restrict__ x1 = nullptr, const double *__restrict__ x2 = nullptr, int ndyn = 0) {
  __shared__ double red[4];
  const int sub = threadIdx.x & (LPR - 1);
  constexpr int RPB = 256 / LPR;
  const int total = n + me + m;
  double gap = 0.0, pc = 0.0, zw = 0.0, nr = 0.0, zmin = 1e300, wmin = 1e300;
  for (int q = blockIdx.x * RPB + threadIdx.x / LPR; q < total; q += gridDim.x * RPB) {
    if (q < n) {
      const double qx = row_dot<LPR>(Q, vals, x, q, sub);
      const double g = qx + c[q];
      const double s = g - row_dot<LPR>(AT, vals, y, q, sub) - row_dot<LPR>(CT, vals, z, q, sub) - (x1 ? x1[q] : 0.0);
      if (sub == 0) {
        r1[q] = s;
        gap += x[q] * g, pc += x[q] * (0.5 * qx + c[q]);
        nr = fmax(nr, nan_to_inf(fabs(s)));
      }
    } else if (q < n + me) {
      const int i = q - n;
      const double s = -(row_dot<LPR>(A, vals, x, i, sub) + (i < ndyn ? x2[i] : 0.0) + b[i]);
      if (sub == 0) {
        r2[i] = s;
        gap += y[i] * b[i];
        nr = fmax(nr, nan_to_inf(fabs(s)));
      }
    } else {
      const int j = q - n - me;
      const double s = -(row_dot<LPR>(C, vals, x, j, sub) + d[j] - w[j]);
      if (sub == 0) {
        r3[j] = s, r4[j] = -(z[j] * w[j]);
        gap += z[j] * d[j], zw += z[j] * w[j];
        nr = fmax(nr, nan_to_inf(fabs(s)));
        zmin = fmin(zmin, z[j]), wmin = fmin(wmin, w[j]);
      }
    }
  }
  double *P = part + blockIdx.x * IP_SLOTS;
  double r;
  r = ip_block_reduce(gap, IP_SUM, red); if (threadIdx.x == 0) P[0] = r;
  r = ip_block_reduce(pc, IP_SUM, red);  if (threadIdx.x == 0) P[1] = r;
  r = ip_block_reduce(zw, IP_SUM, red);  if (threadIdx.x == 0) P[2] = r;
  r = ip_block_reduce(nr, IP_MAX, red);  if (threadIdx.x == 0) P[3] = r;
  r = ip_block_reduce(zmin, IP_MIN, red); if (threadIdx.x == 0) P[4] = r;
  r = ip_block_reduce(wmin, IP_MIN, red); if (threadIdx.x == 0) P[5] = r, P[6] = 0.0, P[7] = 0.0;
}

// largest feasible step (hqp/Hqp_IpsMehrotra.C:566-572, 605-611) and Terlaky's
// t = max dz dw / (z w) over dz dw > 0 (:584-588)
//   slots: 0 min ratio (1e300 if none), 1 t
__global__ void __launch_bounds__(256)
k_ip_ratio(int m, const double *__restrict__ z, const double *__restrict__ w, const double *__restrict__ dz,
           const double *__restrict__ dw, double *__restrict__ part) {
  __shared__ double red[4];
  double a = 1e300, t = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
    const double zi = z[i], wi = w[i], dzi = dz[i], dwi = dw[i];
    if (dzi < 0.0) a = fmin(a, -zi / dzi);
    if (dwi < 0.0) a = fmin(a, -wi / dwi);
    if (dzi * dwi > 0.0) t = fmax(t, dzi * dwi / zi / wi);
  }
  double *P = part + blockIdx.x * IP_SLOTS;
  double r;
  r = ip_block_reduce(a, IP_MIN, red); if (threadIdx.x == 0) P[0] = r;
  r = ip_block_reduce(t, IP_MAX, red);
  if (threadIdx.x == 0) {
    P[1] = r;
    for (int k = 2; k < IP_SLOTS; k++) P[k] = 0.0;
  }
}

// corrector right-hand side r4 = -(z.*w + dza.*dwa - sigma mu)  (:596-600, :617-622)
__global__ void k_ip_corr_rhs(int m, const double *__restrict__ z, const double *__restrict__ w,
                              const double *__restrict__ dza, const double *__restrict__ dwa, double smm,
                              const double *__restrict__ smm_dev, double *__restrict__ r4) {
  if (smm_dev) smm = *smm_dev;  // sigma mu computed on the device (k_ip_sigma): no host round trip
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) r4[i] = ip_corr_elem(z[i], w[i], dza[i], dwa[i], smm);
}

// Mehrotra's adaptive step (:629-646): the blocking component of z and of w, first
// index of the minimum as the reference's loop finds it.  One block; out: zmin,
// izmin, wmin, iwmin (indices as doubles, -1 if none).
__global__ void __launch_bounds__(256)
k_ip_minratio_part(int m, const double *__restrict__ z, const double *__restrict__ w,
                   const double *__restrict__ dz, const double *__restrict__ dw, double *__restrict__ part) {
  __shared__ double sv[256];
  __shared__ int si[256];
  double zv = 1e300, wv = 1e300;
  int zi = 0x7fffffff, wi = 0x7fffffff;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
    if (dz[i] < 0.0) {
      const double q = -z[i] / dz[i];
      if (q < zv || (q == zv && i < zi)) zv = q, zi = i;
    }
    if (dw[i] < 0.0) {
      const double q = -w[i] / dw[i];
      if (q < wv || (q == wv && i < wi)) wv = q, wi = i;
    }
  }
  double *P = part + blockIdx.x * IP_SLOTS;
  for (int pass = 0; pass < 2; pass++) {
    sv[threadIdx.x] = pass ? wv : zv, si[threadIdx.x] = pass ? wi : zi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (threadIdx.x < s) {
        const double ov = sv[threadIdx.x + s];
        const int oi = si[threadIdx.x + s];
        if (ov < sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x]))
          sv[threadIdx.x] = ov, si[threadIdx.x] = oi;
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) P[2 * pass] = sv[0], P[2 * pass + 1] = (double)si[0];
    __syncthreads();
  }
}
__global__ void __launch_bounds__(256)
k_ip_minratio_final(const double *__restrict__ part, const double *__restrict__ z, const double *__restrict__ w,
                    const double *__restrict__ dz, const double *__restrict__ dw, double *__restrict__ out, int m,
                    double gamma, double *__restrict__ S) {  // S != nullptr: Mehrotra's step rule up to the damping
  __shared__ double sv[256];
  __shared__ int si[256];
  for (int pass = 0; pass < 2; pass++) {
    sv[threadIdx.x] = part[threadIdx.x * IP_SLOTS + 2 * pass];
    si[threadIdx.x] = (int)part[threadIdx.x * IP_SLOTS + 2 * pass + 1];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (threadIdx.x < s) {
        const double ov = sv[threadIdx.x + s];
        const int oi = si[threadIdx.x + s];
        if (ov < sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x]))
          sv[threadIdx.x] = ov, si[threadIdx.x] = oi;
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      const bool none = !(sv[0] < 1e300);
      const int i = none ? -1 : si[0];
      out[6 * pass] = sv[0], out[6 * pass + 1] = (double)i;
      // the four components the damping rule looks at (:660-668)
      out[6 * pass + 2] = none ? 0.0 : z[i], out[6 * pass + 3] = none ? 0.0 : dz[i];
      out[6 * pass + 4] = none ? 0.0 : w[i], out[6 * pass + 5] = none ? 0.0 : dw[i];
      if (pass == 1 && S) ip_alpha_pre(out, m, gamma, S);  // same thread wrote all twelve values
    }
    __syncthreads();
  }
}

// slot 0: (z + alpha dz)'(w + alpha dw)  (:657-659)
__global__ void __launch_bounds__(256)
k_ip_mupl(int m, double alpha, const double *__restrict__ alpha_dev, const double *__restrict__ z,
          const double *__restrict__ w, const double *__restrict__ dz, const double *__restrict__ dw,
          double *__restrict__ part) {
  __shared__ double red[4];
  if (alpha_dev) alpha = *alpha_dev;
  double s = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x)
    s += ip_mupl_elem(z[i], w[i], dz[i], dw[i], alpha);
  const double r = ip_block_reduce(s, IP_SUM, red);
  if (threadIdx.x == 0) {
    double *P = part + blockIdx.x * IP_SLOTS;
    P[0] = r;
    for (int k = 1; k < IP_SLOTS; k++) P[k] = 0.0;
  }
}

// the step (:677-680): x, y, z, w += alpha d*;  slots: 0 z'w, 1 max|x| (NaN -> inf)
__global__ void __launch_bounds__(256)
k_ip_update(int n, int me, int m, double alpha, const double *__restrict__ alpha_dev, double *__restrict__ x,
            double *__restrict__ y, double *__restrict__ z, double *__restrict__ w, const double *__restrict__ dx,
            const double *__restrict__ dy, const double *__restrict__ dz, const double *__restrict__ dw,
            double *__restrict__ part) {
  __shared__ double red[4];
  if (alpha_dev) alpha = *alpha_dev;
  double zw = 0.0, xm = 0.0;
  const int total = n + me + m;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < total; q += gridDim.x * blockDim.x) {
    if (q < n) {
      const double v = x[q] + alpha * dx[q];
      x[q] = v;
      xm = fmax(xm, nan_to_inf(fabs(v)));
    } else if (q < n + me) {
      y[q - n] += alpha * dy[q - n];
    } else {
      const int j = q - n - me;
      const double zn = z[j] + alpha * dz[j], wn = w[j] + alpha * dw[j];
      z[j] = zn, w[j] = wn;
      zw += zn * wn;
    }
  }
  double *P = part + blockIdx.x * IP_SLOTS;
  double r;
  r = ip_block_reduce(zw, IP_SUM, red); if (threadIdx.x == 0) P[0] = r;
  r = ip_block_reduce(xm, IP_MAX, red);
  if (threadIdx.x == 0) {
    P[1] = r;
    for (int k = 2; k < IP_SLOTS; k++) P[k] = 0.0;
  }
}

// ---- small QPs: the vector work between the solves of an iteration in ONE workgroup ---------------------
// With m <= IP_SMALL_M (the double-integrator QP at K = 2000: 8000) the five launches between the predictor solve and the
// corrector solve - k_ip_ratio, k_ip_final, k_ip_corr_rhs - and the five behind the corrector solve - k_ip_minratio_part,
// _final, k_ip_mupl, k_ip_final, k_ip_update - are each a few microseconds of launch and first-touch latency around a
// few thousand elements (profiles/r05_ip_did_kstat.txt: 54 us of an iteration's 410).  k_ip_pred_small / k_ip_step_small
// do the same work in one workgroup of 1024 threads each, with the SAME arithmetic: minima and maxima do not depend on
// the order; the one sum - (z + alpha dz)'(w + alpha dw) - is added up exactly as the 256 blocks of 256 threads and
// k_ip_final add it (virtual wavefront by virtual wavefront with the same butterfly, the four wavefronts of a block in
// their order, then the 256 block sums the same way), so the loop's iterates are bit for bit those of the separate
// launches (tests/test_gpu_franke.py::test_small_ip_kernels_bit_identical; HQPKKT_NO_IP_SMALL=1 keeps the launches).
#define IP_SMALL_M 8192  // at most IP_SMALL_E elements of an m-vector per thread: z, w and the step stay in registers over the phases
#define IP_SMALL_E 8
// combine over the workgroup's 1024 threads (any order: minima / maxima only); every thread gets the result
__device__ __forceinline__ double ip_small_minmax(double v, bool is_max, double *red16) {
  v = is_max ? wave_max(v) : -wave_max(-v);
  if ((threadIdx.x & 63) == 0) red16[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = red16[0];
#pragma unroll
  for (int k = 1; k < 16; k++) r = is_max ? fmax(r, red16[k]) : fmin(r, red16[k]);
  __syncthreads();
  return r;
}
// element tid + 1024 u of four m-vectors (u < IP_SMALL_E), all loads of a thread in flight together (a loop of dependent
// loads - one memory latency per trip - was most of these kernels' time); beyond m: element 0, not used
#define IP_SMALL_LOAD(A, B, Cc, D, a, b, c, d)                                 \
  double a[IP_SMALL_E], b[IP_SMALL_E], c[IP_SMALL_E], d[IP_SMALL_E];           \
  _Pragma("unroll") for (int u = 0; u < IP_SMALL_E; u++) {                     \
    const int i_ = (int)threadIdx.x + 1024 * u, ic_ = i_ < m ? i_ : 0;         \
    a[u] = A[ic_], b[u] = B[ic_], c[u] = Cc[ic_], d[u] = D[ic_];               \
  }
__global__ void __launch_bounds__(1024)
k_ip_pred_small(int m, const double *__restrict__ z, const double *__restrict__ w, const double *__restrict__ dza,
                const double *__restrict__ dwa, const double *__restrict__ gap_sum, double gamma, double *__restrict__ S,
                double *__restrict__ r4) {
  __shared__ double red16[16];
  __shared__ double bc[2];
  // mu = z'w / m of this iterate, from the sum the iteration's first reduction has left on the device (the same
  // division the host makes with the word it read back: the kernel takes no value of the host, so that it can be
  // replayed inside a captured graph)
  const double mu = *gap_sum / m;
  IP_SMALL_LOAD(z, w, dza, dwa, zz, ww, dzz, dww)
  // k_ip_ratio + k_ip_final + ip_sigma
  double a = 1e300, t = 0.0;
#pragma unroll
  for (int u = 0; u < IP_SMALL_E; u++)
    if ((int)threadIdx.x + 1024 * u < m) {
      const double zi = zz[u], wi = ww[u], dzi = dzz[u], dwi = dww[u];
      if (dzi < 0.0) a = fmin(a, -zi / dzi);
      if (dwi < 0.0) a = fmin(a, -wi / dwi);
      if (dzi * dwi > 0.0) t = fmax(t, dzi * dwi / zi / wi);
    }
  const double amin = ip_small_minmax(a, false, red16), tmax = ip_small_minmax(t, true, red16);
  if (threadIdx.x == 0) {
    const double two[2] = {amin, tmax};
    ip_sigma(two, mu, gamma, S);
    bc[0] = S[IPS_SMM];
  }
  __syncthreads();
  // k_ip_corr_rhs
  const double smm = bc[0];
#pragma unroll
  for (int u = 0; u < IP_SMALL_E; u++) {
    const int i = (int)threadIdx.x + 1024 * u;
    if (i < m) r4[i] = ip_corr_elem(zz[u], ww[u], dzz[u], dww[u], smm);
  }
}

__global__ void __launch_bounds__(1024)
k_ip_step_small(int n, int me, int m, double *__restrict__ x, double *__restrict__ y, double *__restrict__ z,
                double *__restrict__ w, const double *__restrict__ dx, const double *__restrict__ dy,
                const double *__restrict__ dz, const double *__restrict__ dw, double *__restrict__ Bk, double gamma,
                double gammaf, double *__restrict__ S) {
  __shared__ double sv[16];
  __shared__ int si[16];
  __shared__ double ws[1024];  // sums of the virtual wavefronts of k_ip_mupl's 256 x 256 launch
  __shared__ double red4[4];
  __shared__ double bc[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  ws[tid] = 0.0;  // (virtual wavefronts behind m: a sum of zeros)
  IP_SMALL_LOAD(z, w, dz, dw, zz, ww, dzz, dww)
  // ---- k_ip_minratio_part + _final: the blocking components (smallest ratio, smallest index among equals)
  double zv = 1e300, wv = 1e300;
  int zi = 0x7fffffff, wi = 0x7fffffff;
#pragma unroll
  for (int u = 0; u < IP_SMALL_E; u++) {
    const int i = tid + 1024 * u;
    if (i < m) {
      if (dzz[u] < 0.0) {
        const double q = -zz[u] / dzz[u];
        if (q < zv || (q == zv && i < zi)) zv = q, zi = i;
      }
      if (dww[u] < 0.0) {
        const double q = -ww[u] / dww[u];
        if (q < wv || (q == wv && i < wi)) wv = q, wi = i;
      }
    }
  }
  double bk[12];
  for (int pass = 0; pass < 2; pass++) {
    double v = pass ? wv : zv;
    int ix = pass ? wi : zi;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double ov = __shfl_xor(v, o);
      const int oi = __shfl_xor(ix, o);
      if (ov < v || (ov == v && oi < ix)) v = ov, ix = oi;
    }
    if (lane == 0) sv[wave] = v, si[wave] = ix;
    __syncthreads();
    if (tid == 0) {
      for (int k = 1; k < 16; k++)
        if (sv[k] < v || (sv[k] == v && si[k] < ix)) v = sv[k], ix = si[k];
      const bool none = !(v < 1e300);
      const int i = none ? -1 : ix;
      bk[6 * pass] = v, bk[6 * pass + 1] = (double)i;
      bk[6 * pass + 2] = none ? 0.0 : z[i], bk[6 * pass + 3] = none ? 0.0 : dz[i];
      bk[6 * pass + 4] = none ? 0.0 : w[i], bk[6 * pass + 5] = none ? 0.0 : dw[i];
    }
    __syncthreads();
  }
  if (tid == 0) {
    for (int k = 0; k < 12; k++) Bk[k] = bk[k];
    ip_alpha_pre(bk, m, gamma, S);
    bc[0] = S[IPS_ALPHA_PRE];
  }
  __syncthreads();
  // ---- k_ip_mupl + k_ip_final: (z + alpha dz)'(w + alpha dw), in the order of the 256 x 256 launch (m <= IP_SMALL_M: a
  // thread of that launch holds at most one element; element tid + 1024 u belongs to lane `lane` of its virtual
  // wavefront wave + 16 u)
  const double alpha_pre = bc[0];
#pragma unroll
  for (int u = 0; u < IP_SMALL_E; u++) {
    const int v = wave + 16 * u;
    if (64 * v < m) {  // wave-uniform
      double s = 0.0;
      if (tid + 1024 * u < m) s += ip_mupl_elem(zz[u], ww[u], dzz[u], dww[u], alpha_pre);
      s = wave_sum(s);
      if (lane == 0) ws[v] = s;
    }
  }
  __syncthreads();
  {
    double bs = 0.0;
    if (tid < 256) {  // the block sums: the four wavefronts of a block in their order (ip_block_reduce)
      bs = ws[4 * tid];
#pragma unroll
      for (int k = 1; k < 4; k++) bs = bs + ws[4 * tid + k];
      bs = wave_sum(bs);  // k_ip_final: thread t holds block t's sum
      if (lane == 0) red4[wave] = bs;
    }
    __syncthreads();
    if (tid == 0) {
      double r = red4[0];
      for (int k = 1; k < 4; k++) r = r + red4[k];
      const double one[1] = {r};
      ip_alpha_fin(one, bk, m, gammaf, S);
      bc[1] = S[IPS_ALPHA];
    }
    __syncthreads();
  }
  // ---- k_ip_update
  const double alpha = bc[1];
#pragma unroll
  for (int u = 0; u < IP_SMALL_E; u++) {
    const int i = tid + 1024 * u;
    if (i < m) z[i] = zz[u] + alpha * dzz[u], w[i] = ww[u] + alpha * dww[u];
  }
#pragma unroll 4
  for (int q = tid; q < n; q += 1024) x[q] = x[q] + alpha * dx[q];
#pragma unroll 4
  for (int q = tid; q < me; q += 1024) y[q] = y[q] + alpha * dy[q];
}

// cold start (:226-250): z = 1, w = w0 (1; a norm ratio with qp_init_method 1, 2), r1 = c,
// r2 = -b, r3 = -d, r4 = r40 (0; -z.*w with qp_init_method 1-3)
__global__ void k_ip_cold_rhs(int n, int me, int m, const double *__restrict__ c, const double *__restrict__ b,
                              const double *__restrict__ d, double *__restrict__ z, double *__restrict__ w,
                              double *__restrict__ r1, double *__restrict__ r2, double *__restrict__ r3,
                              double *__restrict__ r4, double w0, double r40) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n)
    r1[q] = c[q];
  else if (q < n + me)
    r2[q - n] = -b[q - n];
  else if (q < n + me + m) {
    const int j = q - n - me;
    z[j] = 1.0, w[j] = w0, r3[j] = -d[j], r4[j] = r40;
  }
}
// slots: 0 min dz, 1 min dw, 2 max|dz|, 3 max|dw|, 4 sum dz, 5 sum dw  (:305-310)
__global__ void __launch_bounds__(256)
k_ip_cold_stats(int m, const double *__restrict__ dz, const double *__restrict__ dw, double *__restrict__ part) {
  __shared__ double red[4];
  double a = 1e300, bq = 1e300, ma = 0.0, mb = 0.0, sa = 0.0, sb = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
    a = fmin(a, dz[i]), bq = fmin(bq, dw[i]);
    ma = fmax(ma, fabs(dz[i])), mb = fmax(mb, fabs(dw[i]));
    sa += dz[i], sb += dw[i];
  }
  double *P = part + blockIdx.x * IP_SLOTS;
  double r;
  r = ip_block_reduce(a, IP_MIN, red);  if (threadIdx.x == 0) P[0] = r;
  r = ip_block_reduce(bq, IP_MIN, red); if (threadIdx.x == 0) P[1] = r;
  r = ip_block_reduce(ma, IP_MAX, red); if (threadIdx.x == 0) P[2] = r;
  r = ip_block_reduce(mb, IP_MAX, red); if (threadIdx.x == 0) P[3] = r;
  r = ip_block_reduce(sa, IP_SUM, red); if (threadIdx.x == 0) P[4] = r;
  r = ip_block_reduce(sb, IP_SUM, red); if (threadIdx.x == 0) P[5] = r, P[6] = 0.0, P[7] = 0.0;
}
// z = dz + delz, w = dw + delw  (:316-319)
__global__ void k_ip_shift(int m, const double *__restrict__ dz, const double *__restrict__ dw, double delz,
                           double delw, double *__restrict__ z, double *__restrict__ w) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) z[i] = dz[i] + delz, w[i] = dw[i] + delw;
}
__global__ void k_ip_fill(int n, double v, double *__restrict__ x) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = v;
}

// ---------------------------------------------------------------- Franke's solver
// Device kernels of hqpkkt_franke, the restatement of hqp/Hqp_IpsFranke.C (potential
// reduction method with the infeasibility measure zeta; cold_start :156-216, step :271-378).
//   slots: 0 min d, 1 max|d|, 2 sum d
__global__ void __launch_bounds__(256) k_fr_dstats(int m, const double *__restrict__ d, double *__restrict__ part) {
  __shared__ double red[4];
  double a = 1e300, b = 0.0, c = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x)
    a = fmin(a, d[i]), b = fmax(b, fabs(d[i])), c += d[i];
  double *P = part + blockIdx.x * IP_SLOTS;
  double r;
  r = ip_block_reduce(a, IP_MIN, red); if (threadIdx.x == 0) P[0] = r;
  r = ip_block_reduce(b, IP_MAX, red); if (threadIdx.x == 0) P[1] = r;
  r = ip_block_reduce(c, IP_SUM, red);
  if (threadIdx.x == 0) {
    P[2] = r;
    for (int k = 3; k < IP_SLOTS; k++) P[k] = 0.0;
  }
}
// cold start (:186-203): x = y = 0, z = Ltilde / m^2, w = Ltilde + d + 1e-10,
// a1 = c - C'z, a2 = -b, a3 = Ltilde (zeta = 1); slot 0: z'w.  m = 0: a1 = c, a2 = -b.
template <int LPR>
__global__ void __launch_bounds__(256)
k_fr_cold(int n, int me, int m, CsrDev CT, double Ltilde, const double *__restrict__ c, const double *__restrict__ b,
          const double *__restrict__ d, double *__restrict__ x, double *__restrict__ y, double *__restrict__ z,
          double *__restrict__ w, double *__restrict__ a1, double *__restrict__ a2, double *__restrict__ a3,
          double *__restrict__ part) {
  __shared__ double red[4];
  const int sub = threadIdx.x & (LPR - 1);
  constexpr int RPB = 256 / LPR;
  const int total = n + me + m;
  const double z0 = m > 0 ? Ltilde / ((double)m * m) : 0.0;
  double zw = 0.0;
  for (int q = blockIdx.x * RPB + threadIdx.x / LPR; q < total; q += gridDim.x * RPB) {
    if (q < n) {
      double s = 0.0;  // (C'z)_q with the constant z
      if (m > 0) {
        const int e = CT.ptr[q + 1];
        for (int k = CT.ptr[q] + sub; k < e; k += LPR) s += CT.val[k];
        s = row_sum<LPR>(s) * z0;
      }
      if (sub == 0) x[q] = 0.0, a1[q] = c[q] - s;
    } else if (q < n + me) {
      if (sub == 0) y[q - n] = 0.0, a2[q - n] = -b[q - n];
    } else if (sub == 0) {
      const int j = q - n - me;
      const double wj = Ltilde + d[j] + 1e-10;
      z[j] = z0, w[j] = wj, a3[j] = Ltilde;
      zw += z0 * wj;
    }
  }
  const double r = ip_block_reduce(zw, IP_SUM, red);
  if (threadIdx.x == 0) {
    double *P = part + blockIdx.x * IP_SLOTS;
    P[0] = r;
    for (int k = 1; k < IP_SLOTS; k++) P[k] = 0.0;
  }
}
// right-hand sides of a step (:291-299): r1..r3 = -zeta a1..a3, r4 = z.*w - mu
// (zeta and mu - the host's scalars of the step - come through two words of mapped host memory, zm[0] and zm[1], read by
// one thread of every workgroup: the launch takes no value that changes from step to step and can be replayed inside a
// captured graph.  The host writes them before it launches and not again before the step's read-back has arrived.)
__global__ void __launch_bounds__(256)
k_fr_rhs(int n, int me, int m, const double *__restrict__ zm, const double *__restrict__ a1,
         const double *__restrict__ a2, const double *__restrict__ a3, const double *__restrict__ z,
         const double *__restrict__ w, double *__restrict__ r1, double *__restrict__ r2,
         double *__restrict__ r3, double *__restrict__ r4) {
  __shared__ double zmu[2];
  if (threadIdx.x < 2)
    zmu[threadIdx.x] = __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)zm + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
  __syncthreads();
  const double zeta = zmu[0], mu = zmu[1];
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n)
    r1[q] = -zeta * a1[q];
  else if (q < n + me)
    r2[q - n] = -zeta * a2[q - n];
  else if (q < n + me + m) {
    const int j = q - n - me;
    r3[j] = -zeta * a3[j], r4[j] = z[j] * w[j] - mu;
  }
}
// maximal feasible step (:315-331): slot 0 = min(2, min over dz_i > 0 of z_i / dz_i, same for w);
// the reference's running test "z_i < val dz_i" with dz_i >= 0 is exactly this minimum
__global__ void __launch_bounds__(256)
k_fr_ratio(int m, const double *__restrict__ z, const double *__restrict__ w, const double *__restrict__ dz,
           const double *__restrict__ dw, double *__restrict__ part) {
  __shared__ double red[4];
  double a = 2.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
    if (dz[i] > 0.0) a = fmin(a, z[i] / dz[i]);
    if (dw[i] > 0.0) a = fmin(a, w[i] / dw[i]);
  }
  const double r = ip_block_reduce(a, IP_MIN, red);
  if (threadIdx.x == 0) {
    double *P = part + blockIdx.x * IP_SLOTS;
    P[0] = r;
    for (int k = 1; k < IP_SLOTS; k++) P[k] = 0.0;
  }
}
// the step (:343-349): x, y, z, w -= alpha d*; slots: 0 z'w, 1 max|x| (NaN -> inf)
__global__ void __launch_bounds__(256)
k_fr_update(int n, int me, int m, double alpha, const double *__restrict__ alpha_dev, double *__restrict__ x,
            double *__restrict__ y, double *__restrict__ z, double *__restrict__ w, const double *__restrict__ dx,
            const double *__restrict__ dy, const double *__restrict__ dz, const double *__restrict__ dw,
            double *__restrict__ part) {
  __shared__ double red[4];
  if (alpha_dev) alpha = *alpha_dev;
  double zw = 0.0, xm = 0.0;
  const int total = n + me + m;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < total; q += gridDim.x * blockDim.x) {
    if (q < n) {
      const double v = x[q] - alpha * dx[q];
      x[q] = v;
      xm = fmax(xm, nan_to_inf(fabs(v)));
    } else if (q < n + me) {
      y[q - n] -= alpha * dy[q - n];
    } else {
      const int j = q - n - me;
      const double zn = z[j] - alpha * dz[j], wn = w[j] - alpha * dw[j];
      z[j] = zn, w[j] = wn;
      zw += zn * wn;
    }
  }
  double *P = part + blockIdx.x * IP_SLOTS;
  double r;
  r = ip_block_reduce(zw, IP_SUM, red); if (threadIdx.x == 0) P[0] = r;
  r = ip_block_reduce(xm, IP_MAX, red);
  if (threadIdx.x == 0) {
    P[1] = r;
    for (int k = 2; k < IP_SLOTS; k++) P[k] = 0.0;
  }
}

}  // namespace kktdev
