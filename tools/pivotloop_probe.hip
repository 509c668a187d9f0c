// Diagnostic: cost model of one pivot step of k_factor_diag (512 threads, 8x4 patch in
// registers, column published through LDS), measured in shader cycles per step.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double PatchT[8][4];
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
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
__device__ __forceinline__ double fast_rcp(double d) {
  double x = __builtin_amdgcn_rcp(d);
  x = fma(fma(-d, x, 1.0), x, x);
  x = fma(fma(-d, x, 1.0), x, x);
  return x;
}
template <int MODE>
__global__ void __launch_bounds__(512) probe(const double *in, double *out, unsigned long long *st, int p, double alpha) {
  __shared__ double b0[128], b1[128];
  const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5, lane = tid & 63;
  PatchT A;
#pragma unroll
  for (int m = 0; m < 8; m++)
#pragma unroll
    for (int n = 0; n < 4; n++) A[m][n] = in[(ty + 16 * m) * 128 + tx + 32 * n];
  if (tx == 0)
    for (int m = 0; m < 8; m++) b0[ty + 16 * m] = A[m][0];
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int k = 0, step = 0;
  while (k < p) {
    k = __builtin_amdgcn_readfirstlane(k);
    double *cur = (step & 1) ? b1 : b0, *nxt = (step & 1) ? b0 : b1;
    int r = p;
    double lambda = 0.0, akk = 1.0;
    if (MODE & 1) {  // decision
      const int i1 = k + 1 + lane, i2 = i1 + 64;
      const double v1 = cur[i1 & 127], v2 = cur[i2 & 127], vkk = cur[k];
      const float t1 = i1 < p ? fabsf((float)v1) : -1.0f;
      const float t2 = i2 < p ? fabsf((float)v2) : -1.0f;
      const float tmax = wave_max_dpp_f(fmaxf(fmaxf(t1, t2), 0.0f));
      const unsigned long long m1 = __ballot(t1 == tmax), m2 = __ballot(t2 == tmax);
      if (m1) r = k + 1 + __builtin_ctzll(m1);
      else if (m2) r = k + 65 + __builtin_ctzll(m2);
      r = __builtin_amdgcn_readfirstlane(r);
      akk = fabs(vkk);
      lambda = r < p ? fabs(cur[r]) : 0.0;
    }
    if ((MODE & 1) && r < p && !(akk >= alpha * lambda)) {
      out[0] = 1.0;  // never in this probe (diagonally dominant input)
    }
    double cv[8], cj[4];
#pragma unroll
    for (int m = 0; m < 8; m++) cv[m] = cur[ty + 16 * m];
#pragma unroll
    for (int n = 0; n < 4; n++) cj[n] = cur[tx + 32 * n];
    double d = cur[k];
    const double di = (MODE & 2) ? fast_rcp(d) : d;
    double lj[4];
#pragma unroll
    for (int n = 0; n < 4; n++) lj[n] = (tx + 32 * n > k) ? cj[n] * di : 0.0;
    if (MODE & 4) {
#pragma unroll
      for (int m = 0; m < 8; m++) {
        const double ck = (ty + 16 * m > k) ? cv[m] : 0.0;
#pragma unroll
        for (int n = 0; n < 4; n++) A[m][n] = fma(-ck, lj[n], A[m][n]);
      }
    } else {
      A[0][0] += lj[0] + lj[1] + lj[2] + lj[3] + cv[0] + cv[1] + cv[2] + cv[3] + cv[4] + cv[5] + cv[6] + cv[7];
    }
    k += 1;
    if (MODE & 8) {  // publish column k (strip by uniform branch)
      const int cn = k >> 5;
      const bool mine = tx == (k & 31);
#define CASE(N) { _Pragma("unroll") for (int m = 0; m < 8; m++) { double v = A[m][N]; asm volatile("" : "+v"(v)); if (mine) nxt[ty + 16 * m] = v; } }
      if (cn == 0) CASE(0) else if (cn == 1) CASE(1) else if (cn == 2) CASE(2) else CASE(3)
    } else if (tx == 0) {
      nxt[ty] = A[0][0];
    }
    step++;
    __syncthreads();
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) st[MODE] = t1 - t0;
#pragma unroll
  for (int m = 0; m < 8; m++)
#pragma unroll
    for (int n = 0; n < 4; n++) out[(ty + 16 * m) * 128 + tx + 32 * n] = A[m][n];
}
int main() {
  const int p = 120;
  double *in, *out;
  unsigned long long *st, h[16];
  hipMalloc(&in, 128 * 128 * 8), hipMalloc(&out, 128 * 128 * 8), hipMalloc(&st, sizeof(h));
  double *hin = new double[128 * 128];
  for (int i = 0; i < 128; i++)
    for (int j = 0; j < 128; j++) hin[i * 128 + j] = i == j ? 300.0 + i : 0.01 * ((i * 7 + j * 13) % 17 - 8);
  hipMemcpy(in, hin, 128 * 128 * 8, hipMemcpyHostToDevice);
  hipMemset(st, 0, sizeof(h));
  for (int rep = 0; rep < 2; rep++) {
    probe<0><<<1, 512>>>(in, out, st, p, 0.64);
    probe<1><<<1, 512>>>(in, out, st, p, 0.64);
    probe<2><<<1, 512>>>(in, out, st, p, 0.64);
    probe<4><<<1, 512>>>(in, out, st, p, 0.64);
    probe<8><<<1, 512>>>(in, out, st, p, 0.64);
    probe<15><<<1, 512>>>(in, out, st, p, 0.64);
    probe<14><<<1, 512>>>(in, out, st, p, 0.64);
    hipDeviceSynchronize();
  }
  hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
  printf("cycles per pivot step (p=%d): skeleton %.0f | +decision %.0f | +rcp %.0f | +fma %.0f | +publish %.0f | all %.0f | all-but-decision %.0f\n", p,
         (double)h[0] / p, (double)h[1] / p, (double)h[2] / p, (double)h[4] / p, (double)h[8] / p, (double)h[15] / p, (double)h[14] / p);
  return 0;
}
