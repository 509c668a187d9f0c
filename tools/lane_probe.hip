// Diagnostic: issue cost (cycles per instruction, one wave alone on its SIMD) of the
// lane-broadcast primitives on gfx950: v_readlane_b32 (uniform dynamic lane) feeding
// an fp64 FMA, ds_bpermute_b32, LDS broadcast read, and independent fp64 FMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#define T0() unsigned long long t0 = __builtin_amdgcn_s_memtime()
#define T1(slot) if (threadIdx.x == 0) st[slot] = __builtin_amdgcn_s_memtime() - t0
__device__ __forceinline__ double bcast_lane(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bperm(double v, int src) {
  const int lo = __builtin_amdgcn_ds_bpermute(src << 2, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(src << 2, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
__global__ void k_fma(double *out, unsigned long long *st, int iters) {
  double x[8];
  for (int q = 0; q < 8; q++) x[q] = threadIdx.x * 1e-9 + q;
  const double y = 1.0000001;
  T0();
  for (int i = 0; i < iters; i++)
#pragma unroll
    for (int q = 0; q < 8; q++) x[q] = fma(x[q], y, 1e-9);
  T1(0);
  double s = 0;
  for (int q = 0; q < 8; q++) s += x[q];
  out[threadIdx.x] = s;
}
// 8 readlane pairs + 8 FMAs per iteration
__global__ void k_readlane(double *out, unsigned long long *st, int iters, int src0) {
  double x[8];
  for (int q = 0; q < 8; q++) x[q] = threadIdx.x * 1e-9 + q;
  T0();
  for (int i = 0; i < iters; i++) {
    const int src = (src0 + i) & 63;
    double c[8];
#pragma unroll
    for (int q = 0; q < 8; q++) c[q] = bcast_lane(x[q], src);
#pragma unroll
    for (int q = 0; q < 8; q++) x[q] = fma(x[q], 0.5, c[(q + 1) & 7]);
  }
  T1(1);
  double s = 0;
  for (int q = 0; q < 8; q++) s += x[q];
  out[threadIdx.x] = s;
}
__global__ void k_bperm(double *out, unsigned long long *st, int iters, int src0) {
  double x[8];
  for (int q = 0; q < 8; q++) x[q] = threadIdx.x * 1e-9 + q;
  T0();
  for (int i = 0; i < iters; i++) {
    const int src = (src0 + i) & 63;
    double c[8];
#pragma unroll
    for (int q = 0; q < 8; q++) c[q] = bperm(x[q], src);
#pragma unroll
    for (int q = 0; q < 8; q++) x[q] = fma(x[q], 0.5, c[(q + 1) & 7]);
  }
  T1(2);
  double s = 0;
  for (int q = 0; q < 8; q++) s += x[q];
  out[threadIdx.x] = s;
}
// owner lane writes 8 doubles to LDS, all lanes read them back (broadcast)
__global__ void k_ldsrow(double *out, unsigned long long *st, int iters, int src0) {
  __shared__ double row[16];
  double x[8];
  for (int q = 0; q < 8; q++) x[q] = threadIdx.x * 1e-9 + q;
  T0();
  for (int i = 0; i < iters; i++) {
    const int src = (src0 + i) & 63;
    if ((int)threadIdx.x == src) {
#pragma unroll
      for (int q = 0; q < 8; q++) row[q] = x[q];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    double c[8];
#pragma unroll
    for (int q = 0; q < 8; q++) c[q] = ((volatile double *)row)[q];
#pragma unroll
    for (int q = 0; q < 8; q++) x[q] = fma(x[q], 0.5, c[(q + 1) & 7]);
  }
  T1(3);
  double s = 0;
  for (int q = 0; q < 8; q++) s += x[q];
  out[threadIdx.x] = s;
}
int main() {
  const int iters = 4000;
  double *out;
  unsigned long long *st, h[8] = {0};
  hipMalloc(&out, sizeof(double) * 64);
  hipMalloc(&st, sizeof(h));
  hipMemset(st, 0, sizeof(h));
  for (int rep = 0; rep < 2; rep++) {
    k_fma<<<1, 64>>>(out, st, iters);
    k_readlane<<<1, 64>>>(out, st, iters, 3);
    k_bperm<<<1, 64>>>(out, st, iters, 3);
    k_ldsrow<<<1, 64>>>(out, st, iters, 3);
    hipDeviceSynchronize();
  }
  hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
  printf("8 indep fp64 FMA / iter:            %.1f cycles per iter (%.1f per FMA)\n", (double)h[0] / iters, (double)h[0] / iters / 8);
  printf("8 readlane pairs + 8 FMA / iter:    %.1f cycles per iter\n", (double)h[1] / iters);
  printf("8 bpermute pairs + 8 FMA / iter:    %.1f cycles per iter\n", (double)h[2] / iters);
  printf("LDS row write + 8 reads + 8 FMA:    %.1f cycles per iter\n", (double)h[3] / iters);
  return 0;
}
