// Diagnostic micro-probes for gfx950 fp64 costs (cycles from s_memtime; one wave per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define T0() unsigned long long t0 = __builtin_amdgcn_s_memtime()
#define T1(slot) if (threadIdx.x == 0) st[slot] = __builtin_amdgcn_s_memtime() - t0
__global__ void dep_fma(double *out, unsigned long long *st, int iters) {
  double x = threadIdx.x * 1e-9, y = 1.0000001;
  T0();
  for (int i = 0; i < iters; i++) x = fma(x, y, 1e-9);
  T1(0);
  out[threadIdx.x] = x;
}
__global__ void indep_fma(double *out, unsigned long long *st, int iters) {
  double x[8], y = 1.0000001;
  for (int q = 0; q < 8; q++) x[q] = threadIdx.x * 1e-9 + q;
  T0();
  for (int i = 0; i < iters; i++)
#pragma unroll
    for (int q = 0; q < 8; q++) x[q] = fma(x[q], y, 1e-9);
  T1(1);
  double s = 0;
  for (int q = 0; q < 8; q++) s += x[q];
  out[threadIdx.x] = s;
}
__global__ void dep_div(double *out, unsigned long long *st, int iters) {
  double x = 1.0 + threadIdx.x * 1e-9;
  T0();
  for (int i = 0; i < iters; i++) x = 1.0 / (x + 0.5);
  T1(2);
  out[threadIdx.x] = x;
}
__global__ void lds_chain(double *out, unsigned long long *st, int iters) {
  __shared__ double buf[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) buf[i] = (double)((i * 7 + 1) & 1023);
  __syncthreads();
  int idx = threadIdx.x & 1023;
  T0();
  for (int i = 0; i < iters; i++) idx = (int)buf[idx];
  T1(3);
  out[threadIdx.x] = idx;
}
__global__ void barrier_loop(double *out, unsigned long long *st, int iters) {
  T0();
  for (int i = 0; i < iters; i++) __syncthreads();
  T1(4);
  out[threadIdx.x] = 0;
}
__global__ void dep_max(double *out, unsigned long long *st, int iters) {
  double x = threadIdx.x, y = 3.0;
  T0();
  for (int i = 0; i < iters; i++) { x = fmax(x, y); y = fmax(y, x + 1e-3); }
  T1(5);
  out[threadIdx.x] = x + y;
}
int main(int argc, char **argv) {
  int iters = 20000;
  double *out;
  unsigned long long *st, h[8];
  hipMalloc(&out, sizeof(double) * 1024);
  hipMalloc(&st, sizeof(h));
  hipMemset(st, 0, sizeof(h));
  for (int r = 0; r < 2; r++) {
    dep_fma<<<1, 256>>>(out, st, iters);
    indep_fma<<<1, 256>>>(out, st, iters);
    dep_div<<<1, 256>>>(out, st, iters);
    lds_chain<<<1, 256>>>(out, st, iters);
    barrier_loop<<<1, 256>>>(out, st, iters);
    dep_max<<<1, 256>>>(out, st, iters);
    hipDeviceSynchronize();
  }
  hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
  printf("dependent fma: %.1f cyc | 8 independent fma: %.1f cyc per fma | 1/(x+.5): %.1f cyc | LDS f64 load->cvt->load chain: %.1f cyc | __syncthreads (4 waves): %.1f cyc | fmax+add chain per iter: %.1f\n",
         (double)h[0] / iters, (double)h[1] / iters / 8, (double)h[2] / iters, (double)h[3] / iters, (double)h[4] / iters, (double)h[5] / iters);
  return 0;
}
