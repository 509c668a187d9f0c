// How many waves per SIMD does it take to keep the fp64 matrix pipe of gfx950 busy?  Each wave issues
// `iters` x 16 independent v_mfma_f64_16x16x4_f64 (register operands, 16 accumulators); grids of 1..4
// 256-thread workgroups per CU (= 1..4 waves per SIMD), then the same with 512-thread workgroups.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_build/mfma_f64_probe tools/mfma_f64_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void __launch_bounds__(256) k(int iters, double *out, unsigned long long *clk) {
  double4_t acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = (double4_t){0, 0, 0, 0};
  double a = threadIdx.x * 1e-3 + 1.0, b = 1.0 - threadIdx.x * 1e-4;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
template <int NACC>
void run(int wgs_per_cu, int iters) {
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int grid = cus * wgs_per_cu;
  double *out;
  unsigned long long *clk;
  hipMalloc(&out, sizeof(double) * grid * 256);
  hipMalloc(&clk, sizeof(unsigned long long) * grid);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  k<NACC><<<grid, 256>>>(iters, out, clk);
  hipEventRecord(e0);
  k<NACC><<<grid, 256>>>(iters, out, clk);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c0;
  hipMemcpy(&c0, clk, 8, hipMemcpyDeviceToHost);
  const double flops = 2048.0 * NACC * iters * 4.0 * grid;
  printf("acc %2d, %d waves/SIMD: %.3f ms  %.1f TFLOP/s  (%.1f %% of 78.6); wave 0: %.1f shader cycles per MFMA\n", NACC, wgs_per_cu, ms,
         flops / ms / 1e9, flops / ms / 1e9 / 78.6 * 100, (double)c0 / ((double)NACC * iters));
  hipFree(out), hipFree(clk);
}
int main() {
  for (int w = 1; w <= 4; w++) run<16>(w, 4000);
  for (int w = 1; w <= 4; w++) run<4>(w, 16000);
  for (int w = 1; w <= 2; w++) run<1>(w, 64000);
  return 0;
}
