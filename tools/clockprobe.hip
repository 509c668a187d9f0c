// Diagnostic: in-kernel shader clock = d(s_memtime) / d(s_memrealtime) * 100 MHz
// (MI355X_MICROARCH.md, DVFS item 6), for a lightly loaded chip (few workgroups).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void spin(double *out, unsigned long long *stamps, int iters) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  double x = threadIdx.x * 1e-9, y = 1.0000001;
  for (int i = 0; i < iters; i++) x = fma(x, y, 1e-9);
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = t1 - t0;
    stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
}
int main(int argc, char **argv) {
  int nblk = argc > 1 ? atoi(argv[1]) : 1, iters = argc > 2 ? atoi(argv[2]) : 200000, reps = argc > 3 ? atoi(argv[3]) : 5;
  double *out;
  unsigned long long *st, h[2];
  hipMalloc(&out, sizeof(double) * nblk * 256);
  hipMalloc(&st, sizeof(unsigned long long) * 2 * nblk);
  for (int r = 0; r < reps; r++) {
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    hipEventRecord(a);
    spin<<<nblk, 256>>>(out, st, iters);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, a, b);
    hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
    printf("blocks %d iters %d: %.3f ms, memtime %llu realtime %llu -> clock %.0f MHz, cycles/fma %.2f\n", nblk, iters, ms,
           h[0], h[1], 100.0 * h[0] / h[1], (double)h[0] / iters);
  }
  return 0;
}
