#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ void __launch_bounds__(256) k_stride(double *p, long long n) {
  const long long n2 = n >> 1, stride = (long long)gridDim.x * blockDim.x;
  d2 *q = (d2 *)p;
  const d2 z{0.0, 0.0};
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n2; i += 4 * stride) {
    if (NT) {
      __builtin_nontemporal_store(z, q + i), __builtin_nontemporal_store(z, q + i + stride);
      __builtin_nontemporal_store(z, q + i + 2 * stride), __builtin_nontemporal_store(z, q + i + 3 * stride);
    } else
      q[i] = z, q[i + stride] = z, q[i + 2 * stride] = z, q[i + 3 * stride] = z;
  }
  for (; i < n2; i += stride) q[i] = z;
}
// each workgroup a contiguous chunk
template <bool NT, int TPB>
__global__ void __launch_bounds__(TPB) k_chunk(double *p, long long n) {
  const long long n2 = n >> 1;
  const long long per = (n2 + gridDim.x - 1) / gridDim.x;
  const long long b0 = (long long)blockIdx.x * per, b1 = b0 + per < n2 ? b0 + per : n2;
  d2 *q = (d2 *)p;
  const d2 z{0.0, 0.0};
  long long i = b0 + threadIdx.x;
  for (; i + 3 * TPB < b1; i += 4 * TPB) {
    if (NT) {
      __builtin_nontemporal_store(z, q + i), __builtin_nontemporal_store(z, q + i + TPB);
      __builtin_nontemporal_store(z, q + i + 2 * TPB), __builtin_nontemporal_store(z, q + i + 3 * TPB);
    } else
      q[i] = z, q[i + TPB] = z, q[i + 2 * TPB] = z, q[i + 3 * TPB] = z;
  }
  for (; i < b1; i += TPB) q[i] = z;
}
template <class F>
static void run(const char *name, F f, double bytes) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int w = 0; w < 2; w++) {
    hipEventRecord(e0);
    for (int r = 0; r < 20; r++) f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-40s %.1f us  %.0f GB/s\n", name, ms * 1e3 / 20, bytes / (ms * 1e-3 / 20) / 1e9);
}
int main() {
  const long long n = 26500000;  // C2's panel arena, doubles
  double *p;
  hipMalloc(&p, sizeof(double) * n);
  const double bytes = 8.0 * n;
  run("hipMemsetAsync", [&]() { hipMemsetAsync(p, 0, sizeof(double) * n, 0); }, bytes);
  for (int g : {1024, 2048, 4096, 8192, 16384}) {
    char nm[64];
    snprintf(nm, 64, "stride grid %d", g);
    run(nm, [&]() { k_stride<false><<<g, 256>>>(p, n); }, bytes);
    snprintf(nm, 64, "stride nt grid %d", g);
    run(nm, [&]() { k_stride<true><<<g, 256>>>(p, n); }, bytes);
    snprintf(nm, 64, "chunk 256 grid %d", g);
    run(nm, [&]() { k_chunk<false, 256><<<g, 256>>>(p, n); }, bytes);
    snprintf(nm, 64, "chunk nt 256 grid %d", g);
    run(nm, [&]() { k_chunk<true, 256><<<g, 256>>>(p, n); }, bytes);
    snprintf(nm, 64, "chunk 1024 grid %d", g / 4);
    run(nm, [&]() { k_chunk<false, 1024><<<g / 4, 1024>>>(p, n); }, bytes);
  }
  return 0;
}
