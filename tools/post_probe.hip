// Read-back of a few words from a running stream: (a) hipMemcpyAsync to pinned memory + hipStreamSynchronize against
// (b) a one-workgroup kernel that stores the words into mapped, coherent host memory and a sequence word behind a
// system-scope fence, the host spinning on that word.  Each round: a dependent chain of `chain` small kernels, then
// the read-back; printed: microseconds per round, over 2000 rounds.
//   hipcc --offload-arch=gfx950 -O2 -o tools/post_probe tools/post_probe.hip && tools/post_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_work(double *p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0000001 + 1e-9;
}
__global__ void k_post(const int *__restrict__ src, int nwords, int *__restrict__ dst, unsigned *__restrict__ seqw, unsigned seq) {
  for (int i = threadIdx.x; i < nwords; i += blockDim.x) __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence_system();
    __hip_atomic_store(seqw, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
int main() {
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  double *d;
  int *dflags, *hpin, *hmap, *hmap_dev;
  const int n = 8000, nw = 128;
  CK(hipMalloc(&d, sizeof(double) * n));
  CK(hipMemset(d, 0, sizeof(double) * n));
  CK(hipMalloc(&dflags, sizeof(int) * nw));
  CK(hipMemset(dflags, 1, sizeof(int) * nw));
  CK(hipHostMalloc((void **)&hpin, sizeof(int) * nw, hipHostMallocDefault));
  CK(hipHostMalloc((void **)&hmap, sizeof(int) * (nw + 64), hipHostMallocMapped | hipHostMallocCoherent));
  CK(hipHostGetDevicePointer((void **)&hmap_dev, hmap, 0));
  volatile unsigned *seqw = (volatile unsigned *)(hmap + nw + 16);
  *seqw = 0;
  for (int chain : {1, 8}) {
    for (int mode = 0; mode < 2; mode++) {
      const int rounds = 2000;
      CK(hipStreamSynchronize(s));
      const auto t0 = std::chrono::steady_clock::now();
      unsigned seq = *seqw;
      for (int r = 0; r < rounds; r++) {
        for (int c = 0; c < chain; c++) k_work<<<(n + 255) / 256, 256, 0, s>>>(d, n);
        if (mode == 0) {
          CK(hipMemcpyAsync(hpin, dflags, sizeof(int) * nw, hipMemcpyDeviceToHost, s));
          CK(hipStreamSynchronize(s));
        } else {
          seq++;
          k_post<<<1, 128, 0, s>>>(dflags, nw, hmap_dev, (unsigned *)(hmap_dev + nw + 16), seq);
          while (*seqw != seq) {
          }
        }
      }
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / rounds;
      CK(hipStreamSynchronize(s));
      printf("chain of %d kernels + %s: %.2f us per round\n", chain, mode == 0 ? "hipMemcpyAsync + hipStreamSynchronize" : "posting kernel + host spin          ", us);
    }
  }
  return 0;
}
