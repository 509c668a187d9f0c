// What a device-wide barrier inside one persistent kernel costs next to a kernel boundary inside a captured
// graph (the shallow top of the tree in one launch).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_persistent(unsigned *ctr, double *data, int rounds, int *timeout) {
  const unsigned G = gridDim.x;
  for (int r = 0; r < rounds; r++) {
    // a little work that the next round of ANOTHER workgroup reads: one value per workgroup
    if (threadIdx.x == 0) data[(blockIdx.x + r) % G] += 1.0;
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      atomicAdd(ctr, 1u);
      long spin = 0;
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < G * (unsigned)(r + 1)) {
        if (++spin > 20000000L) { *timeout = 1; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
}
__global__ void k_step(double *data, int r, int G) {
  if (threadIdx.x == 0) data[(blockIdx.x + r) % G] += 1.0;
}

int main() {
  const int rounds = 200;
  unsigned *ctr; double *data; int *to;
  CHK(hipMalloc(&ctr, 4)); CHK(hipMalloc(&data, 8 * 1024)); CHK(hipMalloc(&to, 4));
  hipStream_t s; CHK(hipStreamCreate(&s));
  hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  for (int G : {8, 64, 256}) {
    CHK(hipMemsetAsync(to, 0, 4, s));
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
      CHK(hipMemsetAsync(ctr, 0, 4, s));
      CHK(hipEventRecord(a, s));
      k_persistent<<<G, 256, 0, s>>>(ctr, data, rounds, to);
      CHK(hipEventRecord(b, s));
      CHK(hipStreamSynchronize(s));
      float ms; CHK(hipEventElapsedTime(&ms, a, b));
      best = ms < best ? ms : best;
    }
    int hto = 0; CHK(hipMemcpy(&hto, to, 4, hipMemcpyDeviceToHost));
    // the same number of dependent steps as kernels of a captured graph
    hipGraph_t g; hipGraphExec_t ge;
    CHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int r = 0; r < rounds; r++) k_step<<<G, 256, 0, s>>>(data, r, G);
    CHK(hipStreamEndCapture(s, &g));
    CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    float bestg = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
      CHK(hipEventRecord(a, s));
      CHK(hipGraphLaunch(ge, s));
      CHK(hipEventRecord(b, s));
      CHK(hipStreamSynchronize(s));
      float ms; CHK(hipEventElapsedTime(&ms, a, b));
      bestg = ms < bestg ? ms : bestg;
    }
    printf("workgroups %3d: device-wide barrier %.2f us per round%s, kernel boundary in a graph %.2f us per step\n", G,
           1e3 * best / rounds, hto ? " (TIMED OUT)" : "", 1e3 * bestg / rounds);
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
  }
  return 0;
}
