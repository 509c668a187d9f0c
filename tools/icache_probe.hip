// Diagnostic: cost of executing straight-line code ONCE (cold instruction cache) on gfx950.
// One wavefront runs N independent 8-byte VALU instructions (v_add_u32 with a literal, eight
// registers in rotation: issue-limited at 4 cycles each when the code is resident), first in a
// fresh kernel launch (cold), then the same code a second time inside the same launch (warm).
//   hipcc --offload-arch=gfx950 -O2 -o tools/icache_probe tools/icache_probe.hip && tools/icache_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define I8(r) "v_add_u32 v" #r ", 0x12345678, v" #r "\n"
#define B8 I8(10) I8(11) I8(12) I8(13) I8(14) I8(15) I8(16) I8(17)
#define B64 B8 B8 B8 B8 B8 B8 B8 B8
#define B512 B64 B64 B64 B64 B64 B64 B64 B64
template <int KB>  // KB kilobytes of code = KB * 128 instructions
__global__ void k_code(unsigned long long *st, int *sink, int reps) {
  int acc = threadIdx.x;
  for (int r = 0; r < reps; r++) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if constexpr (KB >= 4) asm volatile(B512 ::: "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17");
    if constexpr (KB >= 8) asm volatile(B512 ::: "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17");
    if constexpr (KB >= 16) asm volatile(B512 B512 ::: "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17");
    if constexpr (KB >= 32) asm volatile(B512 B512 B512 B512 ::: "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) st[r] = t1 - t0;
    acc += r;
  }
  sink[threadIdx.x] = acc;
}
template <int KB>
static void run(const char *name, int blocks) {
  unsigned long long *st, h[4];
  int *sink;
  hipMalloc(&st, 64), hipMalloc(&sink, 4 * 64);
  for (int trial = 0; trial < 2; trial++) {
    hipMemset(st, 0, 64);
    k_code<KB><<<blocks, 64>>>(st, sink, 3);
    hipMemcpy(h, st, 32, hipMemcpyDeviceToHost);
    printf("%s, %d wavefront(s), launch %d: %d instructions: first pass %llu cycles (%.1f / instr), second %llu (%.1f), third %llu\n",
           name, blocks, trial, KB * 128, h[0], (double)h[0] / (KB * 128), h[1], (double)h[1] / (KB * 128), h[2]);
  }
  hipFree(st), hipFree(sink);
}
int main() {
  run<4>("4 KB", 1);
  run<8>("8 KB", 1);
  run<16>("16 KB", 1);
  run<32>("32 KB", 1);
  run<32>("32 KB", 256);
  return 0;
}
