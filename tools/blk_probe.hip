// Micro-probes of the pieces of k_factor_blk on gfx950 (cycles from s_memtime): build on the GPU box with
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/_build/blk_probe tools/blk_probe.hip && tools/_build/blk_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../hqp_amd/csrc/kernels.hip.h"
#include "../hqp_amd/csrc/factor_blk.hip.h"
using namespace kktdev;

#define T0() unsigned long long t0 = __builtin_amdgcn_s_memtime()
#define T1(slot) if (threadIdx.x == tmark) st[slot] = __builtin_amdgcn_s_memtime() - t0

// the elimination of a diagonal block by the last wavefront of the workgroup; the others wait at the barrier
__global__ void ge_probe(const double *G, double *out, unsigned long long *st, int iters, int slot) {
  extern __shared__ double lds[];
  double *Gb = lds, *Tb = Gb + 272, *Ldg = Tb + 272, *Xq = Ldg + 272, *dv = Xq + 16 * 208, *di = dv + 16;
  int *bad = (int *)(di + 16);
  const int tmark = blockDim.x - 64;
  for (int i = threadIdx.x; i < 256; i += blockDim.x) Gb[(i >> 4) * 17 + (i & 15)] = G[i];
  __syncthreads();
  const bool ge = threadIdx.x >= blockDim.x - 64;
  T0();
  for (int it = 0; it < iters; it++) {
    if (ge) fb_eliminate_block<208>(Gb, Tb, Xq + 32, Xq, dv, di, bad, 0.64, 1e-300, 0, threadIdx.x & 63);
    fb_barrier();
  }
  T1(slot);
  out[threadIdx.x] = Tb[threadIdx.x & 255];
}

// 15 independent v_fmac_f64 with the DPP row broadcast / plain, per trip
__global__ void fmac_probe(double *out, unsigned long long *st, int iters) {
  const int tmark = 0;
  double g[16], nl = 1e-9 * threadIdx.x;
  for (int c = 0; c < 16; c++) g[c] = c + threadIdx.x;
  {
    T0();
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int c = 0; c < 15; c++) FB_FMAC(c, 3, nl);
    }
    T1(2);
  }
  {
    T0();
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int c = 0; c < 15; c++) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(g[c]) : "v"(g[15]), "v"(nl));
    }
    T1(3);
  }
  {
    T0();
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int c = 0; c < 15; c++) asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(g[0]) : "v"(nl));
    }
    T1(4);  // dependent chain of the DPP form
  }
  {
    T0();
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int c = 0; c < 15; c++) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(g[0]) : "v"(g[15]), "v"(nl));
    }
    T1(5);  // dependent chain, plain
  }
  {
    double d = g[1];
    T0();
    for (int it = 0; it < iters; it++) {
      double x = __builtin_amdgcn_rcp(d);
      const double e = fma(-d, x, 1.0);
      const double e2 = fma(e, e, e);
      d = fma(x, e2, x) + 1.5;
    }
    T1(6);  // the reciprocal chain of a step
    g[2] = d;
  }
  double s = 0;
  for (int c = 0; c < 16; c++) s += g[c];
  out[threadIdx.x] = s;
}

// per wavefront `nslot` block updates per trip: 8 LDS operand reads, 4 dependent f64 MFMAs
template <int NSLOT>
__global__ void upd_probe(double *out, unsigned long long *st, int iters, int slot) {
  extern __shared__ double lds[];
  constexpr int ld = 208;
  double *Op = lds, *Lb = lds + 16 * ld;
  const int tmark = 0;
  for (int i = threadIdx.x; i < 32 * ld; i += blockDim.x) lds[i] = 1e-3 * (i % 97);
  __syncthreads();
  const int lane = threadIdx.x & 63, ln = lane & 15, lg = lane >> 4, wave = threadIdx.x >> 6;
  double4_t R[NSLOT];
  for (int s = 0; s < NSLOT; s++) R[s] = double4_t{0, 0, 0, 0};
  T0();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int s = 0; s < NSLOT; s++) {
      const int J = (s + wave + it) % 12, I = (s * 5 + wave) % 12;
      const double *ab = Op + (lg * ld + 16 * J + ln), *lb = Lb + (lg * ld + 16 * I + ln);
      double a[4], l[4];
#pragma unroll
      for (int q = 0; q < 4; q++) a[q] = ab[4 * q * ld], l[q] = lb[4 * q * ld];
      double4_t acc = R[s];
#pragma unroll
      for (int q = 0; q < 4; q++) acc = mfma_f64(a[q], l[q], acc);
      R[s] = acc;
    }
  }
  T1(slot);
  double s = 0;
  for (int k = 0; k < NSLOT; k++) s += R[k][0] + R[k][3];
  out[threadIdx.x] = s;
}

// dependent / independent f64 MFMAs
__global__ void mfma_probe(double *out, unsigned long long *st, int iters) {
  const int tmark = 0;
  double a = 1e-3 * threadIdx.x, b = 1.0 + 1e-6 * threadIdx.x;
  double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  {
    T0();
    for (int it = 0; it < iters; it++) {
      c0 = mfma_f64(a, b, c0), c0 = mfma_f64(a, b, c0), c0 = mfma_f64(a, b, c0), c0 = mfma_f64(a, b, c0);
    }
    T1(10);
  }
  {
    T0();
    for (int it = 0; it < iters; it++) {
      c0 = mfma_f64(a, b, c0), c1 = mfma_f64(a, b, c1), c2 = mfma_f64(a, b, c2), c3 = mfma_f64(a, b, c3);
    }
    T1(11);
  }
  out[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

// a long straight-line sequence (about 24 KB of code: 3000 VALU + SALU instructions) run repeatedly: what a
// wavefront pays for code that is not a short loop, alone and with other wavefronts doing the same
#define SL4(x) x x x x
#define SL16(x) SL4(SL4(x))
#define SL256(x) SL16(SL16(x))
__global__ void straight_probe(double *out, unsigned long long *st, int iters, int slot) {
  const int tmark = 0;
  int v = threadIdx.x, u = blockIdx.x;
  T0();
  for (int it = 0; it < iters; it++) {
    SL256(asm volatile("v_add_u32 %0, %0, %1\n\ts_add_u32 %1, %1, 1\n\tv_xor_b32 %0, %0, %1\n\ts_and_b32 %1, %1, 1023\n\tv_add_u32 %0, 3, %0\n\ts_add_u32 %1, %1, 7\n\tv_lshlrev_b32 %0, 1, %0\n\ts_xor_b32 %1, %1, 5\n\tv_add_u32 %0, 1, %0\n\ts_add_u32 %1, %1, 3\n\tv_xor_b32 %0, 9, %0\n\ts_add_u32 %1, %1, 1" : "+v"(v), "+s"(u) : : "scc");)
  }
  T1(slot);
  out[threadIdx.x] = v + u;
}

__global__ void barrier_probe(double *out, unsigned long long *st, int iters, int slot) {
  const int tmark = 0;
  T0();
  for (int it = 0; it < iters; it++) fb_barrier();
  T1(slot);
  out[threadIdx.x] = 0;
}

int main() {
  const int iters = 2000;
  double *out, *G;
  unsigned long long *st;
  hipMalloc(&out, sizeof(double) * 2048);
  hipMalloc(&G, sizeof(double) * 256);
  hipMalloc(&st, sizeof(unsigned long long) * 32);
  hipMemset(st, 0, sizeof(unsigned long long) * 32);
  std::vector<double> g(256);
  for (int i = 0; i < 16; i++)
    for (int j = 0; j < 16; j++) g[i * 16 + j] = i == j ? -20.0 - i : 0.3 * ((i * 7 + j * 3) % 5 - 2);
  for (int i = 0; i < 16; i++)
    for (int j = 0; j < i; j++) g[i * 16 + j] = g[j * 16 + i];
  hipMemcpy(G, g.data(), sizeof(double) * 256, hipMemcpyHostToDevice);
  const size_t lds = sizeof(double) * (3 * 272 + 16 * 208 + 64);
  ge_probe<<<1, 64, lds>>>(G, out, st, iters, 0);
  ge_probe<<<1, 1024, lds>>>(G, out, st, iters, 1);
  fmac_probe<<<1, 64>>>(out, st, iters);
  hipFuncSetAttribute((const void *)upd_probe<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * 208 * 8);
  hipFuncSetAttribute((const void *)upd_probe<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * 208 * 8);
  upd_probe<4><<<1, 64, 32 * 208 * 8>>>(out, st, iters, 12);
  upd_probe<4><<<1, 256, 32 * 208 * 8>>>(out, st, iters, 13);
  upd_probe<4><<<1, 1024, 32 * 208 * 8>>>(out, st, iters, 14);
  upd_probe<8><<<1, 512, 32 * 208 * 8>>>(out, st, iters, 15);
  mfma_probe<<<1, 64>>>(out, st, iters);
  straight_probe<<<1, 64>>>(out, st, 200, 20);
  straight_probe<<<1, 256>>>(out, st, 200, 21);
  straight_probe<<<1, 1024>>>(out, st, 200, 22);
  barrier_probe<<<1, 512>>>(out, st, iters, 16);
  barrier_probe<<<1, 1024>>>(out, st, iters, 17);
  hipDeviceSynchronize();
  unsigned long long h[32];
  hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
  auto per = [&](int k) { return (double)h[k] / iters; };
  printf("elimination of a 16x16 block, one wavefront alone: %.0f cycles; last of 16 wavefronts (others at the barrier): %.0f\n", per(0), per(1));
  printf("15 fp64 multiply-adds: DPP row broadcast %.0f, plain %.0f; dependent chain of 15: DPP %.0f, plain %.0f; reciprocal chain %.0f\n", per(2), per(3), per(4), per(5), per(6));
  printf("4 f64 MFMA 16x16x4: dependent %.0f, independent %.0f\n", per(10), per(11));
  printf("block update (8 LDS reads + 4 MFMA), 4 per wavefront and trip: 1 wavefront %.0f, 4 wavefronts %.0f, 16 wavefronts %.0f; 8 per wavefront, 8 wavefronts %.0f\n", per(12), per(13), per(14), per(15));
  printf("3072 straight-line instructions (6 VALU + 6 SALU per group): 1 wavefront %.0f cycles per pass, 4 wavefronts %.0f, 16 wavefronts %.0f\n", (double)h[20] / 200, (double)h[21] / 200, (double)h[22] / 200);
  printf("barrier: 8 wavefronts %.0f, 16 wavefronts %.0f\n", per(16), per(17));
  return 0;
}
