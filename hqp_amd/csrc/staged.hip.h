// gfx950 kernels of the STAGED engine (HQPKKT_MODE_STAGED): the stage-structured
// solution of the interior-point Newton system for multistage (DOCP) problems, i.e. the
// job of the reference's Hqp_IpLQDOCP (hqp/Hqp_IpLQDOCP.C:796-976; extended Riccati
// recursion ExRiccatiFactorSc :1794-1999, ExRiccatiSolveSc :2007-2182).  Included once by
// hqpkkt.hip.  The algorithm and its notation: tests/model_staged.py (numpy model).
//
// Layout in HBM: every dense block row-major with an even leading dimension (16-byte
// loads), so that EVERY matrix product of the recursion is of the one form
//     C (M x N) = alpha * sum_k A[k][i] * B[k][j] + beta * Cin        ("TN": both operands k-major)
//   W   = V+ F          A = V+ (symmetric), B = F = [fx fu] (n+ x (n+m))
//   G   = F' W          lower tiles only
//   Nc  = B+ F          A = BT+ (n+ x cap: the carried constraint rows, transposed)
//   Rm  = K^-1 Y        A = K^-1 (symmetric)
//   V   = Gxx - Y' Rm   lower tiles + mirror
// v_mfma_f64_16x16x4 operand layout (hqpkkt_selftest_mfma): A: lane l holds A[l&15][l>>4];
// B: B[l>>4][l&15]; C/D: col = l&15, row = (l>>4) + 4*reg.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>

#include "staged_plan.hpp"
#include "sk_table.hpp"

namespace stg {

using kktdev::double4_t;
using kktdev::mfma_f64;

typedef double double2_t __attribute__((ext_vector_type(2)));

// One system over several ranks (staged_plan.hpp): rank p owns the state columns [cut[p], cut[p+1]) of a stage
// (multiples of 128, so a tile lies inside one strip).  StripTab: where the ranks' local blocks of F lie once they are
// gathered - strip p = n+ rows of cut[p+1] - cut[p] columns (and the control columns behind them), row-major with
// leading dimension ld[p], at fg + off[p].
struct StripTab {
  int nranks;
  int cut[17];
  int ld[17];  // leading dimension of strip p (its width + the control columns, a multiple of 8)
  long long off[17];
};
// ... and where the blocks of G_xx lie after the second: block (a, b), a >= b, = rows of strip a x columns of strip b in
// one part, or in two cut at row rsplit (the pairs half the ring apart, staged_plan.cpp); part t row-major with leading
// dimension cut[b+1] - cut[b] at x + off[t]
struct RectTab {
  int nranks;
  int cut[17];
  struct Block {
    long long off[2];
    int rsplit;  // first row of the second part (a large number: one part)
    int pad;
  } blk[16 * 16];  // [a * 16 + b]
};
static __device__ __forceinline__ int strip_of(const int *cut, int nranks, int j) {
  int p = 0;
  while (p + 1 < nranks && cut[p + 1] <= j) p++;
  return p;
}

struct GemmArgs {
  const double *A;
  long long lda;  // K x M, row-major (k-major)
  const double *B;
  long long ldb;  // K x N
  const double *Cin;
  long long ldcin;  // M x N, read when beta != 0
  double *C;
  long long ldc;
  int M, N, K;
  double alpha, beta;
  int lower;   // only tiles with tile row >= tile column (M == N)
  int mirror;  // with lower: C[j][i] = C[i][j] as well (exactly symmetric result)
  const int *tile_map;  // 128 x 128 tiles: tile index -> tile row << 16 | tile column.  Lower: in blocks of
                        // 8 x 8 tiles (neighbours in the launch order share operand panels in their XCD's L2);
                        // not lower: the tiles of a launch that computes a part of the product only (the blocks of G_xx
                        // one rank owns, bstrips); null: row by row
  const double *zeros;  // >= 128 zero doubles (16-byte aligned): the source of the operand rows k >= K when the
                        // 128 x 128 kernels stage their operands by LDS-DMA; null: staging through registers
  const RectTab *rects;        // with beta != 0: Cin(i, j), i >= j, is read from the blocks in the exchange buffer `Cin`
                               // (the rank-q update of a sharded stage takes G_xx straight from what the ranks sent)
  const StripTab *bstrips;     // the columns of B come from the ranks' strips in the exchange buffer `B` (128-wide tiles)
  unsigned long long *stamps;  // diagnostic builds of the plain kernel only (hqpkkt_debug_dgemm): 4 constant-clock
                               // (100 MHz) time stamps per workgroup: start, operands of the first slab in LDS, end of
                               // the k loop, end of the epilogue; null in every product of the engine
};

static const int GEMM_BK = 16;

// number of b x b tiles of an M x N product; lower: only tiles with tile row >= tile column (M >= N: a triangle of
// ceil(N / b) tile columns on top of a rectangle - the column strip of a lower triangle that one rank computes)
static inline long long gemm_tiles(int M, int N, int b, int lower) {
  const long long tm = (M + b - 1) / b, tn = (N + b - 1) / b;
  return lower ? tn * (tn + 1) / 2 + (tm > tn ? (tm - tn) * tn : 0) : tm * tn;
}
static inline size_t gemm_lds_bytes(int bm, int bn, int nbuf = 2) { return sizeof(double) * nbuf * GEMM_BK * (size_t)(bm + 16 + bn + 16); }

// 128 x 128 tiles from 384 tiles on (the grid of 2 x 256 workgroups three quarters full).  Below that the 64 x 64
// kernel with four times the tiles is faster (same-box comparisons of round 3: 2000 x 2050 x 2000
// 0.41 against 0.50 ms, 1500 x 1540 x 1500 0.19 against 0.21) with one exception: a deep rectangular product of 160 -
// 256 tiles - the column strip of W when a C4 system is sharded over 8 ranks, 5000 x 640 x 5000 - as ONE round of one
// workgroup per CU (gemm_launch_plain): 0.71 against 0.83 ms.
static inline bool gemm_big_tiles(int M, int N, int lower, int K = 0) {
  const long long t = gemm_tiles(M, N, 128, lower);
  return t >= 384 || (!lower && t >= 160 && t <= 256 && K >= 256 * GEMM_BK);
}

// The split form (k_dgemm_tn_sk, below) pays where whole rounds of 128 x 128 tiles would leave slots idle
// and the product is deep enough to be cut.  Returns true when the launch should use it with the whole
// `grid` (two workgroups per CU): more than one tile, not a multiple of the grid, and either more tiles
// than half the grid or a plan that puts at least a quarter of the grid to work (below that the 64 x 64
// tiles fill the chip better).
static inline bool gemm_use_split(int M, int N, int K, int lower, int grid);
// blockIdx -> position in a sequence in which the workgroups of one XCD (blockIdx % 8) are
// neighbours (each XCD has its own L2; neighbouring tiles share operand panels)
__device__ __forceinline__ int xcd_swizzle(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return x * q + (x < r ? x : r) + (bid >> 3);
}

// One workgroup (256 threads, 2 x 2 wavefronts) per BM x BN tile of C; k-slabs of 16 rows
// of both operands go global -> registers -> LDS (two buffers: the loads of slab t+1 are in
// flight while slab t is multiplied), fragments LDS -> registers with ds_read_b64, conflict
// free because an LDS row is BM + 16 doubles (rows k, k+1 of a fragment: banks 32 apart).
// WGM x WGN wavefronts per workgroup (2 x 2: 64 x 64 per wave, 16 accumulator tiles = 128 registers, two waves
// per SIMD; 2 x 4: 64 x 32 per wave, 8 accumulator tiles, under 128 registers, FOUR waves per SIMD with two
// workgroups per CU - one wave issues a v_mfma_f64_16x16x4 only every ~140 cycles (stamps of the 2 x 2 kernel:
// every workgroup proceeds at that pace whoever its partner is, profiles/r03_dgemm_stamps.txt), the pipe takes
// one per 64, so two waves per SIMD top out near 90 % of the peak and it takes three or four to fill it)
template <int BM, int BN, int WGM = 2, int WGN = 2>
struct GemmTile {
  static constexpr int BK = GEMM_BK;
  static constexpr int NW = WGM * WGN, NT = 64 * NW;
  static constexpr int LDA = BM + 16, LDB = BN + 16;
  static constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 16, TN = WN / 16;
  static constexpr int LA = BK * BM / 2 / 256, LB = BK * BN / 2 / 256;  // 16-byte loads per thread and slab
  static constexpr int RA = 256 / (BM / 2), RB = 256 / (BN / 2);        // slab rows covered by one pass

  // tile index -> (tile row, tile column)
  static __device__ __forceinline__ void tile_of(const GemmArgs &g, int t, int &tm, int &tn) {
    if (g.tile_map && BM == 128) {
      const int e = g.tile_map[t];
      tm = e >> 16, tn = e & 0xffff;
    } else if (g.lower) {
      const int tcols = (g.N + BN - 1) / BN, tri = tcols * (tcols + 1) / 2;
      if (t < tri) {
        tm = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
        while ((tm + 1) * (tm + 2) / 2 <= t) tm++;
        while (tm * (tm + 1) / 2 > t) tm--;
        tn = t - tm * (tm + 1) / 2;
      } else {  // (M > N: the rectangle below the triangle, row by row)
        tm = tcols + (t - tri) / tcols;
        tn = (t - tri) % tcols;
      }
    } else {
      const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
      constexpr int GM = 8;  // tile rows walked together: their A panels stay in L2
      const int grp = t / (GM * tiles_n), first = grp * GM;
      const int rows = min(GM, tiles_m - first);
      const int in = t - grp * GM * tiles_n;
      tm = first + in % rows;
      tn = in / rows;
    }
  }

  // acc += sum over the slabs [s0, s1) of the tile at (i0, j0); ends with a barrier (LDS free again)
  static __device__ __forceinline__ void accumulate(const GemmArgs &g, int i0, int j0, int s0, int s1,
                                                    double4_t (&acc)[TM][TN], double *As, double *Bs) {
    static_assert(NT == 256, "the register-staged loop is written for 256 threads");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int lr = lane & 15, lk = lane >> 4;
    // global -> register staging: thread covers columns ca, ca+1 of rows ra + p*RA
    const int ca = 2 * (tid % (BM / 2)), ra = tid / (BM / 2);
    const int cb = 2 * (tid % (BN / 2)), rb = tid / (BN / 2);
    // a 16-byte load is inside its row when its first column is < ld (ld even)
    const long long acol = (i0 + ca < g.lda) ? i0 + ca : 0;
    const double *Bp = g.B;
    long long ldb = g.ldb;
    int jb = j0;  // first column of the tile inside its B block
    if (g.bstrips) {
      const int q = strip_of(g.bstrips->cut, g.bstrips->nranks, j0);
      Bp = g.B + g.bstrips->off[q], ldb = g.bstrips->ld[q], jb = j0 - g.bstrips->cut[q];
    }
    const long long bcol = (jb + cb < ldb) ? jb + cb : 0;
    // D register sets: the loads of slab t + D are issued before the multiplications of slab t and consumed (masked
    // for k >= K, stored to LDS) after those of slab t + D - 1.  With 64 x 64 tiles a slab is 16 multiplications per
    // wavefront (0.4 us) and a load from L2 takes 1.4: one set (round 2) left the loop waiting for its loads - 1.44 us
    // per slab (profiles: 272 tiles of a 1000-state stage in 91 us); the 128 x 128 form of this loop keeps one set
    // (its slab is four times the work, its sets four times the registers).
    constexpr int D = (BM * BN <= 64 * 64) ? 4 : 1;
    double2_t sa[D][LA], sb[D][LB];
    // (no branch anywhere in the loop: rows k >= K are read from row K - 1 and zeroed on their way to LDS, slabs
    // behind the last one are the last one again and go to a buffer nobody reads - with branches between the loads and
    // their use the compiler waits for ALL loads in flight at every join, also those just issued)
    const int last = s1 - 1;
    auto gload = [&](double2_t(&xa)[LA], double2_t(&xb)[LB], int slab) {
      const int k0 = (slab < last ? slab : last) * BK;
#pragma unroll
      for (int p = 0; p < LA; p++) {
        const int k = k0 + ra + p * RA, kc = k < g.K ? k : g.K - 1;
        xa[p] = *(const double2_t *)(g.A + (long long)kc * g.lda + acol);
      }
#pragma unroll
      for (int p = 0; p < LB; p++) {
        const int k = k0 + rb + p * RB, kc = k < g.K ? k : g.K - 1;
        xb[p] = *(const double2_t *)(Bp + (long long)kc * ldb + bcol);
      }
    };
    auto lstore = [&](int buf, const double2_t(&xa)[LA], const double2_t(&xb)[LB], int slab) {
      const int k0 = (slab < last ? slab : last) * BK;
#pragma unroll
      for (int p = 0; p < LA; p++) {
        double2_t v = xa[p];
        if (k0 + ra + p * RA >= g.K) v = (double2_t){0.0, 0.0};
        *(double2_t *)(As + (buf * BK + ra + p * RA) * LDA + ca) = v;
      }
#pragma unroll
      for (int p = 0; p < LB; p++) {
        double2_t v = xb[p];
        if (k0 + rb + p * RB >= g.K) v = (double2_t){0.0, 0.0};
        *(double2_t *)(Bs + (buf * BK + rb + p * RB) * LDB + cb) = v;
      }
    };
    auto multiply = [&](int buf) {
      const double *Ab = As + buf * BK * LDA + wm * WM + lr;
      const double *Bb = Bs + buf * BK * LDB + wn * WN + lr;
#pragma unroll
      for (int ks = 0; ks < BK / 4; ks++) {
        double af[TM], bf[TN];
#pragma unroll
        for (int x = 0; x < TM; x++) af[x] = Ab[(ks * 4 + lk) * LDA + 16 * x];
#pragma unroll
        for (int y = 0; y < TN; y++) bf[y] = Bb[(ks * 4 + lk) * LDB + 16 * y];
#pragma unroll
        for (int x = 0; x < TM; x++)
#pragma unroll
          for (int y = 0; y < TN; y++) acc[x][y] = mfma_f64(af[x], bf[y], acc[x][y]);
      }
    };
    if (s1 <= s0) return;  // (uniform)
#pragma unroll
    for (int d = 0; d < D; d++) gload(sa[d], sb[d], s0 + d);
    lstore(0, sa[0], sb[0], s0);
    __syncthreads();
    // whole groups of D slabs (D even or 1: the LDS buffer of a step is its position in the group, mod 2), straight-line
    int s = s0;
    for (; s + D <= s1; s += D) {
#pragma unroll
      for (int d = 0; d < D; d++) {
        const int buf = d & 1;
        gload(sa[d], sb[d], s + d + D);  // set d: its slab went to LDS one step ago
        multiply(D == 1 ? ((s - s0) & 1) : buf);
        lstore(D == 1 ? (((s - s0) & 1) ^ 1) : (buf ^ 1), sa[(d + 1) % D], sb[(d + 1) % D], s + d + 1);
        __syncthreads();
      }
    }
    // the remaining 0 .. D - 1 slabs one by one (set (s - s0) % D holds slab s + 1 ... the sets rotate as above)
#pragma unroll
    for (int d = 0; d < D - 1; d++) {
      if (s + d < s1) {  // (uniform)
        multiply((D == 1 ? (s + d - s0) : d) & 1);
        if (s + d + 1 < s1) lstore(((D == 1 ? (s + d - s0) : d) & 1) ^ 1, sa[(d + 1) % D], sb[(d + 1) % D], s + d + 1);
        __syncthreads();
      }
    }
  }

  // The same with the operand slabs brought global -> LDS by the DMA path (global_load_lds_dwordx4: no
  // staging registers, no ds_write, no vector ALU work besides the address of a row), 128-wide tiles only: a
  // k-row of a panel is 128 doubles = the 1 KiB one wave-instruction writes (lane l -> bytes 16 l .. 16 l + 15
  // behind a wave-uniform LDS address), so padded LDS rows are no obstacle.  Wave w brings the rows w, w + 4,
  // w + 8, w + 12 of both panels: 8 instructions per wave and slab, issued BETWEEN the first 16 multiplications
  // of the slab before (one per two v_mfma_f64_16x16x4, which take 64 cycles each), into the buffer the
  // barrier at the end of the slab before has released; they have the rest of the slab (~4000 cycles) to land
  // and are waited for (vmcnt(0)) in front of the barrier that ends the slab.  Rows k >= K come from g.zeros,
  // so the last, partial slab needs no masking; behind the last slab of the range the same 8 instructions
  // copy zero rows into the buffer nobody reads any more (no branch in the loop).
  // Per slab a wave is outside its MFMA stream only for the barrier and the latency of its first fragment
  // reads: the register-staged loop above spends ~150 vector instructions per slab on addresses, masks and
  // ds_write_b128 behind the last MFMA, during which the matrix pipe has nothing from this wave (two
  // workgroups per CU that started together stay in step, so the partner wave is in the same phase).
  static __device__ __forceinline__ void glds16(const double *src, double *lds_row) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)lds_row, 16, 0, 0);
  }
  // `skip_upper`: the tile lies on the diagonal of a lower-triangular product - its 16 x 16 blocks strictly above the
  // diagonal are not needed.  Blocks of 16 rows / columns beyond M / N (the ragged last tile row and column: 5000 = 39 x
  // 128 + 8) and those blocks are left out of the multiplications: a wave whose blocks are all wanted runs the plain
  // loop, the others a copy of it with a (wave-uniform, scalar) test in front of every MFMA.  The operands are staged
  // and the barriers kept as always; what is saved is the matrix pipe's time, which the partner workgroup of the CU
  // gets (3.8 % of W's and 5.3 % of G's multiplications at the C4 shapes).
  template <bool MASKED>
  static __device__ __forceinline__ void slabs_dma(const GemmArgs &g, const double *pa, const double *pb, const double *zr, int wave,
                                                   int wm, int wn, int lr, int lk, int s0, int s1, unsigned mask,
                                                   double4_t (&acc)[TM][TN], double *As, double *Bs) {
    constexpr int RPW = BK / NW;  // rows of each panel per wave and slab
    // piece p of the slab that starts at row k0 -> buffer buf: row wave + NW (p % RPW) of A (p < RPW) or B
    auto dma = [&](int buf, int k0, int p) {
      const int r = wave + NW * (p % RPW), k = k0 + r;
      if (p < RPW)
        glds16(k < g.K ? pa + (long long)k * g.lda : zr, As + (buf * BK + r) * LDA);
      else
        glds16(k < g.K ? pb + (long long)k * g.ldb : zr, Bs + (buf * BK + r) * LDB);
    };
    if (s1 > s0) {
#pragma unroll
      for (int p = 0; p < 2 * RPW; p++) dma(0, s0 * BK, p);
    }
    __syncthreads();  // (waits for the DMA: vmcnt(0))
    constexpr int GAP = TM * TN / (2 * RPW);  // multiplications between two pieces
    // (measured and dropped: the waves w and w + 4, which share a SIMD, issuing their pieces two k-steps apart: 8192^3
    // 90.8 -> 88.3 % of peak)
    for (int s = s0; s < s1; s++) {
      const int buf = (s - s0) & 1;
      const int knext = s + 1 < s1 ? (s + 1) * BK : g.K;  // behind the last slab: zero rows
      const double *Ab = As + buf * BK * LDA + wm * WM + lr;
      const double *Bb = Bs + buf * BK * LDB + wn * WN + lr;
#pragma unroll
      for (int ks = 0; ks < BK / 4; ks++) {
        double af[TM], bf[TN];
#pragma unroll
        for (int x = 0; x < TM; x++) af[x] = Ab[(ks * 4 + lk) * LDA + 16 * x];
#pragma unroll
        for (int y = 0; y < TN; y++) bf[y] = Bb[(ks * 4 + lk) * LDB + 16 * y];
#pragma unroll
        for (int x = 0; x < TM; x++)
#pragma unroll
          for (int y = 0; y < TN; y++) {
            if (!MASKED || ((mask >> (x * TN + y)) & 1u)) acc[x][y] = mfma_f64(af[x], bf[y], acc[x][y]);
            if (ks == 0 && (x * TN + y) % GAP == GAP - 1) dma(buf ^ 1, knext, (x * TN + y) / GAP);
          }
      }
      __syncthreads();
    }
  }
  // The same loop over THREE LDS buffers (110 KB: one workgroup per CU): the DMA of slab s + 2 is issued during slab s
  // and has two slab times to land - with two buffers the DMA of slab s + 1, issued at the start of slab s, is waited
  // for at its end, and under load (every CU streaming its panels out of L2) its 1-2 us do not always fit into the
  // 1.7 us a slab takes a workgroup that has the CU to itself.  Counted wait: vmcnt(2 RPW) leaves the newest slab's
  // pieces in flight across the barrier (raw s_barrier: __syncthreads() would drain them).  As / Bs: 3 BK rows each.
  template <bool MASKED>
  static __device__ __forceinline__ void slabs_dma3(const GemmArgs &g, const double *pa, const double *pb, const double *zr, int wave,
                                                    int wm, int wn, int lr, int lk, int s0, int s1, unsigned mask,
                                                    double4_t (&acc)[TM][TN], double *As, double *Bs) {
    constexpr int RPW = BK / NW;
    static_assert(2 * RPW == 4 || 2 * RPW == 8, "the counted waits below are written for 4 or 8 pieces per wave and slab");
    auto dma = [&](int buf, int k0, int p) {
      const int r = wave + NW * (p % RPW), k = k0 + r;
      if (p < RPW)
        glds16(k < g.K ? pa + (long long)k * g.lda : zr, As + (buf * BK + r) * LDA);
      else
        glds16(k < g.K ? pb + (long long)k * g.ldb : zr, Bs + (buf * BK + r) * LDB);
    };
    auto wait_all_but_newest_slab = [&]() {
      if constexpr (2 * RPW == 4)
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    // slabs s0 and s0 + 1 (zero rows where the range is shorter) -> buffers 0 and 1
#pragma unroll
    for (int p = 0; p < 2 * RPW; p++) dma(0, s1 > s0 ? s0 * BK : g.K, p);
#pragma unroll
    for (int p = 0; p < 2 * RPW; p++) dma(1, s0 + 1 < s1 ? (s0 + 1) * BK : g.K, p);
    wait_all_but_newest_slab();
    constexpr int GAP = TM * TN / (2 * RPW);
    int buf = 0;
    for (int s = s0; s < s1; s++) {
      const int bnext = buf >= 1 ? buf - 1 : 2;  // (buf + 2) % 3
      const int knext = s + 2 < s1 ? (s + 2) * BK : g.K;
      const double *Ab = As + buf * BK * LDA + wm * WM + lr;
      const double *Bb = Bs + buf * BK * LDB + wn * WN + lr;
#pragma unroll
      for (int ks = 0; ks < BK / 4; ks++) {
        double af[TM], bf[TN];
#pragma unroll
        for (int x = 0; x < TM; x++) af[x] = Ab[(ks * 4 + lk) * LDA + 16 * x];
#pragma unroll
        for (int y = 0; y < TN; y++) bf[y] = Bb[(ks * 4 + lk) * LDB + 16 * y];
#pragma unroll
        for (int x = 0; x < TM; x++)
#pragma unroll
          for (int y = 0; y < TN; y++) {
            if (!MASKED || ((mask >> (x * TN + y)) & 1u)) acc[x][y] = mfma_f64(af[x], bf[y], acc[x][y]);
            if (ks == 0 && (x * TN + y) % GAP == GAP - 1) dma(bnext, knext, (x * TN + y) / GAP);
          }
      }
      wait_all_but_newest_slab();
      buf = buf == 2 ? 0 : buf + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the zero rows behind the range: LDS is reused after this)
    __syncthreads();
  }
  static __device__ __forceinline__ void accumulate_dma(const GemmArgs &g, int i0, int j0, int s0, int s1,
                                                        double4_t (&acc)[TM][TN], double *As, double *Bs, bool skip_upper = false, int nbuf = 2) {
    static_assert(BM == 128 && BN == 128, "one k-row of a panel must be one 1-KiB wave-instruction");
    static_assert((BK / NW) * NW == BK && TM * TN >= 2 * (BK / NW) && TM * TN <= 32,
                  "pieces are issued behind the multiplications of the first k-step");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int lr = lane & 15, lk = lane >> 4;
    // a 16-byte load is inside its row when its first column is < ld (ld even)
    GemmArgs gl = g;  // (B and its leading dimension: the strip of the tile's columns)
    int jb = j0;
    if (g.bstrips) {
      const int q = strip_of(g.bstrips->cut, g.bstrips->nranks, j0);
      gl.B = g.B + g.bstrips->off[q], gl.ldb = g.bstrips->ld[q], jb = j0 - g.bstrips->cut[q];
    }
    const double *pa = g.A + ((i0 + 2 * lane < g.lda) ? i0 + 2 * lane : 0);
    const double *pb = gl.B + ((jb + 2 * lane < gl.ldb) ? jb + 2 * lane : 0);
    const double *zr = g.zeros + 2 * lane;
    unsigned mask = 0;
#pragma unroll
    for (int x = 0; x < TM; x++)
#pragma unroll
      for (int y = 0; y < TN; y++) {
        const int rb = wm * TM + x, cb = wn * TN + y;  // 16 x 16 block of the tile
        const bool want = i0 + 16 * rb < g.M && j0 + 16 * cb < g.N && !(skip_upper && cb > rb);
        mask |= (want ? 1u : 0u) << (x * TN + y);
      }
    mask = __builtin_amdgcn_readfirstlane(mask);
    const bool all = mask == (TM * TN == 32 ? 0xffffffffu : (1u << (TM * TN)) - 1u);
    if (nbuf == 3) {
      if (all)
        slabs_dma3<false>(gl, pa, pb, zr, wave, wm, wn, lr, lk, s0, s1, mask, acc, As, Bs);
      else
        slabs_dma3<true>(gl, pa, pb, zr, wave, wm, wn, lr, lk, s0, s1, mask, acc, As, Bs);
    } else if (all)
      slabs_dma<false>(gl, pa, pb, zr, wave, wm, wn, lr, lk, s0, s1, mask, acc, As, Bs);
    else
      slabs_dma<true>(gl, pa, pb, zr, wave, wm, wn, lr, lk, s0, s1, mask, acc, As, Bs);
  }

  // `lds`: the workgroup's LDS (free after accumulate's last barrier), used to write the MIRROR image of an
  // off-diagonal tile in whole rows: the values of 64 tile columns at a time go to LDS transposed ([column][row],
  // leading dimension BM + 2: conflict-free), and every wave then writes rows of the mirrored block in 1-KiB (BM = 128)
  // pieces - written element by element the image costs one 32-byte sector per value (the rank-q update V = G_xx -
  // Y'Rm of a C4 stage, which is nothing but reading G and writing V and its image: 158 us, 1.9 TB/s).  All threads
  // of the workgroup must call (barriers inside when g.mirror is set).
  static __device__ __forceinline__ void epilogue(const GemmArgs &g, int tm, int tn, const double4_t (&acc)[TM][TN], double *lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int lr = lane & 15, lk = lane >> 4;
    const int i0 = tm * BM, j0 = tn * BN;
    const bool diag = g.lower && tm == tn;
    const double *cin = g.Cin;
    long long ldcin = g.ldcin;
    if (g.rects && g.beta != 0.0) {  // the block (part) that holds this tile
      const RectTab &R = *g.rects;
      const int a = strip_of(R.cut, R.nranks, i0), b = strip_of(R.cut, R.nranks, j0);
      const RectTab::Block &blk = R.blk[a * 16 + b];
      const int t = i0 >= blk.rsplit ? 1 : 0, r0 = t ? blk.rsplit : R.cut[a], c0 = R.cut[b];
      ldcin = R.cut[b + 1] - c0;
      cin = g.Cin + blk.off[t] - ((long long)r0 * ldcin + c0);
    }
    constexpr int HC = BN < 64 ? BN : 64, LDT = BM + 2;  // columns per pass of the mirrored write
    const bool via_lds = g.mirror && !diag && lds != nullptr;
    double4_t val[TM][TN];
#pragma unroll
    for (int x = 0; x < TM; x++)
#pragma unroll
      for (int y = 0; y < TN; y++)
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
          const int i = i0 + wm * WM + 16 * x + lk + 4 * rg, j = j0 + wn * WN + 16 * y + lr;
          double v = 0.0;
          if (i < g.M && j < g.N && !(diag && i < j)) {
            v = g.alpha * acc[x][y][rg];
            if (g.beta != 0.0) v += g.beta * cin[(long long)i * ldcin + j];
            g.C[(long long)i * g.ldc + j] = v;
            if (g.mirror && i != j && !via_lds) g.C[(long long)j * g.ldc + i] = v;
          }
          val[x][y][rg] = v;
        }
    if (via_lds) {  // (uniform for the workgroup)
#pragma unroll
      for (int h = 0; h < BN / HC; h++) {
        if ((wn * WN) / HC == h) {
#pragma unroll
          for (int x = 0; x < TM; x++)
#pragma unroll
            for (int y = 0; y < TN; y++)
#pragma unroll
              for (int rg = 0; rg < 4; rg++)
                lds[(wn * WN - h * HC + 16 * y + lr) * LDT + wm * WM + 16 * x + lk + 4 * rg] = val[x][y][rg];
        }
        __syncthreads();
        // row jj of the image = column j0 + h HC + jj of the tile: BM values, two per lane and row
        for (int jj = wave; jj < HC; jj += NW) {
          const int j = j0 + h * HC + jj;
          if (j >= g.N) break;
          for (int ii = 2 * lane; ii < BM; ii += 128) {
            const int i = i0 + ii;
            double *dst = g.C + (long long)j * g.ldc + i;
            if (i + 1 < g.M && (((size_t)dst) & 15) == 0)
              *(double2_t *)dst = *(const double2_t *)(lds + jj * LDT + ii);
            else {
              if (i < g.M) dst[0] = lds[jj * LDT + ii];
              if (i + 1 < g.M) dst[1] = lds[jj * LDT + ii + 1];
            }
          }
        }
        __syncthreads();
      }
    }
  }
};

// wavefronts per SIMD the launch is compiled for: two workgroups per CU (one with three LDS buffers)
// (A 256 x 128 tile on 4 x 4 wavefronts, one workgroup per CU, was written and measured in round 4 - commit 44e8e46,
// profiles/r04_tile256_ab.txt: 88.4 % of the peak at 8192^3 against 89.7 % of the 2 x 4 form, 64 % against 79.5 % on
// the 800 tiles of a C4 stage's W - and taken out again; in the split form (whole rounds + cut remainder) it reaches
// 80.7 % on W against 81.0 % of the form in use: the shape's ceiling - 313 slabs per tile, ragged last tile row and
// column - not the pairing of workgroups, is what holds W at 81 %.)
constexpr int gemm_waves_per_simd(int nw, int nbuf) { return nbuf == 3 ? nw / 4 : nw / 2; }
template <int BM, int BN, bool DMA = false, int WGM = 2, int WGN = 2, int NBUF = 2>
__global__ void __launch_bounds__(64 * WGM * WGN, gemm_waves_per_simd(WGM * WGN, NBUF)) k_dgemm_tn(GemmArgs g) {
  using T = GemmTile<BM, BN, WGM, WGN>;
  extern __shared__ __attribute__((aligned(16))) double lds[];  // NBUF * BK * (LDA + LDB) doubles
  double *As = lds, *Bs = lds + NBUF * T::BK * ((DMA && BM == 64) ? 64 : T::LDA);  // (the 64 x 64 DMA form: unpadded rows)
  int tm, tn;
  const unsigned long long t0 = g.stamps ? __builtin_amdgcn_s_memrealtime() : 0;
  T::tile_of(g, xcd_swizzle(blockIdx.x, gridDim.x), tm, tn);
  double4_t acc[T::TM][T::TN];
#pragma unroll
  for (int x = 0; x < T::TM; x++)
#pragma unroll
    for (int y = 0; y < T::TN; y++) acc[x][y] = (double4_t){0.0, 0.0, 0.0, 0.0};
  if constexpr (DMA)
    T::accumulate_dma(g, tm * BM, tn * BN, 0, (g.K + T::BK - 1) / T::BK, acc, As, Bs, g.lower && tm == tn, NBUF);
  else
    T::accumulate(g, tm * BM, tn * BN, 0, (g.K + T::BK - 1) / T::BK, acc, As, Bs);
  const unsigned long long t2 = g.stamps ? __builtin_amdgcn_s_memrealtime() : 0;
  T::epilogue(g, tm, tn, acc, lds);
  if (g.stamps && threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g.stamps[4 * blockIdx.x + 0] = t0, g.stamps[4 * blockIdx.x + 1] = (unsigned long long)(xcc & 15);
    g.stamps[4 * blockIdx.x + 2] = t2, g.stamps[4 * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
  }
}

// A THIN product - a handful of tiles, thousands of k (the control rows of G: 50 x 690 x 5000; the carried rows) - is a
// chain of 313 slabs in each of its few workgroups: 330 us for 0.3 GFlop.  Cut in k: workgroup (tile, y) sums the slabs
// of piece y into part[y] (M x N, raw sums), k_dgemm_ks_finish adds the pieces in their order (reproducible) and applies
// alpha / beta.  Not lower, not mirrored; register-staged 64-wide tiles.
template <int BM, int BN>
__global__ void __launch_bounds__(256) k_dgemm_tn_ks(GemmArgs g, double *__restrict__ part, int nsplit) {
  using T = GemmTile<BM, BN, 2, 2>;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double *As = lds, *Bs = lds + 2 * T::BK * T::LDA;
  int tm, tn;
  T::tile_of(g, blockIdx.x, tm, tn);
  const int nslab = (g.K + T::BK - 1) / T::BK, L = (nslab + nsplit - 1) / nsplit;
  const int s0 = min(nslab, (int)blockIdx.y * L), s1 = min(nslab, s0 + L);
  double4_t acc[T::TM][T::TN];
#pragma unroll
  for (int x = 0; x < T::TM; x++)
#pragma unroll
    for (int y = 0; y < T::TN; y++) acc[x][y] = (double4_t){0.0, 0.0, 0.0, 0.0};
  T::accumulate(g, tm * BM, tn * BN, s0, s1, acc, As, Bs);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave / 2, wn = wave % 2, lr = lane & 15, lk = lane >> 4;
  double *out = part + (long long)blockIdx.y * g.M * g.N;
#pragma unroll
  for (int x = 0; x < T::TM; x++)
#pragma unroll
    for (int y = 0; y < T::TN; y++)
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        const int i = tm * BM + wm * T::WM + 16 * x + lk + 4 * rg, j = tn * BN + wn * T::WN + 16 * y + lr;
        if (i < g.M && j < g.N) out[(long long)i * g.N + j] = acc[x][y][rg];
      }
}
__global__ void k_dgemm_ks_finish(GemmArgs g, const double *__restrict__ part, int nsplit) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x, tot = (long long)g.M * g.N;
  if (e >= tot) return;
  double s = 0.0;
  for (int y = 0; y < nsplit; y++) s += part[(long long)y * tot + e];
  const int i = (int)(e / g.N), j = (int)(e % g.N);
  double v = g.alpha * s;
  if (g.beta != 0.0) v += g.beta * g.Cin[(long long)i * g.ldcin + j];
  g.C[(long long)i * g.ldc + j] = v;
}

// The same product for tile counts that do not fill the chip evenly (1600 tiles on 512 workgroup
// slots: the last of four rounds would be an eighth full; 100 tiles of a column slice: a fifth of
// the slots busy for a whole tile time).  A fixed grid of G workgroups (two per CU) runs
//   * `dp_rounds` rounds of whole tiles, then
//   * up to three SPLIT phases: phase q takes count[q] tiles and cuts the k range of each into split[q]
//     equal pieces: the last whole round as halves, the rest of the tiles floor(G / rest) ways.
// In every phase the workgroups that run side by side (neighbours in w: the same XCD after the swizzle)
// work on neighbouring tiles at the SAME k, so they share their operand panels in that XCD's L2 -
// round 2's form of this kernel gave every workgroup a contiguous range of (tile, k-slab) units, which
// balances as well but leaves the 512 workgroups at 512 different k: 3 GB of operand reads in 0.7 ms
// that no cache level could share (the Infinity Cache holds 256 MB, F and W are 200 MB each), and the
// shared part of a launch ran at the speed of those reads, not of the matrix pipes.
// The pieces of a tile park their partial sums (plain stores, then an agent-scope release and one
// counter add); the workgroup that arrives last adds all of them in the order of the k ranges (its
// own comes back from memory too: one code path, one order) and writes the tile: the result does not
// depend on the order of arrival, and nobody waits for anybody.
struct SplitPlan {
  double *ws;     // partial tiles: piece p (numbered through the split phases) -> ws + p * 128 * 128 (gemm_split_plan_pieces)
  unsigned *cnt;  // arrival counter per tile (zero between launches: the last arriver of a tile resets it)
  int dp_rounds;  // (whole / G, informative)
  int whole;      // the first `whole` tiles are computed whole
  int nphase;
  int begin[3], count[3], split[3];
  // The fractional form (gemm_split_plan_frac): the tiles' k-slabs in one sequence (tile t holds the units t nslab ...),
  // cut into `per` units per workgroup - workgroup w takes [w per, (w + 1) per): the end of one tile, whole tiles, the
  // start of another.  A tile that several workgroups share is summed by its last arriver in the order of the
  // workgroups; a workgroup parks at most two partial tiles (slots 2 w: the tile its range starts in, 2 w + 1: the
  // tile it ends in).  For products of a few hundred tiles (stages of 1000 - 3000 states, the strips of a system
  // over several ranks), where whole rounds and cut remainders leave a large part of the chip idle.
  int frac, per, ntiles;
  // The table form (gemm_split_table): the units of every workgroup listed by the host - workgroup b (blockIdx.x) does
  // table[b * stride + i], i = 0 ... until a tile < 0 (SkUnit, sk_table.hpp).
  const SkUnit *table;
  int stride;
};

// Host: the plan for `tiles` tiles of `nslab` k-slabs on `grid` workgroups.  The remainder R of the whole
// rounds is cut floor(grid / R) ways; a remainder of more than half a round first gives grid / 2 tiles to
// two workgroups each (a full phase of half the depth) and cuts the rest after that.  A piece holds at
// least 16 slabs (below that the pipeline fill of a piece and the parked partial sums cost more than the
// balance gains) and a tile has at most 16 pieces (the last arriver reads them one after the other).
static inline SplitPlan gemm_split_plan(long long tiles, long long nslab, int grid) {
  SplitPlan sp{};
  const long long smax = std::max<long long>(1, std::min<long long>(16, nslab / 16));
  sp.dp_rounds = (int)(tiles / grid);
  long long R = tiles - (long long)sp.dp_rounds * grid, begin = (long long)sp.dp_rounds * grid;
  sp.whole = (int)begin;
  while (R > 0 && sp.nphase < 2) {
    long long s = std::min<long long>(smax, grid / R), r = R;
    if (s <= 1) {
      s = 1;
      // (the half round needs an even grid: with an odd one workgroup grid - 1 would start on the next phase's first
      // unit, which workgroup 0 takes as well)
      if (sp.nphase == 0 && smax >= 2 && R > grid / 2 && grid % 2 == 0) s = 2, r = grid / 2;
    }
    sp.begin[sp.nphase] = (int)begin, sp.count[sp.nphase] = (int)r, sp.split[sp.nphase] = (int)s;
    sp.nphase++, begin += r, R -= r;
  }
  return sp;
}
static inline SplitPlan gemm_split_plan_frac(long long tiles, long long nslab, int grid) {
  SplitPlan sp{};
  sp.frac = 1, sp.ntiles = (int)tiles;
  sp.per = (int)((tiles * nslab + grid - 1) / grid);
  return sp;
}
static inline long long gemm_split_plan_pieces(const SplitPlan &sp) {
  long long n = 0;
  for (int q = 0; q < sp.nphase; q++) n += (long long)sp.count[q] * sp.split[q];
  return n;
}
// time of the plan in units of one k-slab of one workgroup (what the launch heuristics compare)
static inline long long gemm_split_plan_depth(const SplitPlan &sp, long long nslab) {
  long long d = (long long)sp.dp_rounds * nslab;
  for (int q = 0; q < sp.nphase; q++) d += (nslab + sp.split[q] - 1) / sp.split[q];
  return d;
}
static inline bool gemm_use_split(int M, int N, int K, int lower, int grid) {
  if (grid <= 0 || (long long)M * N < 256LL * 256) return false;
  const long long tiles = gemm_tiles(M, N, 128, lower);
  const long long nslab = (K + GEMM_BK - 1) / GEMM_BK;
  if (nslab < 32) return false;                                 // too shallow to cut
  if (tiles % grid == 0 || tiles >= 16LL * grid) return false;  // even, or the tail does not matter
  // (a CU with one workgroup reaches 92 % of what it does with two: up to 5/8 of the grid one plain round of one
  // or two workgroups per CU is as fast as cut pieces, without their parked partial sums)
  if (tiles > grid * 5 / 8) return true;
  // (few tiles - a stage of ~1000 states: 72 - run on 64 x 64 tiles; cut pieces for them were measured slower)
  return false;
}
// The fractional form pays for a few hundred tiles - between 5/16 and 5/8 of the grid, where neither whole rounds nor
// the 64 x 64 tiles fill the chip (measured, one MI355X, tools/dgemm_shapes.py: W of a stage of 2000 states, 272 tiles:
// 367 us against 413; G of 3000 states, 300 lower tiles: 532 against 609; a 640-column strip of the headline's W, 200
// tiles: 611 against 652).  Its workgroups are at different k at any moment, so they share less of the operands in L2
// than the rounds of the plan above: with more tiles (the headline's 1600: 4.13 against 3.90 ms) the plan stays.
static inline bool gemm_use_frac(int M, int N, int K, int lower, int grid) {
  if (grid <= 0) return false;
  const long long tiles = gemm_tiles(M, N, 128, lower), nslab = (K + GEMM_BK - 1) / GEMM_BK;
  return nslab >= 64 && tiles * 16 >= grid * 5LL && tiles * 8 <= grid * 5LL;
}
template <bool DMA, int WGM = 2, int WGN = 2, int NBUF = 2, int BM = 128, int BN = 128>
__global__ void __launch_bounds__(64 * WGM * WGN, NBUF == 3 ? WGM * WGN / 4 : WGM * WGN / 2) k_dgemm_tn_sk(GemmArgs g, SplitPlan sk) {
  using T = GemmTile<BM, BN, WGM, WGN>;
  extern __shared__ __attribute__((aligned(16))) double lds[];  // tiles + one word for the arrival order
  double *As = lds, *Bs = lds + NBUF * T::BK * T::LDA;
  unsigned *s_old = (unsigned *)(lds + NBUF * T::BK * (T::LDA + T::LDB));
  const int G = gridDim.x, v = xcd_swizzle(blockIdx.x, G);
  const int nslab = (g.K + T::BK - 1) / T::BK;
  constexpr int SLOT = BM * BN;
  unsigned long long *stamp = g.stamps ? g.stamps + 32 * (long long)blockIdx.x : nullptr;  // (diagnostic launches only)
  if (stamp && threadIdx.x == 0) stamp[0] = __builtin_amdgcn_s_memrealtime();
  // Units of work: the whole tiles of the rounds (unit u = tile u), then the pieces of the split phases (phase q:
  // unit = piece j of tile ti at j * count[q] + ti).  Workgroup w does unit w of every round and phase: its
  // neighbours in the XCD work on the neighbouring tiles at the same k.  (A queue of units was measured in round 3
  // and does not pay: profiles/NOTES.md.)  The result does not depend on who computes what: a tile's pieces are fixed
  // k ranges, summed in their order.
  if (sk.table) {
    auto unit = [&](const SkUnit u, int r) {
      const int t = u.tile, s0 = u.s0, s1 = u.s1, pieces = u.pieces, j = u.j;
      int tm, tn;
      T::tile_of(g, t, tm, tn);
      double4_t acc[T::TM][T::TN];
#pragma unroll
      for (int x = 0; x < T::TM; x++)
#pragma unroll
        for (int y = 0; y < T::TN; y++) acc[x][y] = (double4_t){0.0, 0.0, 0.0, 0.0};
      if constexpr (DMA)
        T::accumulate_dma(g, tm * BM, tn * BN, s0, s1, acc, As, Bs, g.lower && tm == tn, NBUF);
      else
        T::accumulate(g, tm * BM, tn * BN, s0, s1, acc, As, Bs);
      bool finish = true;
      if (stamp && threadIdx.x == 0 && r < 10) stamp[1 + 3 * r] = __builtin_amdgcn_s_memrealtime();
      if (pieces > 1) {
        double *mine = sk.ws + (long long)(u.slot0 + j) * SLOT;
#pragma unroll
        for (int x = 0; x < T::TM; x++)
#pragma unroll
          for (int y = 0; y < T::TN; y++)
#pragma unroll
            for (int rg = 0; rg < 4; rg++) mine[((x * T::TN + y) * 4 + rg) * T::NT + threadIdx.x] = acc[x][y][rg];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          *s_old = __hip_atomic_fetch_add(sk.cnt + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        finish = *s_old == (unsigned)(pieces - 1);
        if (finish) {
          if (threadIdx.x == 0) {
            sk.cnt[t] = 0;  // (every piece of the tile has arrived: the counter is ready for the next launch)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
          __syncthreads();
#pragma unroll
          for (int x = 0; x < T::TM; x++)
#pragma unroll
            for (int y = 0; y < T::TN; y++) acc[x][y] = (double4_t){0.0, 0.0, 0.0, 0.0};
          for (int jj = 0; jj < pieces; jj++) {  // in the order of the k ranges, whoever arrived last
            const double *theirs = sk.ws + (long long)(u.slot0 + jj) * SLOT;
#pragma unroll
            for (int x = 0; x < T::TM; x++)
#pragma unroll
              for (int y = 0; y < T::TN; y++)
#pragma unroll
                for (int rg = 0; rg < 4; rg++) acc[x][y][rg] += theirs[((x * T::TN + y) * 4 + rg) * T::NT + threadIdx.x];
          }
        }
        __syncthreads();  // s_old is rewritten at the next shared tile
      }
      if (stamp && threadIdx.x == 0 && r < 10) stamp[2 + 3 * r] = __builtin_amdgcn_s_memrealtime();
      if (finish) T::epilogue(g, tm, tn, acc, lds);  // (uniform: the whole workgroup)
      if (stamp && threadIdx.x == 0 && r < 10) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp[3 + 3 * r] = __builtin_amdgcn_s_memrealtime();
      }
      __syncthreads();
    };
    const SkUnit *tab = sk.table + (long long)blockIdx.x * sk.stride;
    for (int r = 0; r < sk.stride; r++) {
      const SkUnit u = tab[r];
      if (u.tile < 0) break;
      unit(u, r);
    }
    return;
  }
  if (sk.frac) {
    const int U = sk.ntiles * nslab, per = sk.per;
    const int lo = min(U, v * per), hi = min(U, lo + per), t_first = lo / nslab;
    for (int x = lo; x < hi;) {
      const int t = x / nslab, s0 = x - t * nslab, s1 = min(nslab, s0 + (hi - x));
      const int w_first = (t * nslab) / per, w_last = ((t + 1) * nslab - 1) / per, pieces = w_last - w_first + 1;
      int tm, tn;
      T::tile_of(g, t, tm, tn);
      double4_t acc[T::TM][T::TN];
#pragma unroll
      for (int xx = 0; xx < T::TM; xx++)
#pragma unroll
        for (int y = 0; y < T::TN; y++) acc[xx][y] = (double4_t){0.0, 0.0, 0.0, 0.0};
      if constexpr (DMA)
        T::accumulate_dma(g, tm * BM, tn * BN, s0, s1, acc, As, Bs, g.lower && tm == tn, NBUF);
      else
        T::accumulate(g, tm * BM, tn * BN, s0, s1, acc, As, Bs);
      bool finish = true;
      if (pieces > 1) {
        double *mine = sk.ws + (long long)(2 * v + (t == t_first ? 0 : 1)) * SLOT;
#pragma unroll
        for (int xx = 0; xx < T::TM; xx++)
#pragma unroll
          for (int y = 0; y < T::TN; y++)
#pragma unroll
            for (int rg = 0; rg < 4; rg++) mine[((xx * T::TN + y) * 4 + rg) * T::NT + threadIdx.x] = acc[xx][y][rg];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          *s_old = __hip_atomic_fetch_add(sk.cnt + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        finish = *s_old == (unsigned)(pieces - 1);
        if (finish) {
          if (threadIdx.x == 0) {
            sk.cnt[t] = 0;  // (ready for the next launch)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
          __syncthreads();
#pragma unroll
          for (int xx = 0; xx < T::TM; xx++)
#pragma unroll
            for (int y = 0; y < T::TN; y++) acc[xx][y] = (double4_t){0.0, 0.0, 0.0, 0.0};
          for (int w = w_first; w <= w_last; w++) {  // in the order of the k ranges, whoever arrived last
            const double *theirs = sk.ws + (long long)(2 * w + (t == (w * per) / nslab ? 0 : 1)) * SLOT;
#pragma unroll
            for (int xx = 0; xx < T::TM; xx++)
#pragma unroll
              for (int y = 0; y < T::TN; y++)
#pragma unroll
                for (int rg = 0; rg < 4; rg++) acc[xx][y][rg] += theirs[((xx * T::TN + y) * 4 + rg) * T::NT + threadIdx.x];
          }
        }
        __syncthreads();  // s_old is rewritten at the next shared tile
      }
      if (finish) T::epilogue(g, tm, tn, acc, lds);  // (uniform: the whole workgroup)
      __syncthreads();
      x += s1 - s0;
    }
    return;
  }
  const int n_whole = sk.whole;
  int n_units = n_whole;
  for (int q = 0; q < sk.nphase; q++) n_units += sk.count[q] * sk.split[q];
  int r = 0;
  for (int u = v; u < n_units; r++) {
    int q = -1;
    int t = u, s0 = 0, s1 = nslab, cnt = G, pieces = 1, ti = 0, pbase = 0;
    if (u >= n_whole) {
      int rel = u - n_whole;
      q = 0;
      while (q + 1 < sk.nphase && rel >= sk.count[q] * sk.split[q]) rel -= sk.count[q] * sk.split[q], pbase += sk.count[q] * sk.split[q], q++;
      cnt = sk.count[q], pieces = sk.split[q];
      ti = rel % cnt;
      const int L = (nslab + pieces - 1) / pieces, j = rel / cnt;
      t = sk.begin[q] + ti, s0 = min(nslab, j * L), s1 = min(nslab, s0 + L);
    }
    const int u_now = u;
    {  // unit w of the next round / phase (the plan's phases hold at most G units each)
      int nu = n_units;
      if (u + G < n_whole)
        nu = u + G;
      else {
        int base = n_whole, qq = 0;
        if (u >= n_whole) base += pbase + cnt * pieces, qq = q + 1;
        for (; qq < sk.nphase && nu == n_units; qq++) {
          if (v < sk.count[qq] * sk.split[qq]) nu = base + v;
          base += sk.count[qq] * sk.split[qq];
        }
      }
      if (threadIdx.x == 0) s_old[1] = (unsigned)nu;
    }
    int tm, tn;
    T::tile_of(g, t, tm, tn);
    double4_t acc[T::TM][T::TN];
#pragma unroll
    for (int x = 0; x < T::TM; x++)
#pragma unroll
      for (int y = 0; y < T::TN; y++) acc[x][y] = (double4_t){0.0, 0.0, 0.0, 0.0};
    if constexpr (DMA)
      T::accumulate_dma(g, tm * BM, tn * BN, s0, s1, acc, As, Bs, g.lower && tm == tn, NBUF);
    else
      T::accumulate(g, tm * BM, tn * BN, s0, s1, acc, As, Bs);
    bool finish = true;
    if (stamp && threadIdx.x == 0 && r < 5) stamp[1 + 3 * r] = __builtin_amdgcn_s_memrealtime();
    if (pieces > 1) {
      double *mine = sk.ws + (long long)(u_now - n_whole) * SLOT;
#pragma unroll
      for (int x = 0; x < T::TM; x++)
#pragma unroll
        for (int y = 0; y < T::TN; y++)
#pragma unroll
          for (int rg = 0; rg < 4; rg++) mine[((x * T::TN + y) * 4 + rg) * T::NT + threadIdx.x] = acc[x][y][rg];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *s_old = __hip_atomic_fetch_add(sk.cnt + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      finish = *s_old == (unsigned)(pieces - 1);
      if (finish) {
        if (threadIdx.x == 0) {
          sk.cnt[t] = 0;  // (every piece of the tile has arrived: the counter is ready for the next launch - no memset between launches)
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
#pragma unroll
        for (int x = 0; x < T::TM; x++)
#pragma unroll
          for (int y = 0; y < T::TN; y++) acc[x][y] = (double4_t){0.0, 0.0, 0.0, 0.0};
        for (int jj = 0; jj < pieces; jj++) {
          const double *theirs = sk.ws + (long long)(pbase + jj * cnt + ti) * SLOT;
#pragma unroll
          for (int x = 0; x < T::TM; x++)
#pragma unroll
            for (int y = 0; y < T::TN; y++)
#pragma unroll
              for (int rg = 0; rg < 4; rg++) acc[x][y][rg] += theirs[((x * T::TN + y) * 4 + rg) * T::NT + threadIdx.x];
        }
      }
      __syncthreads();  // s_old is rewritten at the next shared tile
    }
    if (stamp && threadIdx.x == 0 && r < 5) stamp[2 + 3 * r] = __builtin_amdgcn_s_memrealtime();
    if (finish) T::epilogue(g, tm, tn, acc, lds);  // (uniform: the whole workgroup)
    if (stamp && threadIdx.x == 0 && r < 5) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      stamp[3 + 3 * r] = __builtin_amdgcn_s_memrealtime();
    }
    __syncthreads();
    u = (int)s_old[1];
    __syncthreads();  // (s_old[1] is rewritten at the top of the next unit)
  }
}
static inline size_t gemm_sk_lds_bytes(int nbuf = 2) { return gemm_lds_bytes(128, 128, nbuf) + 16; }  // + two words: arrival order, next unit

// Host: the variants of the 128 x 128 product.  0: operands staged through registers, 4 waves (round 2's loop, kept
// for comparisons: HQPKKT_NO_LDSDMA); 1: LDS-DMA, 2 x 2 waves of 64 x 64; 2: LDS-DMA, 2 x 4 waves of 64 x 32 (default)
enum { GEMM_REG4 = 0, GEMM_DMA4 = 1, GEMM_DMA8 = 2, GEMM_DMA8X3 = 3 };  // X3: three LDS buffers, one workgroup per CU
static inline int gemm_wgs_per_cu(int variant) { return variant == GEMM_DMA8X3 ? 1 : 2; }
// cus > 0: a launch of at most that many tiles takes the three-buffer kernel, whose 110 KB of LDS admit ONE workgroup
// per CU - the dispatcher otherwise puts two workgroups on some CUs and none on others, and a pair takes twice as long
// as a workgroup alone (the column strip 5000 x 640 x 5000 of a system sharded over 8 ranks: 0.75 -> 0.62 ms)
static inline void gemm_launch_plain(int variant, unsigned tiles, hipStream_t s, const GemmArgs &g, int cus = 0) {
  if (variant == GEMM_DMA8 && cus > 0 && (int)tiles <= cus) variant = GEMM_DMA8X3;
  if (variant == GEMM_DMA8X3)
    k_dgemm_tn<128, 128, true, 2, 4, 3><<<tiles, 512, gemm_lds_bytes(128, 128, 3), s>>>(g);
  else if (variant == GEMM_DMA8)
    k_dgemm_tn<128, 128, true, 2, 4><<<tiles, 512, gemm_lds_bytes(128, 128), s>>>(g);
  else if (variant == GEMM_DMA4)
    k_dgemm_tn<128, 128, true><<<tiles, 256, gemm_lds_bytes(128, 128), s>>>(g);
  else
    k_dgemm_tn<128, 128><<<tiles, 256, gemm_lds_bytes(128, 128), s>>>(g);
}
static inline void gemm_launch_split(int variant, int grid, hipStream_t s, const GemmArgs &g, const SplitPlan &sk) {
  if (variant == GEMM_DMA8X3)
    k_dgemm_tn_sk<true, 2, 4, 3><<<grid, 512, gemm_sk_lds_bytes(3), s>>>(g, sk);
  else if (variant == GEMM_DMA8)
    k_dgemm_tn_sk<true, 2, 4><<<grid, 512, gemm_sk_lds_bytes(), s>>>(g, sk);
  else if (variant == GEMM_DMA4)
    k_dgemm_tn_sk<true><<<grid, 256, gemm_sk_lds_bytes(), s>>>(g, sk);
  else
    k_dgemm_tn_sk<false><<<grid, 256, gemm_sk_lds_bytes(), s>>>(g, sk);
}
// 64 x 64 tiles (register-staged loop) with their k ranges cut: products of a few hundred small tiles, where one
// workgroup per CU leaves the matrix pipe two thirds idle (a stage of ~1000 states: 272 tiles of 63 slabs, 91 us)
// the rule for 64 x 32 tiles (st_gemm, hqpkkt_debug_dgemm): a rectangular product of at most two 64 x 64 tiles per CU, deep
static inline bool gemm_tiles_6432(int M, int N, int K, int lower, int mirror, int cus) {
  return cus > 0 && !lower && !mirror && gemm_tiles(M, N, 64, 0) <= 2LL * cus && K >= 16 * GEMM_BK;
}
static inline hipError_t gemm_set_attributes() {
  hipError_t e = hipSuccess;
  auto set = [&](const void *f, size_t bytes) {
    const hipError_t r = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) e = r;
  };
  set((const void *)k_dgemm_tn<128, 128>, gemm_lds_bytes(128, 128));
  set((const void *)k_dgemm_tn<128, 128, true>, gemm_lds_bytes(128, 128));
  set((const void *)k_dgemm_tn<128, 128, true, 2, 4>, gemm_lds_bytes(128, 128));
  set((const void *)k_dgemm_tn<64, 64>, gemm_lds_bytes(64, 64));
  set((const void *)k_dgemm_tn<64, 32>, gemm_lds_bytes(64, 32));
  set((const void *)k_dgemm_tn_ks<64, 64>, gemm_lds_bytes(64, 64));
  set((const void *)k_dgemm_tn_sk<false>, gemm_sk_lds_bytes());
  set((const void *)k_dgemm_tn_sk<true>, gemm_sk_lds_bytes());
  set((const void *)k_dgemm_tn_sk<true, 2, 4>, gemm_sk_lds_bytes());
  set((const void *)k_dgemm_tn<128, 128, true, 2, 4, 3>, gemm_lds_bytes(128, 128, 3));
  set((const void *)k_dgemm_tn_sk<true, 2, 4, 3>, gemm_sk_lds_bytes(3));
  return e;
}
// (HQPKKT_SK_TABLE=0: the cut form with equal shares, gemm_split_plan, for same-box comparisons)
static inline bool gemm_sk_table_from_env() {
  const char *r = getenv("HQPKKT_SK_TABLE");
  return !r || atoi(r) != 0;
}
static inline int gemm_variant_from_env() {
  if (getenv("HQPKKT_NO_LDSDMA")) return GEMM_REG4;
  const char *w = getenv("HQPKKT_DGEMM_WAVES");
  if (w && atoi(w) == 4) return GEMM_DMA4;
  return GEMM_DMA8;
}

// ---------------------------------------------------------------------------------------
// H = Q + C'(Z/W)C of one stage added into the dense block (term lists as in the REDUCED
// plugin: value = sum sgn * vals[s1] * vals[s2] * wt[wi]); every entry has one writer
struct HTerm {
  int s1, s2, wi;
};
__global__ void k_st_add_h(int nent, const long long *__restrict__ dst, const int *__restrict__ tptr,
                           const HTerm *__restrict__ terms, const double *__restrict__ vals,
                           const double *__restrict__ wt, double *__restrict__ G, int add) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nent) return;
  double s = 0.0;
  for (int k = tptr[e]; k < tptr[e + 1]; k++) s += vals[terms[k].s1] * vals[terms[k].s2] * wt[terms[k].wi];
  if (add)
    G[dst[e]] += s;
  else
    G[dst[e]] = s;
}

// The same for a system sharded over ranks: a rank computes only ITS tiles of the work block G (the plan's tile list);
// everything else in G is left over from earlier stages.  H is added only into the rank's own tiles (bit
// tr * ntc + tc of `owned`, 128 x 128 tiles): a "+=" on a tile nobody overwrites would grow from stage to stage and
// from factorisation to factorisation.
__global__ void k_st_add_h_owned(int nent, const long long *__restrict__ dst, const int *__restrict__ tptr,
                                 const HTerm *__restrict__ terms, const double *__restrict__ vals,
                                 const double *__restrict__ wt, double *__restrict__ G, long long ld,
                                 const unsigned *__restrict__ owned, int ntc) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nent) return;
  const long long d = dst[e];
  const int t = (int)(d / ld >> 7) * ntc + (int)(d % ld >> 7);
  if (!((owned[t >> 5] >> (t & 31)) & 1u)) return;
  double s = 0.0;
  for (int k = tptr[e]; k < tptr[e + 1]; k++) s += vals[terms[k].s1] * vals[terms[k].s2] * wt[terms[k].wi];
  G[d] += s;
}

// values of the A block scattered into dense storage: dst >= 0 offset into the F arena (dynamics
// rows), dst <= -2 offset -(dst + 2) into the misc arena (own equality rows of N), -1 not stored
// (the -1.0 of a dynamics row, the rows that fix x_0)
__global__ void k_st_scatter(long long nent, const long long *__restrict__ dst, const double *__restrict__ vals,
                             double *__restrict__ F, double *__restrict__ misc) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nent) return;
  const long long d = dst[e];
  if (d >= 0)
    F[d] = vals[e];
  else if (d <= -2)
    misc[-(d + 2)] = vals[e];
}
// check of the staircase: the last entry of every dynamics row is -1.0 (Hqp_IpLQDOCP::Get_Dim,
// hqp/Hqp_IpLQDOCP.C:214-215), the entries that fix x_0 are non-zero
__global__ void k_st_check(int nchk, const int *__restrict__ idx, const int *__restrict__ kind,
                           const double *__restrict__ vals, int *__restrict__ status) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nchk) return;
  const double v = vals[idx[e]];
  if (kind[e] == 0 ? v != -1.0 : v == 0.0) atomicExch(status, 6);
}

// ---------------------------------------------------------------------------------------
// block-wide arg-max of |value| with the smallest index winning ties (deterministic)
struct ArgMax {
  double v;
  int i;
};
__device__ __forceinline__ ArgMax better(ArgMax a, ArgMax b) { return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a; }
__device__ __forceinline__ ArgMax block_argmax(ArgMax a, ArgMax *red) {
  // wavefront: the maximum by DPP row operations, then the first lane that holds it (a fixed rule:
  // the thread -> entry map is fixed, so ties always go the same way)
  const double vmax = kktdev::wave_max_dpp(a.v);
  const unsigned long long hit = __ballot(a.v == vmax);
  const int src = hit ? __ffsll((long long)hit) - 1 : 0;
  a.v = vmax;
  a.i = __builtin_amdgcn_readlane(a.i, src);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  ArgMax r = red[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); w++)
    if (red[w].v > r.v) r = red[w];
  return r;
}

// In-place inverse of the q x q matrix a (LDS, leading dimension ld) by Gauss-Jordan
// elimination with complete pivoting.  Returns 0, or 1 when a pivot is exactly zero / NaN.
// ip, ir, ic: int work arrays of q entries; colv, rowv: double work arrays of q entries.
// 256 threads as a 16 x 16 grid: thread (ty, tx) owns the entries (ty + 16 i, tx + 16 j); the
// pivot search of step s+1 rides on the update sweep of step s.
#ifdef HQPKKT_STAMPS
__device__ int g_gj_stamps[32];
#define GSTAMP(slot)                                                                              \
  do {                                                                                            \
    if (threadIdx.x == 0 && s == 10) g_gj_stamps[slot] = (int)__builtin_amdgcn_s_memtime();       \
  } while (0)
#else
#define GSTAMP(slot)
#endif
#ifndef HQPKKT_GJ_RB
#define HQPKKT_GJ_RB 2  // rows per trip of the sweeps whose matrix is in global memory (1024 threads: 128 registers each)
#endif
template <int RB = 1>  // RB rows x 8 columns of loads in flight per thread
__device__ int gj_inverse(double *a, int q, int ld, int *ip, int *ir, int *ic, double *colv, double *rowv,
                          ArgMax *red) {
  const int tid = threadIdx.x, nt = blockDim.x;
  const int ty = tid >> 4, tx = tid & 15, RS = nt >> 4;  // (ty, tx): rows ty + RS i, columns tx + 16 j
  const double INF = __longlong_as_double(0x7ff0000000000000LL);
  const long long ABS = 0x7fffffffffffffffLL, INFB = 0x7ff0000000000000LL;
  for (int j = tid; j < q; j += nt) ip[j] = 0;
  __syncthreads();
  int bad = 0;
  // this thread's largest |entry| among the free rows and columns as its BIT PATTERN: for non-negative doubles the
  // order of the patterns is the order of the values, and a NaN lies above infinity - one integer comparison per
  // entry.  Ties: the first entry in this thread's fixed order, then the first lane, then the first wavefront - the
  // same pivot sequence in every run.
  long long bb = -1;
  int bi = 0x7fffffff;
  for (int r = ty; r < q; r += RS)
    for (int c = tx; c < q; c += 16) {
      const long long bits = __double_as_longlong(a[r * ld + c]) & ABS;
      if (bits > bb) bb = bits, bi = r * q + c;
    }
  for (int s = 0; s < q; s++) {
    GSTAMP(0);
    ArgMax best = block_argmax(ArgMax{bb < 0 ? -1.0 : (bb > INFB ? INF : __longlong_as_double(bb)), bi}, red);
    GSTAMP(1);
    const int irow = best.i / q, icol = best.i - irow * q;
    if (!(best.v > 0.0) || best.v == INF) bad = 1;
    if (irow != icol)
      for (int c = tid; c < q; c += nt) {
        const double x = a[irow * ld + c];
        a[irow * ld + c] = a[icol * ld + c];
        a[icol * ld + c] = x;
      }
    if (tid == 0) ip[icol] = 1, ir[s] = irow, ic[s] = icol;
    __syncthreads();
    GSTAMP(2);
    const double pinv = bad ? 1.0 : 1.0 / a[icol * ld + icol];
    for (int c = tid; c < q; c += nt) {
      colv[c] = (c == icol) ? 0.0 : a[c * ld + icol];
      rowv[c] = (c == icol ? 1.0 : a[icol * ld + c]) * pinv;
    }
    __syncthreads();
    // the pivot column is cleared and the pivot row scaled in place, so that the sweep has no special rows or columns:
    // every entry becomes a - colv[r] rowv[c] (colv is zero in the pivot row)
    for (int c = tid; c < q; c += nt) {
      if (c != icol) a[c * ld + icol] = 0.0;
      a[icol * ld + c] = rowv[c];
    }
    __syncthreads();
    GSTAMP(3);
    bb = -1, bi = 0x7fffffff;
    // RB x 8 entries per trip: what depends on the column alone (pivot row, flags) is read once per trip, the entries'
    // loads are issued together from clamped addresses (no branch between them) - the matrix may live in global memory
    // (stages with hundreds of controls), and in LDS too a wait per entry costs more than the arithmetic
    for (int c0 = tx; c0 < q; c0 += 128) {
      double rv[8];
      int cc[8];
      bool cfree[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int c = c0 + 16 * u;
        cc[u] = c < q ? c : q - 1;
        rv[u] = rowv[cc[u]];
        cfree[u] = c < q && ip[cc[u]] == 0;
      }
      for (int r0 = ty; r0 < q; r0 += RS * RB) {
        double x[RB][8];
#pragma unroll
        for (int k = 0; k < RB; k++) {
          const int r = min(r0 + k * RS, q - 1);
#pragma unroll
          for (int u = 0; u < 8; u++) x[k][u] = a[r * ld + cc[u]];
        }
#pragma unroll
        for (int k = 0; k < RB; k++) {
          const int r = r0 + k * RS, rr = min(r, q - 1);
          const double cr = colv[rr];
          const bool rfree = r < q && ip[rr] == 0;
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const int c = c0 + 16 * u;
            const double v = fma(-rv[u], cr, x[k][u]);
            if (r < q && c < q) a[r * ld + c] = v;
            const long long bits = __double_as_longlong(v) & ABS;
            if (rfree && cfree[u] && bits > bb) bb = bits, bi = r * q + c;
          }
        }
      }
    }
    GSTAMP(4);
  }
  __syncthreads();
  for (int s = q - 1; s >= 0; s--) {
    const int r1 = ir[s], c1 = ic[s];
    if (r1 != c1)
      for (int r = tid; r < q; r += nt) {
        const double x = a[r * ld + r1];
        a[r * ld + r1] = a[r * ld + c1];
        a[r * ld + c1] = x;
      }
    __syncthreads();
  }
  return bad;
}

// The same inverse for q <= 16 NB with the matrix in REGISTERS: thread (ty, tx) of the 16 x 16 grid
// holds the entries (ty + 16 i, tx + 16 j) of the augmented matrix [K | I] (NB x 2 NB doubles), so a
// step costs the broadcast of the pivot row and column through LDS and NB x 2 NB multiply-adds; no
// row or column is ever moved: after q steps the left half is a permutation, row r of the right half
// is row pcol(r) of the inverse.  a (LDS, q x q, ld) holds K on entry and the inverse on return.
template <int NB>
__device__ int gj_inverse_reg(double *a, int q, int ld, int *ip, int *pc_of_row, double *colv, double *rowv, ArgMax *red) {
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  const double INF = __longlong_as_double(0x7ff0000000000000LL);
  double m[NB][2 * NB];
  int rfree = 0, cfree = 0;  // bit i: own row ty + 16 i / own column tx + 16 i not yet a pivot row / column
#pragma unroll
  for (int i = 0; i < NB; i++) {
    const int r = ty + 16 * i;
    if (r < q) rfree |= 1 << i;
    if (tx + 16 * i < q) cfree |= 1 << i;
#pragma unroll
    for (int j = 0; j < NB; j++) {
      const int c = tx + 16 * j;
      m[i][j] = (r < q && c < q) ? a[r * ld + c] : 0.0;
      m[i][NB + j] = (r == c && r < q) ? 1.0 : 0.0;
    }
  }
  __syncthreads();
  int bad = 0;
  for (int s = 0; s < q; s++) {
    ArgMax best{-1.0, 0x7fffffff};
#pragma unroll
    for (int i = 0; i < NB; i++)
#pragma unroll
      for (int j = 0; j < NB; j++)
        if ((rfree >> i & 1) && (cfree >> j & 1)) {
          const double v = fabs(m[i][j]);
          best = better(best, ArgMax{v == v ? v : INF, (ty + 16 * i) * q + tx + 16 * j});
        }
    best = block_argmax(best, red);
    const int pr = best.i / q, pc = best.i - pr * q;
    if (!(best.v > 0.0) || best.v == INF) bad = 1;
    // owners publish the pivot row (2 q values) and the pivot column
    if ((pr & 15) == ty) {
      const int i = pr >> 4;
#pragma unroll
      for (int ii = 0; ii < NB; ii++)
        if (ii == i) {
#pragma unroll
          for (int j = 0; j < 2 * NB; j++) rowv[tx + 16 * j] = m[ii][j];
        }
    }
    if ((pc & 15) == tx) {
      const int j = pc >> 4;
#pragma unroll
      for (int jj = 0; jj < NB; jj++)
        if (jj == j) {
#pragma unroll
          for (int i = 0; i < NB; i++) colv[ty + 16 * i] = m[i][jj];
        }
    }
    if (tid == 0) pc_of_row[pr] = pc;
    __syncthreads();
    const double piv = rowv[(pc & 15) + 16 * (pc >> 4)];
    const double pinv = bad ? 1.0 : 1.0 / piv;
    double rv[2 * NB];
#pragma unroll
    for (int j = 0; j < 2 * NB; j++) rv[j] = rowv[tx + 16 * j] * pinv;
#pragma unroll
    for (int i = 0; i < NB; i++) {
      const int r = ty + 16 * i;
      const double f = colv[r];
      if (r == pr) {
#pragma unroll
        for (int j = 0; j < 2 * NB; j++) m[i][j] = rv[j];
      } else {
#pragma unroll
        for (int j = 0; j < 2 * NB; j++) m[i][j] -= f * rv[j];
      }
    }
    if ((pr & 15) == ty) rfree &= ~(1 << (pr >> 4));
    if ((pc & 15) == tx) cfree &= ~(1 << (pc >> 4));
    __syncthreads();  // rowv / colv are rewritten in the next step
  }
  // row r of the right half is row pc_of_row[r] of the inverse
#pragma unroll
  for (int i = 0; i < NB; i++) {
    const int r = ty + 16 * i;
    if (r < q) {
      const int dst = pc_of_row[r];
#pragma unroll
      for (int j = 0; j < NB; j++) {
        const int c = tx + 16 * j;
        if (c < q) a[dst * ld + c] = m[i][NB + j];
      }
    }
  }
  __syncthreads();
  (void)ip;
  return bad;
}
// The same for a matrix that is positive definite after its scaling (K = G_uu of a stage without consumed
// constraint rows): pivots in the natural order down the diagonal - no search, two barriers per step instead of
// four.  Elimination without pivoting is backward stable on such a matrix.  Returns 2 without having touched `a`
// when a pivot is not safely positive (G_uu indefinite or nearly singular): the caller then runs the search.
template <int NB>
__device__ int gj_inverse_reg_spd(double *a, int q, int ld, double *colv, double *rowv) {
  // in place (no identity beside the matrix: half the multiply-adds and LDS reads of the form with the search): step s
  // replaces the pivot by 1/p, its row by row/p, its column by -column/p and every other entry by a_ij - a_is a_sj / p
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  double m[NB][NB];
#pragma unroll
  for (int i = 0; i < NB; i++) {
    const int r = ty + 16 * i;
#pragma unroll
    for (int j = 0; j < NB; j++) {
      const int c = tx + 16 * j;
      m[i][j] = (r < q && c < q) ? a[r * ld + c] : (r == c ? 1.0 : 0.0);
    }
  }
  __syncthreads();
  int bad = 0;
  for (int s = 0; s < q; s++) {
    if ((s & 15) == ty) {
      const int i = s >> 4;
#pragma unroll
      for (int ii = 0; ii < NB; ii++)
        if (ii == i) {
#pragma unroll
          for (int j = 0; j < NB; j++) rowv[tx + 16 * j] = m[ii][j];
        }
    }
    if ((s & 15) == tx) {
      const int j = s >> 4;
#pragma unroll
      for (int jj = 0; jj < NB; jj++)
        if (jj == j) {
#pragma unroll
          for (int i = 0; i < NB; i++) colv[ty + 16 * i] = m[i][jj];
        }
    }
    __syncthreads();
    const double piv = rowv[s];
    if (!(piv > 1e-10)) bad = 2;  // uniform: every thread reads the same value (the matrix is scaled: diagonal <= 1)
    const double pinv = bad ? 1.0 : 1.0 / piv;
    double rv[NB];
#pragma unroll
    for (int j = 0; j < NB; j++) rv[j] = (tx + 16 * j == s ? 1.0 : rowv[tx + 16 * j]) * pinv;
#pragma unroll
    for (int i = 0; i < NB; i++) {
      const int r = ty + 16 * i;
      const double f = colv[r];
#pragma unroll
      for (int j = 0; j < NB; j++) {
        const double old = tx + 16 * j == s ? 0.0 : m[i][j];
        m[i][j] = r == s ? rv[j] : fma(-f, rv[j], old);
      }
    }
    __syncthreads();  // rowv / colv are rewritten in the next step
  }
  if (bad) return bad;  // (uniform) `a` still holds the matrix
#pragma unroll
  for (int i = 0; i < NB; i++) {
    const int r = ty + 16 * i;
    if (r < q) {
#pragma unroll
      for (int j = 0; j < NB; j++) {
        const int c = tx + 16 * j;
        if (c < q) a[r * ld + c] = m[i][j];
      }
    }
  }
  __syncthreads();
  return 0;
}
// The inverse of a matrix that is positive definite after its scaling, in place in LDS (order 65 .. 136, sixteen
// wavefronts): pivots down the diagonal, no search - per step one broadcast of the pivot row / column and the sweep.
// Returns 2 at the first pivot that is not safely positive; the matrix is then partly eliminated (the caller builds
// it again and runs the search).
__device__ int gj_inverse_spd(double *a, int q, int ld, double *colv, double *rowv) {
  const int tid = threadIdx.x, nt = blockDim.x;
  const int ty = tid >> 4, tx = tid & 15, RS = nt >> 4;
  for (int s = 0; s < q; s++) {
    const double piv = a[s * ld + s];
    if (!(piv > 1e-10)) return 2;  // (uniform: every thread reads the same entry)
    const double pinv = 1.0 / piv;
    for (int c = tid; c < q; c += nt) {
      colv[c] = (c == s) ? 0.0 : a[c * ld + s];
      rowv[c] = (c == s ? 1.0 : a[s * ld + c]) * pinv;
    }
    __syncthreads();
    for (int c = tid; c < q; c += nt) {
      if (c != s) a[c * ld + s] = 0.0;
      a[s * ld + c] = rowv[c];
    }
    __syncthreads();
    for (int c0 = tx; c0 < q; c0 += 128) {
      double rv[8];
      int cc[8];
#pragma unroll
      for (int u = 0; u < 8; u++) cc[u] = min(c0 + 16 * u, q - 1), rv[u] = rowv[cc[u]];
      for (int r = ty; r < q; r += RS) {
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) x[u] = a[r * ld + cc[u]];
        const double cr = colv[r];
#pragma unroll
        for (int u = 0; u < 8; u++)
          if (c0 + 16 * u < q) a[r * ld + c0 + 16 * u] = fma(-rv[u], cr, x[u]);
      }
    }
    __syncthreads();
  }
  return 0;
}
// dispatch: registers up to order 64 (the register forms are laid out for 256 threads), LDS / global memory above
template <int NT, bool BIG = (NT != 256)>
__device__ int gj_inverse_any(double *a, int q, int ld, int *ip, int *ir, int *ic, double *colv, double *rowv, ArgMax *red,
                              bool spd = false) {
  if constexpr (NT != 256) return gj_inverse<BIG ? HQPKKT_GJ_RB : 1>(a, q, ld, ip, ir, ic, colv, rowv, red);
  if (spd && q <= 64) {
    const int e = q <= 16 ? gj_inverse_reg_spd<1>(a, q, ld, colv, rowv)
                : q <= 32 ? gj_inverse_reg_spd<2>(a, q, ld, colv, rowv) : gj_inverse_reg_spd<4>(a, q, ld, colv, rowv);
    if (e == 0) return 0;
  }
  if (q <= 16) return gj_inverse_reg<1>(a, q, ld, ip, ir, colv, rowv, red);
  if (q <= 32) return gj_inverse_reg<2>(a, q, ld, ip, ir, colv, rowv, red);
  if (q <= 64) return gj_inverse_reg<4>(a, q, ld, ip, ir, colv, rowv, red);
  return gj_inverse(a, q, ld, ip, ir, ic, colv, rowv, red);
}

// LU factors of the q x q matrix a (LDS or global memory, leading dimension ld) in place - unit lower L below the
// diagonal, U on and above it - with complete pivoting; rows and columns are exchanged physically, pr[s] / pc[s]
// record the row / column that came to position s.  Returns 0, or 1 when a pivot is exactly zero / NaN.  The pivot
// search of step s+1 rides on the update sweep of step s (thread grid 16 columns x nt/16 rows, as in gj_inverse).
// Used where the factors are applied by substitution (k_st_x0_free): a solve by factors leaves a residual of
// eps |K| |x| where the product with an explicit inverse leaves cond(K) eps |b|.
template <int RB = 1>
__device__ int lu_complete(double *a, int q, int ld, int *pr, int *pc, double *colv, double *rowv, ArgMax *red) {
  const int tid = threadIdx.x, nt = blockDim.x;
  const int ty = tid >> 4, tx = tid & 15, RS = nt >> 4;
  const double INF = __longlong_as_double(0x7ff0000000000000LL);
  int bad = 0;
  ArgMax best{-1.0, 0x7fffffff};
  for (int r = ty; r < q; r += RS)
    for (int c = tx; c < q; c += 16) {
      const double v = fabs(a[r * ld + c]);
      best = better(best, ArgMax{v == v ? v : INF, r * q + c});
    }
  for (int s = 0; s < q; s++) {
    best = block_argmax(best, red);
    const int irow = best.i / q, icol = best.i - irow * q;
    if (!(best.v > 0.0) || best.v == INF) {
      bad = 1;
      break;
    }
    if (irow != s)
      for (int c = tid; c < q; c += nt) {
        const double x = a[irow * ld + c];
        a[irow * ld + c] = a[s * ld + c];
        a[s * ld + c] = x;
      }
    __syncthreads();
    if (icol != s)
      for (int r = tid; r < q; r += nt) {
        const double x = a[r * ld + icol];
        a[r * ld + icol] = a[r * ld + s];
        a[r * ld + s] = x;
      }
    if (tid == 0) pr[s] = irow, pc[s] = icol;
    __syncthreads();
    const double pinv = 1.0 / a[s * ld + s];
    for (int c = s + 1 + tid; c < q; c += nt) {
      colv[c] = a[c * ld + s] * pinv;
      rowv[c] = a[s * ld + c];
    }
    __syncthreads();
    best = ArgMax{-1.0, 0x7fffffff};
    for (int c0 = s + 1 + tx; c0 < q; c0 += 128) {  // (as in gj_inverse: column data once per trip, RB x 8 loads in flight)
      double rv[8];
      int cc[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        cc[u] = min(c0 + 16 * u, q - 1);
        rv[u] = rowv[cc[u]];
      }
      for (int r0 = s + 1 + ty; r0 < q; r0 += RS * RB) {
        double x[RB][8];
#pragma unroll
        for (int k = 0; k < RB; k++) {
          const int r = min(r0 + k * RS, q - 1);
#pragma unroll
          for (int u = 0; u < 8; u++) x[k][u] = a[r * ld + cc[u]];
        }
#pragma unroll
        for (int k = 0; k < RB; k++) {
          const int r = r0 + k * RS;
          const double cr = colv[min(r, q - 1)];
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const int c = c0 + 16 * u;
            const double v = x[k][u] - cr * rv[u];
            if (r < q && c < q) {
              a[r * ld + c] = v;
              const double av = fabs(v);
              best = better(best, ArgMax{av == av ? av : INF, r * q + c});
            }
          }
        }
      }
    }
    for (int r = s + 1 + tid; r < q; r += nt) a[r * ld + s] = colv[r];  // the multipliers
  }
  __syncthreads();
  return bad;
}

// Per stage, one workgroup: (A) rank-revealing elimination of the control part N_u of the
// stage's constraint rows (own equalities, then the rows carried back from stage k+1) with
// complete pivoting: consumed rows R (they determine controls), leftover rows L (free of u
// after subtracting t times the consumed rows: carried on to stage k-1) - the job of GE_QP
// in the reference (meschach/addon_hqp.c:399-475, QR with column pivoting there);
// (B) K = [G_uu N_uR'; N_uR 0], symmetric scaling, explicit inverse (the reference factors
// Z' G_uu Z by Bunch-Kaufman and forms Z (Z'G_uu Z)^-1 Z' explicitly, :1907-1924).
struct SmallArgs {
  const double *G;
  long long ldg;
  int n, m;             // states, controls of the stage
  const double *N;
  long long ldn;        // constraint rows, capn x (n + m)
  int e;                // own rows
  const int *cnt_next;  // carried rows coming in (device), nullptr: none
  int capn, cap, qmax;
  double ge_tol;
  double *Kinv;
  long long ldq;        // qmax x qmax, zero padded
  double *Kmat;         // K itself (same shape): the products with K^-1 are refined against it
  double *t;
  long long ldt;        // cap x ldt : t[li][s]
  int *dyn;             // [0] r, [1] number of leftover rows, [2..2+capn) R, [2+capn..2+2capn) L
  int *status;          // set to 4 (E_SING) on a singular K
  double *scratch;      // null: the matrices of (A) and (B) live in LDS; else in this global area (stages with hundreds of
                        // controls / carried rows: StagedPlan::big) and LDS holds the flags and vectors only
  int mode;             // (LDS form: 100 = test hook, see k_st_small)  global-memory form:  0: everything; 1: (A) and the scaled K, no inverse (the blocked
                        // elimination k_blk_* follows); 2: the inverse by this kernel only if the blocked one gave up
};
// layout of SmallArgs::scratch in the global-memory form (qmax = SmallArgs::qmax, ldk = up8(qmax)):
// scaled K | its scaling | panel T0 (64 x ldk) | panel R (64 x ldk) | block P (64 x 64) | flags (ints)
struct BigScratch {
  double *Ks, *dsc, *T0, *R, *Pb;
  int *flags;
  long long ldk;
};
__host__ __device__ inline BigScratch big_scratch(double *base, int qmax) {
  BigScratch b;
  b.ldk = (qmax + 7) / 8 * 8;
  b.Ks = base, b.dsc = b.Ks + (long long)qmax * b.ldk, b.T0 = b.dsc + b.ldk, b.R = b.T0 + 64 * b.ldk, b.Pb = b.R + 64 * b.ldk;
  b.flags = (int *)(b.Pb + 64 * 64);
  return b;
}
#ifdef HQPKKT_STAMPS
#define SSTAMP(slot)                                                                          \
  do {                                                                                        \
    if (threadIdx.x == 0) a.status[9 + (slot)] = (int)__builtin_amdgcn_s_memtime();           \
  } while (0)
#else
#define SSTAMP(slot)
#endif
// <256, false>: K of order <= 64, register-resident inverse; <1024, false>: up to what LDS holds (order ~136), sixteen
// wavefronts on the sweeps (measured: the sweep of the LDS inverse is bound by instruction issue, not by LDS);
// <1024, true>: matrices in global memory.
template <int NT, bool BIG = (NT != 256)>
__global__ void __launch_bounds__(NT, NT / 256) k_st_small(SmallArgs a) {  // (one workgroup: no occupancy to protect, all registers)
  extern __shared__ __attribute__((aligned(16))) double sm[];
  __shared__ ArgMax red[16];
  __shared__ int s_r, s_stop;
  const int tid = threadIdx.x, nt = blockDim.x;
  // BIG: the matrices of (A) and (B) in a.scratch (global memory) instead of LDS
  SSTAMP(0);
  const int m = a.m, n = a.n;
  const int c = a.e + (a.cnt_next ? *a.cnt_next : 0);
  int *Rl = a.dyn + 2, *Ll = a.dyn + 2 + a.capn;
  // ---------------- (A)
  int r = 0;
  if (BIG && a.mode == 2) {
    // (second call for this stage: the ranks stand in dyn, and the area of (A) now holds K's scaling and the flags)
    if (tid == 0) atomicAdd(a.status + 6, 1);
    if (big_scratch(a.scratch, a.qmax).flags[0] == 0) return;  // (uniform)
    if (tid == 0) atomicAdd(a.status + 7, 1);
    r = a.dyn[0];
  } else if (c > 0 && m > 0) {
    const int ld = m + c;
    // (the kind of memory is fixed by the instantiation, so that the accesses are LDS / global instructions and not
    // flat ones: with a pointer chosen at run time every load also waits for the stores in front of it)
    double *aug;
    int *rfree;
    if constexpr (BIG)
      aug = a.scratch, rfree = (int *)sm;
    else
      aug = sm, rfree = (int *)(sm + (size_t)c * ld);
    int *cfree = rfree + c;
    for (int e = tid; e < c * ld; e += nt) {
      const int i = e / ld, j = e - i * ld;
      aug[e] = j < m ? a.N[(long long)i * a.ldn + n + j] : (j - m == i ? 1.0 : 0.0);
    }
    for (int i = tid; i < c; i += nt) rfree[i] = 1;
    for (int j = tid; j < m; j += nt) cfree[j] = 1;
    if (tid == 0) s_r = 0, s_stop = 0;
    __syncthreads();
    const int steps = c < m ? c : m;
    for (int s = 0; s < steps; s++) {
      ArgMax best{-1.0, 0x7fffffff};
      for (int e = tid; e < c * m; e += nt) {
        const int i = e / m, j = e - i * m;
        if (rfree[i] && cfree[j]) {
          const double v = fabs(aug[i * ld + j]);
          best = better(best, ArgMax{v, e});
        }
      }
      best = block_argmax(best, red);
      if (!(best.v > a.ge_tol)) break;  // uniform
      const int pi = best.i / m, pj = best.i - pi * m;
      __syncthreads();
      if (tid == 0) rfree[pi] = 0, cfree[pj] = 0, Rl[s] = pi, s_r = s + 1;
      const double piv = aug[pi * ld + pj];
      // multipliers of the free rows (column pj), then the update
      double *fac = (double *)(cfree + m + ((c + m) & 1));  // c doubles behind the flags, 8-byte aligned
      __syncthreads();
      for (int i = tid; i < c; i += nt) fac[i] = (rfree[i]) ? aug[i * ld + pj] / piv : 0.0;
      __syncthreads();
      {  // (thread grid 16 columns x nt/16 rows; the pivot row's entries once per trip, eight loads in flight)
        const int ty = tid >> 4, tx = tid & 15, RS = nt >> 4;
        for (int j0 = tx; j0 < ld; j0 += 128) {
          double pv[8];
          int jj[8];
#pragma unroll
          for (int u = 0; u < 8; u++) jj[u] = min(j0 + 16 * u, ld - 1), pv[u] = aug[pi * ld + jj[u]];
          for (int i = ty; i < c; i += RS) {
            const double f = fac[i];  // (zero for the rows that are not free)
            if (f != 0.0) {
              double x[8];
#pragma unroll
              for (int u = 0; u < 8; u++) x[u] = aug[i * ld + jj[u]];
#pragma unroll
              for (int u = 0; u < 8; u++) {
                const int j = j0 + 16 * u;
                if (j < ld) aug[i * ld + j] = (j == pj) ? 0.0 : x[u] - f * pv[u];
              }
            }
          }
        }
      }
      __syncthreads();
    }
    __syncthreads();
    r = s_r;
    // leftover rows in increasing order, t = -coef[L, R]
    if (tid == 0) {
      int nl = 0;
      for (int i = 0; i < c; i++)
        if (rfree[i]) Ll[nl++] = i;
      if (nl > a.cap) nl = a.cap, atomicExch(a.status, 1);  // more rows to carry than the plan holds
      a.dyn[0] = r, a.dyn[1] = nl;
    }
    __syncthreads();
    const int nl = min(c - r, a.cap);
    for (int e = tid; e < nl * r; e += nt) {
      const int li = e / r, s = e - li * r;
      a.t[(long long)li * a.ldt + s] = -aug[Ll[li] * ld + m + Rl[s]];
    }
    __syncthreads();
  } else {
    if (tid == 0) {
      int nl = c;
      if (nl > a.cap) nl = a.cap, atomicExch(a.status, 1);
      a.dyn[0] = 0, a.dyn[1] = nl;
      for (int i = 0; i < nl; i++) Ll[i] = i;
    }
    __syncthreads();
  }
  // ---------------- (B)
  SSTAMP(1);
  const int q = m + r;
  if (q > 0) {
    int ld = q | 1;
    double *Km, *dsc, *colv;
    if constexpr (BIG) {
      const BigScratch bs = big_scratch(a.scratch, a.qmax);
      Km = bs.Ks, dsc = bs.dsc, ld = (int)bs.ldk, colv = sm;
    } else
      Km = sm, dsc = sm + (size_t)q * ld, colv = dsc + q;
    double *rowv = colv + (q > 64 ? q : 64);
    int *ip = (int *)(rowv + (q > 128 ? q : 128)), *ir = ip + q, *ic = ir + q;
    if (BIG && a.mode == 2) {
      // the blocked elimination gave up (a diagonal block without a safe pivot, or its result failed the check
      // against K): K is scaled again from its copy and inverted here, with the search over the whole matrix
      const BigScratch bs = big_scratch(a.scratch, a.qmax);
      if (bs.flags[0] == 0) return;  // (uniform)
      for (int e = tid; e < q * q; e += nt) {
        const int i = e / q, j = e - i * q;
        Km[i * ld + j] = a.Kmat[(long long)i * a.ldq + j] * dsc[i] * dsc[j];
      }
      __syncthreads();
    } else {
    for (int e = tid; e < q * q; e += nt) {
      const int i = e / q, j = e - i * q;
      double v;
      if (i < m && j < m) {
        const int hi = i > j ? i : j, lo = i > j ? j : i;
        v = a.G[(long long)(n + hi) * a.ldg + n + lo];
      } else if (i >= m && j >= m)
        v = 0.0;
      else {
        const int s = (i >= m ? i : j) - m, u = i >= m ? j : i;
        v = a.N[(long long)Rl[s] * a.ldn + n + u];
      }
      Km[i * ld + j] = v;
    }
    __syncthreads();
    SSTAMP(2);
    for (int e = tid; e < a.qmax * a.qmax; e += nt) {
      const int i = e / a.qmax, j = e - i * a.qmax;
      a.Kmat[(long long)i * a.ldq + j] = (i < q && j < q) ? Km[i * ld + j] : 0.0;
    }
    SSTAMP(3);
    // scaling: u rows 1/sqrt(K_ii) where K_ii > 1 (hqp/Hqp_IpLQDOCP.C:1851-1858), constraint rows
    // by their largest entry
    for (int i = tid; i < q; i += nt) {
      double d = 1.0;
      if (i < m) {
        const double kii = Km[i * ld + i];
        if (kii > 1.0) d = 1.0 / sqrt(kii);
      } else {
        double mx = 0.0;
        for (int j = 0; j < m; j++) mx = fmax(mx, fabs(Km[i * ld + j]));
        if (mx > 0.0) d = 1.0 / mx;
      }
      dsc[i] = d;
    }
    __syncthreads();
    for (int e = tid; e < q * q; e += nt) {
      const int i = e / q, j = e - i * q;
      Km[i * ld + j] *= dsc[i] * dsc[j];
    }
    __syncthreads();
    if (BIG && a.mode == 1) {
      // identity padding up to qmax (the ranks are decided here, the launches behind this one were sized by the host)
      const BigScratch bs = big_scratch(a.scratch, a.qmax);
      for (int e = tid; e < a.qmax * a.qmax; e += nt) {
        const int i = e / a.qmax, j = e - i * a.qmax;
        if (i >= q || j >= q) Km[i * ld + j] = i == j ? 1.0 : 0.0;
      }
      for (int i = q + tid; i < a.qmax; i += nt) dsc[i] = 0.0;
      if (tid == 0) bs.flags[0] = 0;
      return;
    }
    }
    SSTAMP(4);
    int bad = -1;
    if constexpr (NT == 1024 && !BIG) {
      if (r == 0) {  // no consumed constraint rows: K = G_uu, positive definite unless the problem is not convex in u
        bad = gj_inverse_spd(Km, q, ld, colv, rowv);
        if (a.mode == 100) bad = 2;  // (test hook, HQPKKT_SPD_TEST_FAIL: the way back from a pivot that was refused)
        if (bad) {  // (uniform) scaled again from the copy, then with the search
          __syncthreads();
          for (int e = tid; e < q * q; e += nt) {
            const int i = e / q, j = e - i * q;
            Km[i * ld + j] = a.Kmat[(long long)i * a.ldq + j] * dsc[i] * dsc[j];
          }
          __syncthreads();
          bad = -1;
        }
      }
    }
    if (bad < 0) bad = gj_inverse_any<NT, BIG>(Km, q, ld, ip, ir, ic, colv, rowv, red, r == 0);
    if (bad && tid == 0) atomicExch(a.status, 4);
    __syncthreads();
    SSTAMP(5);
    for (int e = tid; e < a.qmax * a.qmax; e += nt) {
      const int i = e / a.qmax, j = e - i * a.qmax;
      double v = 0.0;
      if (i < q && j < q) v = 0.5 * (Km[i * ld + j] + Km[j * ld + i]) * dsc[i] * dsc[j];
      a.Kinv[(long long)i * a.ldq + j] = v;
    }
    SSTAMP(6);
  } else {
    if (BIG && a.mode == 2) return;
    for (int e = tid; e < a.qmax * a.qmax; e += nt)
      a.Kinv[(long long)(e / a.qmax) * a.ldq + e % a.qmax] = 0.0, a.Kmat[(long long)(e / a.qmax) * a.ldq + e % a.qmax] = 0.0;
    if constexpr (BIG) {
      if (a.mode == 1) {
        const BigScratch bs = big_scratch(a.scratch, a.qmax);
        for (int e = tid; e < a.qmax * a.qmax; e += nt) bs.Ks[(long long)(e / a.qmax) * bs.ldk + e % a.qmax] = e / a.qmax == e % a.qmax ? 1.0 : 0.0;
        for (int i = tid; i < a.qmax; i += nt) bs.dsc[i] = 0.0;
        if (tid == 0) bs.flags[0] = 0;
      }
    }
  }
}

// ---- Blocked elimination of a stage's K where it does not fit the LDS of one CU (order up to 768): the symmetric
// sweep operator with pivot BLOCKS of 64 down the diagonal - per block one workgroup inverts the 64 x 64 diagonal block
// (complete pivoting inside it, registers), two small products on the whole chip form the block row P K_j: and the
// update K - K_:j P K_j:, one kernel puts the block row, its mirror image and -P in place.  After the last block the
// area holds -K^-1.  No pivoting ACROSS blocks: that is safe where the control Hessian of the stage is positive definite
// and the consumed constraint rows come behind it (the quasidefinite order); a diagonal block without a safe pivot, or a
// result that fails the check K K^-1 = I, raises flags[0] and the one-workgroup elimination with the search over the
// whole matrix runs instead (k_st_small mode 2).  200 controls: 3 ms -> ~0.5 ms per stage, 512: 34 -> ~1 ms.
struct BlkArgs {
  double *scratch;
  int qmax, j;
};
__global__ void __launch_bounds__(256) k_blk_pivot(BlkArgs a) {
  __shared__ double A[64 * 65], colv[64], rowv[128];
  __shared__ int ip[64], pcr[64];
  __shared__ ArgMax red[16];
  const BigScratch bs = big_scratch(a.scratch, a.qmax);
  if (bs.flags[0]) return;  // (uniform) an earlier block gave up
  const int tid = threadIdx.x, j0 = a.j * 64, b = min(64, a.qmax - j0);
  ArgMax am{0.0, 0};
  for (int e = tid; e < 64 * 64; e += 256) {
    const int r = e >> 6, c = e & 63;
    const double v = (r < b && c < b) ? bs.Ks[(long long)(j0 + r) * bs.ldk + j0 + c] : 0.0;
    A[r * 65 + c] = v;
    const double av = fabs(v);
    if (!(av <= am.v)) am.v = av == av ? av : __longlong_as_double(0x7ff0000000000000LL);
  }
  for (int e = tid; e < 64 * a.qmax; e += 256) {
    const int r = e / a.qmax, c = e - r * a.qmax;
    bs.T0[(long long)r * bs.ldk + c] = r < b ? bs.Ks[(long long)(j0 + r) * bs.ldk + c] : 0.0;
  }
  am = block_argmax(am, red);
  __syncthreads();
  // (a positive definite block - the control Hessian's blocks and their Schur complements - down its diagonal first;
  // the register form leaves A untouched when it refuses a pivot, the search takes over)
  int bad = gj_inverse_reg_spd<4>(A, b, 65, colv, rowv);
  if (bad) bad = gj_inverse_reg<4>(A, b, 65, ip, pcr, colv, rowv, red);
  ArgMax pm{0.0, 0};
  for (int e = tid; e < 64 * 64; e += 256) {
    const int r = e >> 6, c = e & 63;
    const double v = (r < b && c < b) ? 0.5 * (A[r * 65 + c] + A[c * 65 + r]) : 0.0;
    bs.Pb[e] = v;
    const double av = fabs(v);
    if (!(av <= pm.v)) pm.v = av == av ? av : __longlong_as_double(0x7ff0000000000000LL);
  }
  pm = block_argmax(pm, red);
  // |K_jj| |K_jj^-1| beyond 1e12: no safe pivot inside this block
  if (tid == 0 && (bad || !(am.v * pm.v < 1e12))) {
    bs.flags[0] = 1;
    // (introspection: the block that gave up, and |K_jj|, |K_jj^-1| as floats)
    bs.flags[1] = a.j + 1, bs.flags[2] = __float_as_int((float)am.v), bs.flags[3] = __float_as_int((float)pm.v);
  }
}
__global__ void k_blk_fixup(BlkArgs a) {
  const BigScratch bs = big_scratch(a.scratch, a.qmax);
  if (bs.flags[0]) return;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= 64 * a.qmax) return;
  const int r = e / a.qmax, c = e - r * a.qmax, j0 = a.j * 64, b = min(64, a.qmax - j0);
  if (r >= b) return;
  if (c >= j0 && c < j0 + b) {
    bs.Ks[(long long)(j0 + r) * bs.ldk + c] = -bs.Pb[r * 64 + c - j0];
  } else {
    const double v = bs.R[(long long)r * bs.ldk + c];
    bs.Ks[(long long)(j0 + r) * bs.ldk + c] = v;
    bs.Ks[(long long)c * bs.ldk + j0 + r] = v;
  }
}
// Kinv = -(swept area), unscaled, symmetrised, zero beyond the live order q = m + r
__global__ void k_blk_final(SmallArgs a) {
  const BigScratch bs = big_scratch(a.scratch, a.qmax);
  if (bs.flags[0]) return;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= a.qmax * a.qmax) return;
  const int i = e / a.qmax, l = e - i * a.qmax, q = a.m + a.dyn[0];
  double v = 0.0;
  if (i < q && l < q) v = -0.5 * (bs.Ks[(long long)i * bs.ldk + l] + bs.Ks[(long long)l * bs.ldk + i]) * bs.dsc[i] * bs.dsc[l];
  a.Kinv[(long long)i * a.ldq + l] = v;
}
// E = K Kinv (in the area of the swept matrix) against the identity of order q
__global__ void __launch_bounds__(1024) k_blk_check(SmallArgs a, double tol) {
  __shared__ ArgMax red[16];
  const BigScratch bs = big_scratch(a.scratch, a.qmax);
  if (bs.flags[0]) return;
  const int q = a.m + a.dyn[0];
  ArgMax am{0.0, 0};
  for (int e = threadIdx.x; e < a.qmax * a.qmax; e += blockDim.x) {
    const int i = e / a.qmax, l = e - i * a.qmax;
    const double d = fabs(bs.Ks[(long long)i * bs.ldk + l] - ((i == l && i < q) ? 1.0 : 0.0));
    if (!(d <= am.v)) am.v = d == d ? d : __longlong_as_double(0x7ff0000000000000LL);
  }
  am = block_argmax(am, red);
  if (threadIdx.x == 0 && !(am.v <= tol)) bs.flags[0] = 1;
}

static size_t st_small_lds(int m, int capn, bool big = false) {
  const size_t q = (size_t)m + (size_t)(capn < m ? capn : m);
  // big: the matrices are in global memory, LDS holds flags, multipliers, scalings, pivot row / column, index arrays
  const size_t a = (big ? 0 : (size_t)capn * (m + capn) * 8) + (size_t)(capn + m + 2) * 4 + (size_t)capn * 8 + 16;
  const size_t b = big ? (size_t)kktdev::gj_lds_bytes((long long)q) - q * (q | 1) * 8 : (size_t)kktdev::gj_lds_bytes((long long)q);
  return (a > b ? a : b) + 64;
}

// Per stage, one thread per state column j: Y = [G_ux ; N_xR ; 0] and the transposed carried
// rows BT[j][li] = (N_L - t N_R)[li][j] (zero beyond the live count)
struct WideArgs {
  const double *G;
  long long ldg;
  int n, m;
  const double *N;
  long long ldn;
  int capn, cap, qmax;
  const double *t;
  long long ldt;
  const int *dyn;
  double *Y;
  long long ldy;
  double *BT;
  long long ldb;
};
__global__ void k_st_wide(WideArgs a) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= a.n) return;
  const int r = a.dyn[0], nl = a.dyn[1];
  const int *Rl = a.dyn + 2, *Ll = a.dyn + 2 + a.capn;
  for (int i = 0; i < a.m; i++) a.Y[(long long)i * a.ldy + j] = a.G[(long long)(a.n + i) * a.ldg + j];
  for (int s = 0; s < a.qmax - a.m; s++)
    a.Y[(long long)(a.m + s) * a.ldy + j] = s < r ? a.N[(long long)Rl[s] * a.ldn + j] : 0.0;
  for (int li = 0; li < a.cap; li++) {
    double v = 0.0;
    if (li < nl) {
      v = a.N[(long long)Ll[li] * a.ldn + j];
      for (int s = 0; s < r; s++) v -= a.t[(long long)li * a.ldt + s] * a.N[(long long)Rl[s] * a.ldn + j];
    }
    a.BT[(long long)j * a.ldb + li] = v;
  }
}
// Rm = K^-1 Y with one round of refinement against K (Rm += K^-1 (Y - K Rm), see staged_run_factor) for stages whose K
// has order <= 64, in ONE launch instead of three products with a 50-deep k loop: a workgroup takes 32 columns of Y;
// K^-1 and K (symmetric, so a row is read as a column: broadcast reads) and the three q x 32 panels live in LDS;
// thread (row group g of 8, column c) computes the rows g, g + 8, ... of its column.
struct RmArgs {
  const double *Kinv, *Kmat;
  long long ldq;
  double *Y;
  double *Rm;
  long long ldy;
  int q, n;
  int wide;  // 1: Y and the carried rows B_k are formed here too, from w (k_st_wide's work: one launch less)
  WideArgs w;
};
static const int RM_COLS = 32;
// K^-1 and K zero-padded to a multiple of 16 (rows of QP + 1 doubles), three QP x 32 panels
static inline size_t st_rm_lds(int q) {
  const size_t QP = ((size_t)q + 15) & ~(size_t)15;
  return sizeof(double) * (2 * QP * (QP + 1) + 3 * QP * RM_COLS);
}
#ifdef HQPKKT_STAMPS
#define RSTAMP(slot)                                                                                     \
  do {                                                                                                   \
    if (threadIdx.x == 0 && blockIdx.x == 0) g_gj_stamps[16 + (slot)] = (int)__builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define RSTAMP(slot)
#endif
// The three products are (q x q)' (q x 32) each: on the matrix pipe (v_mfma_f64_16x16x4: wavefront w takes the rows
// 16 w .. 16 w + 15 of the result, both halves of the 32 columns) - as scalar sums out of LDS they had been bound by the
// LDS issue rate (nine reads per eight multiply-adds; 14 us of a 22-us kernel at q = 50, tools/stamps_stsmall.py).
__global__ void __launch_bounds__(256) k_st_rm(RmArgs a) {
  typedef double d4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) double sm[];
  RSTAMP(0);
  const int q = a.q, QP = (q + 15) & ~15, LDK = QP + 1, tid = threadIdx.x, c = tid & (RM_COLS - 1), g = tid / RM_COLS;
  double *Ki = sm, *Km = Ki + (size_t)QP * LDK, *Ys = Km + (size_t)QP * LDK, *Rs = Ys + (size_t)QP * RM_COLS, *Es = Rs + (size_t)QP * RM_COLS;
  const int j = blockIdx.x * RM_COLS + c;
  {  // (lane = column, the wavefronts take the rows in turn; the loads of a thread in flight together)
    const int l = tid & 63;
    double vi[16], vm[16];
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const int i = (tid >> 6) + 4 * u;
      const bool in = i < q && l < q;
      vi[u] = in ? a.Kinv[(long long)i * a.ldq + l] : 0.0;
      vm[u] = in ? a.Kmat[(long long)i * a.ldq + l] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const int i = (tid >> 6) + 4 * u;
      if (i < QP && l < QP) Ki[i * LDK + l] = vi[u], Km[i * LDK + l] = vm[u];
    }
  }
  RSTAMP(1);
  if (a.wide) {
    const WideArgs &w = a.w;
    const int r = w.dyn[0], nl = w.dyn[1];
    const int *Rl = w.dyn + 2, *Ll = w.dyn + 2 + w.capn;
    double yv[8];  // (all loads before the first store: Y may alias G and N as far as the compiler knows)
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int i = g + 8 * u;
      yv[u] = 0.0;
      if (j < a.n && i < q) {
        if (i < w.m)
          yv[u] = w.G[(long long)(w.n + i) * w.ldg + j];
        else if (i - w.m < r)
          yv[u] = w.N[(long long)Rl[i - w.m] * w.ldn + j];
      }
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int i = g + 8 * u;
      if (j < a.n && i < q) a.Y[(long long)i * a.ldy + j] = yv[u];
      if (i < QP) Ys[i * RM_COLS + c] = yv[u];
    }
    if (j < a.n)
      for (int li = g; li < w.cap; li += 8) {
        double v = 0.0;
        if (li < nl) {
          v = w.N[(long long)Ll[li] * w.ldn + j];
          for (int s = 0; s < r; s++) v -= w.t[(long long)li * w.ldt + s] * w.N[(long long)Rl[s] * w.ldn + j];
        }
        w.BT[(long long)j * w.ldb + li] = v;
      }
  } else
    for (int i = g; i < QP; i += 8) Ys[i * RM_COLS + c] = (j < a.n && i < q) ? a.Y[(long long)i * a.ldy + j] : 0.0;
  __syncthreads();
  RSTAMP(2);
  const int wave = tid >> 6, lane = tid & 63, lr = lane & 15, lk = lane >> 4, m0 = 16 * wave;
  const int kend = (q + 3) & ~3;
  // out = init + sign * Kx' X for this wavefront's 16 rows (operand layout of the instruction: A[m = lr][k = lk],
  // B[k = lk][n = lr], result rows lk + 4 rg)
  auto product = [&](const double *Kx, const double *X, double sign, const double *init, double *out, bool global) {
    if (m0 >= QP) return;  // (wave-uniform)
    d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    if (init) {
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        acc0[rg] = init[(m0 + lk + 4 * rg) * RM_COLS + lr];
        acc1[rg] = init[(m0 + lk + 4 * rg) * RM_COLS + 16 + lr];
      }
    }
    double av[16], b0[16], b1[16];  // all fragments of the k loop first (QP <= 64: 16 steps at most), then the chain
#pragma unroll
    for (int st = 0; st < 16; st++)
      if (4 * st < kend) {  // (wave-uniform)
        av[st] = Kx[(4 * st + lk) * LDK + m0 + lr];
        b0[st] = sign * X[(4 * st + lk) * RM_COLS + lr], b1[st] = sign * X[(4 * st + lk) * RM_COLS + 16 + lr];
      }
#pragma unroll
    for (int st = 0; st < 16; st++)
      if (4 * st < kend) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[st], b0[st], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[st], b1[st], acc1, 0, 0, 0);
      }
#pragma unroll
    for (int rg = 0; rg < 4; rg++) {
      const int m = m0 + lk + 4 * rg;
      if (global) {
        const int jj = blockIdx.x * RM_COLS + lr;
        if (m < q && jj < a.n) out[(long long)m * a.ldy + jj] = acc0[rg];
        if (m < q && jj + 16 < a.n) out[(long long)m * a.ldy + jj + 16] = acc1[rg];
      } else {
        out[m * RM_COLS + lr] = acc0[rg];
        out[m * RM_COLS + 16 + lr] = acc1[rg];
      }
    }
  };
  product(Ki, Ys, 1.0, nullptr, Rs, false);  // Rs = K^-1 Y
  __syncthreads();
  product(Km, Rs, -1.0, Ys, Es, false);  // Es = Y - K Rs
  __syncthreads();
  product(Ki, Es, 1.0, Rs, a.Rm, true);  // Rm = Rs + K^-1 Es
  RSTAMP(3);
}

// last stage K: all its equality rows are carried (no control): BT_K = E_K', count = e
__global__ void k_st_last(int n, int e, int cap, const double *__restrict__ N, long long ldn, double *__restrict__ BT,
                          long long ldb, int *__restrict__ dyn) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j == 0) dyn[0] = 0, dyn[1] = e;
  if (j >= n) return;
  for (int li = 0; li < cap; li++) BT[(long long)j * ldb + li] = li < e ? N[(long long)li * ldn + j] : 0.0;
}
// a fixed initial state must not be left with constraints of its own (the reference cannot
// solve that either, hqp/Hqp_IpLQDOCP.C:2097-2108)
__global__ void k_st_check_fixed(const int *__restrict__ dyn0, int *__restrict__ status) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && dyn0[1] > 0) atomicExch(status, 4);
}

constexpr int X0_BLOCKED = 40, X0_FELL_BACK = 41;  // counters in the status words: blocked inverse of K0 ran / gave up
// free initial state: LU factors (complete pivoting) of the diagonally scaled [V_0 B_0'; B_0 0]; the reference
// factorises the same scaled matrix by Bunch-Kaufman-Parlett (hqp/Hqp_IpLQDOCP.C:1972-1996) and solves by
// substitution.  K0lu: the factors; K0s: 3 q doubles - the scaling, the row and the column exchanges.
template <int NT>
__global__ void __launch_bounds__(NT, NT / 256) k_st_init_factor(int n0, int cap, const double *__restrict__ V, long long ldv,
                                                       const double *__restrict__ BT, long long ldb,
                                                       const int *__restrict__ dyn0, double *__restrict__ K0lu,
                                                       double *__restrict__ K0mat, double *__restrict__ K0s, long long ldq, int qmax,
                                                       int *__restrict__ status, double *scratch, const int *only_if) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  __shared__ ArgMax red[16];
  const int tid = threadIdx.x, nt = blockDim.x;
  // (only_if: the flag of the blocked inverse, k_x0_*; zero = its result stands, nothing to do here)
  if (only_if) {
    if (*only_if == 0) return;  // (uniform)
    if (tid == 0) atomicAdd(status + X0_FELL_BACK, 1);
  }
  const int c = dyn0[1], q = n0 + c, ld = q | 1;
  constexpr bool BIG = NT != 256;  // (the kind of memory fixed by the instantiation: see k_st_small)
  double *Km, *dsc;
  if constexpr (BIG)
    Km = scratch, dsc = sm;
  else
    Km = sm, dsc = sm + (size_t)q * ld;
  double *colv = dsc + q, *rowv = colv + (q > 64 ? q : 64);
  int *pr = (int *)(rowv + (q > 128 ? q : 128)), *pc = pr + q;
  for (int e = tid; e < q * q; e += nt) {
    const int i = e / q, j = e - i * q;
    double v;
    if (i < n0 && j < n0)
      v = V[(long long)i * ldv + j];
    else if (i >= n0 && j >= n0)
      v = 0.0;
    else
      v = BT[(long long)(i < n0 ? i : j) * ldb + (i < n0 ? j : i) - n0];
    Km[i * ld + j] = v;
  }
  __syncthreads();
  for (int e = tid; e < qmax * qmax; e += nt) {
    const int i = e / qmax, j = e - i * qmax;
    K0mat[(long long)i * ldq + j] = (i < q && j < q) ? Km[i * ld + j] : 0.0;
  }
  for (int i = tid; i < q; i += nt) {
    double d = 1.0;
    if (i < n0) {
      const double kii = Km[i * ld + i];
      if (kii > 1.0) d = 1.0 / sqrt(kii);
    }
    dsc[i] = d;
  }
  __syncthreads();
  for (int e = tid; e < q * q; e += nt) Km[(e / q) * ld + e % q] *= dsc[e / q] * dsc[e % q];
  __syncthreads();
  const int bad = lu_complete<BIG ? HQPKKT_GJ_RB : 1>(Km, q, ld, pr, pc, colv, rowv, red);
  if (bad && tid == 0) atomicExch(status, 4);
  __syncthreads();
  for (int e = tid; e < qmax * qmax; e += nt) {
    const int i = e / qmax, j = e - i * qmax;
    K0lu[(long long)i * ldq + j] = (i < q && j < q) ? Km[i * ld + j] : 0.0;
  }
  for (int i = tid; i < q; i += nt) K0s[i] = dsc[i], K0s[qmax + i] = (double)pr[i], K0s[2 * qmax + i] = (double)pc[i];
  if (tid == 0) K0s[3 * qmax] = 0.0;  // the area K0lu holds LU factors
}

// The same matrix of a free initial state with hundreds of components, inverted by the blocked sweep on the whole chip
// (k_blk_pivot / two products / k_blk_fixup per block of 64, as for a stage's K): k_x0_prepare writes the scaled matrix
// (identity beyond the live order n0 + carried rows), k_x0_final the inverse, unscaled and symmetrised, into the area
// of the factors, k_x0_check compares K0 K0^-1 with the identity and leaves the verdict in K0s[3 qmax] (1: the area
// holds the inverse; 0: LU factors - written by k_st_init_factor behind it, which runs only where the sweep gave up or
// failed the check).  1000 components: 99 ms -> a few ms.  hqp/Hqp_IpLQDOCP.C:1972-1996 factorises the same matrix.
struct X0Args {
  int n0, qmax;
  const double *V;
  long long ldv;
  const double *BT;
  long long ldb;
  const int *dyn0;
  double *K0inv, *K0mat, *K0s;
  long long ldq;
  double *scratch;
  int *status;
};
__global__ void k_x0_prepare(X0Args a) {
  const BigScratch bs = big_scratch(a.scratch, a.qmax);
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long long)a.qmax * a.qmax) return;
  const int i = (int)(e / a.qmax), j = (int)(e - (long long)i * a.qmax), q = a.n0 + a.dyn0[1];
  double v = 0.0, sc = 1.0;
  if (i < q && j < q) {
    if (i < a.n0 && j < a.n0)
      v = a.V[(long long)i * a.ldv + j];
    else if (i < a.n0 || j < a.n0)
      v = a.BT[(long long)(i < a.n0 ? i : j) * a.ldb + (i < a.n0 ? j : i) - a.n0];
    if (i < a.n0) {
      const double kii = a.V[(long long)i * a.ldv + i];
      if (kii > 1.0) sc = 1.0 / sqrt(kii);
    }
    if (j < a.n0) {
      const double kjj = a.V[(long long)j * a.ldv + j];
      if (kjj > 1.0) sc *= 1.0 / sqrt(kjj);
    }
  }
  a.K0mat[(long long)i * a.ldq + j] = v;
  bs.Ks[(long long)i * bs.ldk + j] = (i < q && j < q) ? v * sc : (i == j ? 1.0 : 0.0);
  if (j == 0) {
    double d = 0.0;
    if (i < q) {
      d = 1.0;
      if (i < a.n0) {
        const double kii = a.V[(long long)i * a.ldv + i];
        if (kii > 1.0) d = 1.0 / sqrt(kii);
      }
    }
    bs.dsc[i] = d;
  }
  if (e == 0) bs.flags[0] = bs.flags[1] = bs.flags[2] = bs.flags[3] = bs.flags[4] = 0, atomicAdd(a.status + X0_BLOCKED, 1);
}
__global__ void k_x0_final(X0Args a) {
  const BigScratch bs = big_scratch(a.scratch, a.qmax);
  if (bs.flags[0]) return;
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long long)a.qmax * a.qmax) return;
  const int i = (int)(e / a.qmax), l = (int)(e - (long long)i * a.qmax), q = a.n0 + a.dyn0[1];
  double v = 0.0;
  if (i < q && l < q) v = -0.5 * (bs.Ks[(long long)i * bs.ldk + l] + bs.Ks[(long long)l * bs.ldk + i]) * bs.dsc[i] * bs.dsc[l];
  a.K0inv[(long long)i * a.ldq + l] = v;
}
// E = K0 K0inv (in the area of the swept matrix) against the identity of the live order; the verdict
__global__ void __launch_bounds__(1024) k_x0_check(X0Args a, double tol) {
  __shared__ ArgMax red[16];
  const BigScratch bs = big_scratch(a.scratch, a.qmax);
  if (bs.flags[0] == 0) {
    const int q = a.n0 + a.dyn0[1];
    ArgMax am{0.0, 0};
    for (long long e = threadIdx.x; e < (long long)a.qmax * a.qmax; e += blockDim.x) {
      const int i = (int)(e / a.qmax), l = (int)(e - (long long)i * a.qmax);
      const double d = fabs(bs.Ks[(long long)i * bs.ldk + l] - ((i == l && i < q) ? 1.0 : 0.0));
      if (!(d <= am.v)) am.v = d == d ? d : __longlong_as_double(0x7ff0000000000000LL);
    }
    am = block_argmax(am, red);
    if (threadIdx.x == 0) {
      bs.flags[4] = __float_as_int((float)am.v);  // (introspection: max |K0 K0^-1 - I|)
      if (!(am.v <= tol)) bs.flags[0] = 1;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) a.K0s[3 * a.qmax] = bs.flags[0] ? 0.0 : 1.0;
}
// the solve with the inverse: rhs (nb = -b, pb = b with b = [v_0 ; beta_0], zero beyond the live order), three products
// by k_st_gemv_rows (y = K0inv nb; r = -(pb + K0 y); z = y + K0inv r: one round of refinement against K0), the result
struct X0Vec {
  int n0, cap0, qmax;
  const int *dyn0;
  const double *K0s;
  const double *v0, *beta0;
  double *nb, *pb;
  const double *z;
  double *x0, *eta0;
};
__global__ void k_x0_rhs(X0Vec a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.qmax) return;
  const int q = a.n0 + a.dyn0[1];
  const double b = i < q ? (i < a.n0 ? a.v0[i] : a.beta0[i - a.n0]) : 0.0;
  a.nb[i] = -b, a.pb[i] = b;
}
__global__ void k_x0_out(X0Vec a) {
  if (a.K0s[3 * a.qmax] != 1.0) return;  // (the factors are in use: k_st_x0_free writes the result)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n0 + a.cap0) return;
  const int q = a.n0 + a.dyn0[1];
  const double s = i < q ? a.z[i] : 0.0;
  if (i < a.n0)
    a.x0[i] = s;
  else
    a.eta0[i - a.n0] = s;
}

// ---------------------------------------------------------------------------------------
// dense matrix-vector products of the two sweeps (HBM-bound: a stage block is read once)
// rows form: y[i] = add[i] + sum_j A[i][j] x[j] (+ sum_l A2[i][l] x2[l]), one wavefront per row
struct GemvRows {
  const double *A;
  long long lda;
  int M, N;
  const double *x;
  const double *add;  // may be null
  const double *A2;   // optional second block (the carried rows: BT eta)
  long long lda2;
  const int *n2;      // live columns of A2 (device)
  const double *x2;
  double *y;
  double scale;       // y = scale * (...)
};
__global__ void __launch_bounds__(256) k_st_gemv_rows(GemvRows g) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= g.M) return;
  const double *ar = g.A + (long long)row * g.lda;
  double s = 0.0;
  if ((((size_t)ar) & 15) == 0) {
    // 16-byte loads of the matrix row (it is what streams from HBM), four in flight per lane
    const double2_t *a2 = (const double2_t *)ar;
    const int n2 = g.N >> 1;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int j = lane;
    for (; j + 192 < n2; j += 256) {
      const double2_t v0 = a2[j], v1 = a2[j + 64], v2 = a2[j + 128], v3 = a2[j + 192];
      s0 += v0.x * g.x[2 * j] + v0.y * g.x[2 * j + 1];
      s1 += v1.x * g.x[2 * (j + 64)] + v1.y * g.x[2 * (j + 64) + 1];
      s2 += v2.x * g.x[2 * (j + 128)] + v2.y * g.x[2 * (j + 128) + 1];
      s3 += v3.x * g.x[2 * (j + 192)] + v3.y * g.x[2 * (j + 192) + 1];
    }
    for (; j < n2; j += 64) {
      const double2_t v0 = a2[j];
      s0 += v0.x * g.x[2 * j] + v0.y * g.x[2 * j + 1];
    }
    if ((g.N & 1) && lane == 0) s0 += ar[g.N - 1] * g.x[g.N - 1];
    s = (s0 + s1) + (s2 + s3);
  } else
    for (int j = lane; j < g.N; j += 64) s += ar[j] * g.x[j];
  if (g.A2) {
    const int k2 = *g.n2;
    const double *br = g.A2 + (long long)row * g.lda2;
    for (int j = lane; j < k2; j += 64) s += br[j] * g.x2[j];
  }
  s = kktdev::wave_sum(s);
  if (lane == 0) g.y[row] = g.scale * ((g.add ? g.add[row] : 0.0) + s);
}
// The same product with a SYMMETRIC matrix stored in full (V of a stage: the lower tiles and their mirror images, exactly
// symmetric), reading only the tiles on and below the diagonal: a tile of SV_R rows x SV_C columns gives the partial dot
// products of its rows (rowpart[column tile][row]) and, from the same registers, the partial products of the mirrored
// elements (colpart[row tile][column]: sum over the tile's rows i of V[i][j] x[i], elements strictly below the
// diagonal).  k_st_symv_finish adds the partials of a row in a fixed order (reproducible from run to run), the carried
// rows' term and `add`.  HBM traffic: half the matrix plus (N / SV_R + N / SV_C) N / 2 doubles of partials, twice.
constexpr int SV_R = 64, SV_P = 4, SV_C = 128 * SV_P, SV_U = 4;  // rows, 16-byte loads per lane and row, columns of a tile; rows in flight
struct SymvArgs {
  const double *V;
  long long ldv;
  int N;
  const double *x;
  double *rowpart;  // [ceil(N / SV_C)][N]
  double *colpart;  // [ceil(N / SV_R)][N]
};
// (measured on the 5000 x 5000 V of the headline, tools/symv_probe.hip: 22.8 us for the 113 MB of the triangle's tiles
// against 37.1 us for the rows form over the whole matrix; tiles of 64 x 256, 32 x 512, 96 x 512, 128 x 512 and 2 / 8
// rows in flight: 22.2 - 32 us; the row sums by DPP, 1 us less than by ds_bpermute)
__device__ __forceinline__ void symv_tile(const SymvArgs &g, const int t, double (*red)[SV_C]) {
  // tile number -> (row tile bi, column tile bj) over the tiles on and below the diagonal: the RT = SV_C / SV_R row tiles
  // RT g .. RT g + RT - 1 have g + 1 column tiles each, RT g (g + 1) / 2 tiles stand before them
  constexpr int RT = SV_C / SV_R;
  int gq = (int)((sqrt(1.0 + 8.0 * t / RT) - 1.0) * 0.5);
  while (RT * gq * (gq + 1) / 2 > t) gq--;
  while (RT * (gq + 1) * (gq + 2) / 2 <= t) gq++;
  const int rem = t - RT * gq * (gq + 1) / 2, bi = RT * gq + rem / (gq + 1), bj = rem % (gq + 1), r0 = bi * SV_R, c0 = bj * SV_C;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool diag = c0 + SV_C - 1 >= r0;  // the tile meets the diagonal: element masks
  int jj[SV_P];                           // this lane's column pairs (jj, jj + 1)
  double xa[SV_P], xb[SV_P], ca[SV_P], cb[SV_P];
#pragma unroll
  for (int p = 0; p < SV_P; p++) {
    jj[p] = c0 + 2 * lane + 128 * p;
    xa[p] = jj[p] < g.N ? g.x[jj[p]] : 0.0, xb[p] = jj[p] + 1 < g.N ? g.x[jj[p] + 1] : 0.0;
    ca[p] = cb[p] = 0.0;
  }
  const int ra = r0 + wave * (SV_R / 4), rb = min(g.N, ra + SV_R / 4);
  const double *row = g.V + (long long)ra * g.ldv;
  double *rp = g.rowpart + (long long)bj * g.N;
  for (int i = ra; i < rb; i += SV_U) {
    double2_t v[SV_U][SV_P];
    double xi[SV_U];
#pragma unroll
    for (int u = 0; u < SV_U; u++) {
      const bool live = i + u < rb;
#pragma unroll
      for (int p = 0; p < SV_P; p++)
        v[u][p] = (live && jj[p] < g.N) ? *(const double2_t *)(row + (long long)u * g.ldv + jj[p]) : double2_t{0.0, 0.0};
      xi[u] = live ? g.x[i + u] : 0.0;
    }
    row += SV_U * g.ldv;
#pragma unroll
    for (int u = 0; u < SV_U; u++) {
      const int ii = i + u;
      double s = 0.0;
#pragma unroll
      for (int p = 0; p < SV_P; p++) {
        double a = v[u][p].x, b = jj[p] + 1 < g.N ? v[u][p].y : 0.0;
        if (diag) {  // row part: columns <= ii
          if (jj[p] > ii) a = 0.0;
          if (jj[p] + 1 > ii) b = 0.0;
        }
        s += a * xa[p] + b * xb[p];
        if (diag) {  // mirrored part: columns < ii
          if (jj[p] == ii) a = 0.0;
          if (jj[p] + 1 == ii) b = 0.0;
        }
        ca[p] += a * xi[u], cb[p] += b * xi[u];
      }
      s = kktdev::wave_sum_dpp(s);
      if (lane == 0 && ii < rb) rp[ii] = s;
    }
  }
#pragma unroll
  for (int p = 0; p < SV_P; p++) red[wave][128 * p + 2 * lane] = ca[p], red[wave][128 * p + 2 * lane + 1] = cb[p];
  __syncthreads();
  for (int c = threadIdx.x; c < SV_C; c += 256)
    if (c0 + c < g.N) g.colpart[(long long)bi * g.N + c0 + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}
__global__ void __launch_bounds__(256) k_st_symv_tiles(SymvArgs g) {
  __shared__ double red[4][SV_C];
  symv_tile(g, blockIdx.x, red);
}
struct SymvFinish {
  int N;
  const double *rowpart, *colpart;
  const double *add;  // may be null
  const double *A2;   // optional block of carried rows (as GemvRows)
  long long lda2;
  const int *n2;
  const double *x2;
  double *y;
  double scale;
};
__device__ __forceinline__ void symv_finish_blk(const SymvFinish &g, const int blk, double (*red)[64]) {
  // 64 columns per workgroup, four threads per column: thread group q adds every fourth partial of the mirrored part
  // (eight loads in flight), group 0 the row parts and the carried rows' term; fixed order throughout
  const int cl = threadIdx.x & 63, q = threadIdx.x >> 6, j = blk * 64 + cl;
  double s = 0.0;
  if (j < g.N) {
    const int bi = j / SV_R, nrt = (g.N + SV_R - 1) / SV_R, ct = (bi * SV_R + SV_R - 1) / SV_C;
    int b = bi + q;
    for (; b + 28 < nrt; b += 32) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = g.colpart[(long long)(b + 4 * u) * g.N + j];
#pragma unroll
      for (int u = 0; u < 8; u++) s += v[u];
    }
    for (; b < nrt; b += 4) s += g.colpart[(long long)b * g.N + j];
    {  // the row parts: every fourth one per group, up to eight loads in flight
      double v[8];
      for (int t0 = q; t0 <= ct; t0 += 32) {
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = t0 + 4 * u <= ct ? g.rowpart[(long long)(t0 + 4 * u) * g.N + j] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; u++) s += v[u];
      }
    }
    if (q == 0) {
      if (g.A2) {
        const int k2 = *g.n2;
        const double *br = g.A2 + (long long)j * g.lda2;
        for (int l = 0; l < k2; l++) s += br[l] * g.x2[l];
      }
    }
  }
  red[q][cl] = s;
  __syncthreads();
  if (q == 0 && j < g.N) g.y[j] = g.scale * ((g.add ? g.add[j] : 0.0) + ((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl])));
}
__global__ void __launch_bounds__(256) k_st_symv_finish(SymvFinish g) {
  __shared__ double red[4][64];
  symv_finish_blk(g, blockIdx.x, red);
}
// The products with V that stand outside the chains of the two sweeps - g_k = V+ f_k with the dynamics' right-hand side
// (known before the backward sweep starts), the dynamics rows' multipliers V+ x+ + v+ + B+' eta+ (wanted by nobody before
// the sweep is over) - for MANY stages per launch: item i holds the tiles tile0 .. (next item's tile0) of the launch and
// the finishing blocks fin0 ..; partial sums of its own.  The vectors of the caller (right-hand side, multipliers) are
// addressed relative to a base, so that one table serves every set of vectors the solve runs on.
struct SymvItem {
  SymvArgs a;
  SymvFinish f;
  int tile0, fin0;      // first tile / finishing block inside the launch
  int xrel, yrel;       // a.x = xbase + xoff / f.y = ybase + yoff
  long long xoff, yoff;
};
__device__ __forceinline__ int symv_item_of(const SymvItem *items, int cnt, int blk, bool fin) {
  int lo = 0, hi = cnt - 1;  // the last item whose first block is <= blk
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((fin ? items[mid].fin0 : items[mid].tile0) <= blk) lo = mid; else hi = mid - 1;
  }
  return lo;
}
// (a grid smaller than the number of tiles / blocks: every workgroup takes several, in a stride)
__global__ void __launch_bounds__(256) k_st_symv_tiles_batch(const SymvItem *__restrict__ items, int cnt, int tiles, const double *xbase) {
  __shared__ double red[4][SV_C];
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    const SymvItem &it = items[symv_item_of(items, cnt, t, false)];
    SymvArgs a = it.a;
    if (it.xrel) a.x = xbase + it.xoff;
    symv_tile(a, t - it.tile0, red);
    __syncthreads();
  }
}
__global__ void __launch_bounds__(256) k_st_symv_finish_batch(const SymvItem *__restrict__ items, int cnt, int fins, double *ybase) {
  __shared__ double red[4][64];
  for (int t = blockIdx.x; t < fins; t += gridDim.x) {
    const SymvItem &it = items[symv_item_of(items, cnt, t, true)];
    SymvFinish f = it.f;
    if (it.yrel) f.y = ybase + it.yoff;
    symv_finish_blk(f, t - it.fin0, red);
    __syncthreads();
  }
}
// the same for few, long rows (Rm x with a handful of controls and thousands of states): one
// workgroup per row, so that a row's bytes are in flight from 256 threads
__global__ void __launch_bounds__(256) k_st_gemv_wide(GemvRows g) {
  __shared__ double red[4];
  const int row = blockIdx.x;
  const double *ar = g.A + (long long)row * g.lda;
  double s = 0.0;
  if ((((size_t)ar) & 15) == 0) {
    // 16-byte loads, four in flight per thread (a row of 5000 states: 10 loads per thread, three batches instead of
    // twenty dependent round trips: 10.7 -> ~4 us)
    const double2_t *a2 = (const double2_t *)ar;
    const int n2 = g.N >> 1;
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int j = threadIdx.x;
    for (; j + 768 < n2; j += 1024) {
      const double2_t v0 = a2[j], v1 = a2[j + 256], v2 = a2[j + 512], v3 = a2[j + 768];
      s += v0.x * g.x[2 * j] + v0.y * g.x[2 * j + 1];
      s1 += v1.x * g.x[2 * (j + 256)] + v1.y * g.x[2 * (j + 256) + 1];
      s2 += v2.x * g.x[2 * (j + 512)] + v2.y * g.x[2 * (j + 512) + 1];
      s3 += v3.x * g.x[2 * (j + 768)] + v3.y * g.x[2 * (j + 768) + 1];
    }
    for (; j < n2; j += 256) {
      const double2_t v0 = a2[j];
      s += v0.x * g.x[2 * j] + v0.y * g.x[2 * j + 1];
    }
    if ((g.N & 1) && threadIdx.x == 0) s += ar[g.N - 1] * g.x[g.N - 1];
    s = (s + s1) + (s2 + s3);
  } else
    for (int j = threadIdx.x; j < g.N; j += 256) s += ar[j] * g.x[j];
  s = kktdev::wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) g.y[row] = g.scale * ((g.add ? g.add[row] : 0.0) + ((red[0] + red[1]) + (red[2] + red[3])));
}
// columns form: part[s][j] = sum over the rows k of chunk s of A[k][j] x[k]; with one chunk the
// result y[j] = add[j] + alpha * sum is written directly
struct GemvCols {
  const double *A;
  long long lda;
  int K, N;
  const double *x;
  const double *add;
  double alpha;
  double *y;
  double *part;  // nchunk x N
  int rows_per_chunk;
  const double *add2 = nullptr;  // optional second result y2 = y + add2 (the backward sweep: v_k and, with the product
  double *y2 = nullptr;          // V_k f_{k-1} computed ahead of the sweep, v_k + V_k f_{k-1})
};
__global__ void __launch_bounds__(256) k_st_gemv_cols(GemvCols g) {
  // two neighbouring columns per thread (one 16-byte load per row; rows 16-byte aligned when lda is even and the
  // block starts aligned), four rows in flight
  const int j = 2 * (blockIdx.x * blockDim.x + threadIdx.x);
  if (j >= g.N) return;
  const int k0 = blockIdx.y * g.rows_per_chunk, k1 = min(g.K, k0 + g.rows_per_chunk);
  double s0 = 0.0, s1 = 0.0;
  const double *a = g.A + (long long)k0 * g.lda + j;
  if (j + 1 < g.N && (((size_t)a) & 15) == 0 && (g.lda & 1) == 0) {
    double t0 = 0.0, t1 = 0.0, u0 = 0.0, u1 = 0.0, w0 = 0.0, w1 = 0.0;
    int k = k0;
    // (eight rows in flight first: with few rows and one chunk - Y' rho, a few dozen rows - the loop is a chain of
    // memory round trips; the sums are added in the order of the four-row form)
    for (; k + 7 < k1; k += 8, a += 8 * g.lda) {
      const double2_t v0 = *(const double2_t *)a, v1 = *(const double2_t *)(a + g.lda), v2 = *(const double2_t *)(a + 2 * g.lda),
                      v3 = *(const double2_t *)(a + 3 * g.lda), v4 = *(const double2_t *)(a + 4 * g.lda), v5 = *(const double2_t *)(a + 5 * g.lda),
                      v6 = *(const double2_t *)(a + 6 * g.lda), v7 = *(const double2_t *)(a + 7 * g.lda);
      const double x0 = g.x[k], x1 = g.x[k + 1], x2 = g.x[k + 2], x3 = g.x[k + 3], x4 = g.x[k + 4], x5 = g.x[k + 5], x6 = g.x[k + 6],
                   x7 = g.x[k + 7];
      s0 += v0.x * x0, s1 += v0.y * x0, t0 += v1.x * x1, t1 += v1.y * x1;
      u0 += v2.x * x2, u1 += v2.y * x2, w0 += v3.x * x3, w1 += v3.y * x3;
      s0 += v4.x * x4, s1 += v4.y * x4, t0 += v5.x * x5, t1 += v5.y * x5;
      u0 += v6.x * x6, u1 += v6.y * x6, w0 += v7.x * x7, w1 += v7.y * x7;
    }
    for (; k + 3 < k1; k += 4, a += 4 * g.lda) {
      const double2_t v0 = *(const double2_t *)a, v1 = *(const double2_t *)(a + g.lda), v2 = *(const double2_t *)(a + 2 * g.lda),
                      v3 = *(const double2_t *)(a + 3 * g.lda);
      const double x0 = g.x[k], x1 = g.x[k + 1], x2 = g.x[k + 2], x3 = g.x[k + 3];
      s0 += v0.x * x0, s1 += v0.y * x0, t0 += v1.x * x1, t1 += v1.y * x1;
      u0 += v2.x * x2, u1 += v2.y * x2, w0 += v3.x * x3, w1 += v3.y * x3;
    }
    for (; k < k1; k++, a += g.lda) {
      const double2_t v0 = *(const double2_t *)a;
      s0 += v0.x * g.x[k], s1 += v0.y * g.x[k];
    }
    s0 = (s0 + t0) + (u0 + w0), s1 = (s1 + t1) + (u1 + w1);
  } else {
    for (int k = k0; k < k1; k++, a += g.lda) {
      s0 += a[0] * g.x[k];
      if (j + 1 < g.N) s1 += a[1] * g.x[k];
    }
  }
  if (gridDim.y == 1) {
    const double r0 = (g.add ? g.add[j] : 0.0) + g.alpha * s0;
    g.y[j] = r0;
    if (g.y2) g.y2[j] = r0 + g.add2[j];
    if (j + 1 < g.N) {
      const double r1 = (g.add ? g.add[j + 1] : 0.0) + g.alpha * s1;
      g.y[j + 1] = r1;
      if (g.y2) g.y2[j + 1] = r1 + g.add2[j + 1];
    }
  } else {
    g.part[(long long)blockIdx.y * g.N + j] = s0;
    if (j + 1 < g.N) g.part[(long long)blockIdx.y * g.N + j + 1] = s1;
  }
}
__global__ void k_st_cols_finish(int N, int nchunk, const double *__restrict__ part, const double *__restrict__ add,
                                 double alpha, double *__restrict__ y, const double *__restrict__ add2 = nullptr,
                                 double *__restrict__ y2 = nullptr) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  double s = 0.0;
  int c = 0;
  for (; c + 8 <= nchunk; c += 8) {  // (eight loads in flight; the order of the sum as before)
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = part[(long long)(c + u) * N + j];
#pragma unroll
    for (int u = 0; u < 8; u++) s += v[u];
  }
  for (; c < nchunk; c++) s += part[(long long)c * N + j];
  const double r = (add ? add[j] : 0.0) + alpha * s;
  y[j] = r;
  if (y2) y2[j] = r + add2[j];
}

// q = -(r1 - C' tz)   (the reference's gx, gu: hqp/Hqp_IpLQDOCP.C:884-918)
__global__ void k_st_q(int n, const int *__restrict__ CTp, const int *__restrict__ CTc, const int *__restrict__ CTs,
                       const double *__restrict__ vals, const double *__restrict__ tz, const double *__restrict__ r1,
                       double *__restrict__ q) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
  for (int k = CTp[i]; k < CTp[i + 1]; k++) s += vals[CTs[k]] * tz[CTc[k]];
  q[i] = s - r1[i];
}

// backward sweep, the small part of a stage (one workgroup):
//   nu = [a_k ; beta+ + B+ f],  y0 = [gam_u ; nu_R],  rho = K^-1 y0 (+ one round of refinement against K:
//   the product with an explicit inverse alone has a residual of cond(K) eps, a solve by factors has not),
//   beta = nu_L - t nu_R
struct BwdSmall {
  int n, m, np, e, capn, cap, qmax;
  const int *eq_rows;        // QP row index of the own equality rows
  const double *r2;
  const int *cnt_next;       // live carried rows coming in (nullptr: none)
  const double *beta_next;
  const double *BT_next;     // np x ldbn
  long long ldbn;
  const double *f;           // r2 + first dynamics row of the stage
  const double *gam;         // n + m
  const double *Kinv;
  const double *Kmat;
  long long ldq;
  const double *t;
  long long ldt;
  const int *dyn;
  double *rho;               // qmax
  double *beta;              // cap
};
__global__ void __launch_bounds__(256) k_st_bwd_small(BwdSmall a) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int tid = threadIdx.x, nt = blockDim.x;
  const int cn = a.cnt_next ? *a.cnt_next : 0;
  const int r = a.dyn[0], nl = a.dyn[1];
  const int *Rl = a.dyn + 2, *Ll = a.dyn + 2 + a.capn;
  double *nu = sm, *y0 = sm + a.capn + 1;
  // K of order <= 64: this thread's 16 entries of column (tid & 63) of K^-1 and of K are requested before anything else
  // (they come from HBM: the factorisation wrote them long ago) and the three products below run out of registers,
  // four partial sums per entry meeting in LDS.  (Both are stored zero-padded to qmax and symmetric.)
  const bool fastq = a.qmax <= 64 && nt == 256;
  const int ci = tid & 63, part = tid >> 6;
  double ki[16], km[16];
  if (fastq) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const int j = part * 16 + u;
      const bool ok = j < a.qmax && ci < a.qmax;
      ki[u] = ok ? a.Kinv[(long long)j * a.ldq + ci] : 0.0;
      km[u] = ok ? a.Kmat[(long long)j * a.ldq + ci] : 0.0;
    }
  }
  for (int i = tid; i < a.e; i += nt) nu[i] = a.r2[a.eq_rows[i]];
  // carried rows: beta+ + B+ f, one wavefront per row
  for (int li = tid >> 6; li < cn; li += nt >> 6) {
    double s = 0.0;
    for (int k = tid & 63; k < a.np; k += 64) s += a.BT_next[(long long)k * a.ldbn + li] * a.f[k];
    s = kktdev::wave_sum(s);
    if ((tid & 63) == 0) nu[a.e + li] = a.beta_next[li] + s;
  }
  __syncthreads();
  const int q = a.m + r;
  for (int i = tid; i < q; i += nt) y0[i] = i < a.m ? a.gam[a.n + i] : nu[Rl[i - a.m]];
  __syncthreads();
  double *rh = y0 + a.qmax, *rs = rh + a.qmax;
  if (fastq) {
    double *P = rs + a.qmax;  // 4 x 64 partial sums
    auto col_times = [&](const double(&kk)[16], const double *x) {
      double s = 0.0;
#pragma unroll
      for (int u = 0; u < 16; u++) {
        const int j = part * 16 + u;
        s = fma(kk[u], j < q ? x[j] : 0.0, s);
      }
      P[part * 64 + ci] = s;
    };
    col_times(ki, y0);
    __syncthreads();
    if (tid < q) rh[tid] = (P[tid] + P[64 + tid]) + (P[128 + tid] + P[192 + tid]);
    __syncthreads();
    col_times(km, rh);
    __syncthreads();
    if (tid < q) rs[tid] = y0[tid] - ((P[tid] + P[64 + tid]) + (P[128 + tid] + P[192 + tid]));
    __syncthreads();
    col_times(ki, rs);
    __syncthreads();
    if (tid < a.qmax) a.rho[tid] = tid < q ? rh[tid] + ((P[tid] + P[64 + tid]) + (P[128 + tid] + P[192 + tid])) : 0.0;
  } else {
  for (int i = tid; i < q; i += nt) {
    double s = 0.0;  // K^-1 and K are stored symmetric: column i, so that neighbouring threads read neighbouring words
    for (int j = 0; j < q; j++) s += a.Kinv[(long long)j * a.ldq + i] * y0[j];
    rh[i] = s;
  }
  __syncthreads();
  for (int i = tid; i < q; i += nt) {
    double s = y0[i];
    for (int j = 0; j < q; j++) s -= a.Kmat[(long long)j * a.ldq + i] * rh[j];
    rs[i] = s;
  }
  __syncthreads();
  for (int i = tid; i < a.qmax; i += nt) {
    double s = 0.0;
    if (i < q) {
      for (int j = 0; j < q; j++) s += a.Kinv[(long long)j * a.ldq + i] * rs[j];
      s += rh[i];
    }
    a.rho[i] = s;
  }
  }
  for (int li = tid; li < a.cap; li += nt) {
    double s = 0.0;
    if (li < nl) {
      s = nu[Ll[li]];
      for (int k = 0; k < r; k++) s -= a.t[(long long)li * a.ldt + k] * nu[Rl[k]];
    }
    a.beta[li] = s;
  }
}

// forward sweep, the small part of a stage (one workgroup):
//   [u ; yhat] = -(Rm x + rho);  multipliers of the rows of N: consumed yhat - t' eta, leftover eta;
//   own rows -> dy, carried rows -> eta of stage k+1
struct FwdSmall {
  int n, m, e, capn, cap, qmax;
  const double *uy;    // -(Rm x_k + rho), qmax entries (k_st_gemv_wide)
  const double *t;
  long long ldt;
  const int *dyn;
  const double *eta;   // multipliers of this stage's leftover rows (cap)
  const int *eq_rows;
  double *u;           // m (inside the s vector)
  double *dy;
  double *eta_next;    // capn - e entries at most
  int cap_next;
};
__global__ void __launch_bounds__(256) k_st_fwd_small(FwdSmall a) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int tid = threadIdx.x, nt = blockDim.x;
  const int r = a.dyn[0], nl = a.dyn[1];
  const int *Rl = a.dyn + 2, *Ll = a.dyn + 2 + a.capn;
  const double *uy = a.uy;
  double *yN = sm;
  for (int i = tid; i < a.m; i += nt) a.u[i] = uy[i];
  for (int s = tid; s < r; s += nt) {
    double v = uy[a.m + s];
    for (int li = 0; li < nl; li++) v -= a.t[(long long)li * a.ldt + s] * a.eta[li];
    yN[Rl[s]] = v;
  }
  for (int li = tid; li < nl; li += nt) yN[Ll[li]] = a.eta[li];
  __syncthreads();
  const int c = r + nl;
  for (int i = tid; i < a.e; i += nt) a.dy[a.eq_rows[i]] = yN[i];
  for (int i = tid; i < a.cap_next; i += nt) a.eta_next[i] = (a.e + i < c) ? yN[a.e + i] : 0.0;
}

// initial state.  Fixed: x_0 = -r2[fix] / val, eta_0 = 0.  Free: [x_0 ; eta_0] = -K0^-1 [v_0 ; beta_0]
__global__ void k_st_x0_fixed(int n0, const int *__restrict__ fix_rows, const int *__restrict__ fix_src,
                              const double *__restrict__ vals, const double *__restrict__ r2, double *__restrict__ x0,
                              double *__restrict__ eta0, int cap0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n0) x0[i] = -r2[fix_rows[i]] / vals[fix_src[i]];
  if (i < cap0) eta0[i] = 0.0;
}
// multipliers of the rows that fix x_0: -(V_0 x_0 + v_0) / val   (tmp = v_0 + V_0 x_0)
__global__ void k_st_y_fixed(int n0, const int *__restrict__ fix_rows, const int *__restrict__ fix_src,
                             const double *__restrict__ vals, const double *__restrict__ tmp, double *__restrict__ dy) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n0) dy[fix_rows[i]] = -tmp[i] / vals[fix_src[i]];
}
// t <- (D K D)^-1 t by the factors of k_st_init_factor (t in LDS, already scaled by D): row exchanges, L, U, column
// exchanges.  Blocks of 64 unknowns: the part of a block's rows outside its diagonal block is one dot product per
// row (a wavefront per row, coalesced), the diagonal block is solved by one wavefront out of LDS.
__device__ void x0_lu_solve(const double *__restrict__ LU, long long ldq, const double *__restrict__ perm, int qmax, int q,
                            double *t, double *T) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  if (tid == 0)
    for (int s = 0; s < q; s++) {
      const int r = (int)perm[qmax + s];
      const double x = t[s];
      t[s] = t[r], t[r] = x;
    }
  __syncthreads();
  for (int b0 = 0; b0 < q; b0 += 64) {  // L y = t
    const int nb = min(64, q - b0);
    for (int i = wave; i < nb; i += nw) {
      double sum = 0.0;
      for (int j = lane; j < b0; j += 64) sum += LU[(long long)(b0 + i) * ldq + j] * t[j];
      sum = kktdev::wave_sum(sum);
      if (lane == 0) t[b0 + i] -= sum;
    }
    for (int e = tid; e < nb * nb; e += blockDim.x) T[(e / nb) * 65 + e % nb] = LU[(long long)(b0 + e / nb) * ldq + b0 + e % nb];
    __syncthreads();
    if (wave == 0) {
      double y = lane < nb ? t[b0 + lane] : 0.0;
      for (int s = 0; s < nb; s++) {
        const double ys = __shfl(y, s);
        if (lane > s && lane < nb) y -= T[lane * 65 + s] * ys;
      }
      if (lane < nb) t[b0 + lane] = y;
    }
    __syncthreads();
  }
  for (int b0 = ((q - 1) / 64) * 64; b0 >= 0; b0 -= 64) {  // U z = y
    const int nb = min(64, q - b0);
    for (int i = wave; i < nb; i += nw) {
      double sum = 0.0;
      for (int j = b0 + nb + lane; j < q; j += 64) sum += LU[(long long)(b0 + i) * ldq + j] * t[j];
      sum = kktdev::wave_sum(sum);
      if (lane == 0) t[b0 + i] -= sum;
    }
    for (int e = tid; e < nb * nb; e += blockDim.x) T[(e / nb) * 65 + e % nb] = LU[(long long)(b0 + e / nb) * ldq + b0 + e % nb];
    __syncthreads();
    if (wave == 0) {
      double y = lane < nb ? t[b0 + lane] : 0.0;
      for (int s = nb - 1; s >= 0; s--) {
        if (lane == s) y /= T[s * 65 + s];
        const double ys = __shfl(y, s);
        if (lane < s) y -= T[lane * 65 + s] * ys;
      }
      if (lane < nb) t[b0 + lane] = y;
    }
    __syncthreads();
  }
  if (tid == 0)
    for (int s = q - 1; s >= 0; s--) {
      const int r = (int)perm[2 * qmax + s];
      const double x = t[s];
      t[s] = t[r], t[r] = x;
    }
  __syncthreads();
}
__global__ void __launch_bounds__(256) k_st_x0_free(int n0, int cap0, int qmax, const double *__restrict__ K0lu,
                                                    const double *__restrict__ K0mat, const double *__restrict__ K0s, long long ldq,
                                                    const int *__restrict__ dyn0, const double *__restrict__ v0,
                                                    const double *__restrict__ beta0, double *__restrict__ x0,
                                                    double *__restrict__ eta0) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  if (K0s[3 * qmax] == 1.0) return;  // (uniform) the area holds the inverse: k_x0_rhs ... k_x0_out have done it
  const int c = dyn0[1], q = n0 + c, tid = threadIdx.x, nt = blockDim.x;
  double *b = sm, *y = sm + qmax, *t = y + qmax, *T = t + qmax;
  for (int i = tid; i < q; i += nt) {
    b[i] = i < n0 ? v0[i] : beta0[i - n0];
    t[i] = K0s[i] * b[i];
  }
  __syncthreads();
  x0_lu_solve(K0lu, ldq, K0s, qmax, q, t, T);
  for (int i = tid; i < q; i += nt) y[i] = K0s[i] * t[i];
  __syncthreads();
  for (int i = tid; i < q; i += nt) {  // one round of refinement against K0 itself (symmetric: column i read as row i)
    double s = b[i];
    for (int j = 0; j < q; j++) s -= K0mat[(long long)j * ldq + i] * y[j];
    t[i] = K0s[i] * s;
  }
  __syncthreads();
  x0_lu_solve(K0lu, ldq, K0s, qmax, q, t, T);
  for (int i = tid; i < n0 + cap0; i += nt) {
    const double s = i < q ? y[i] + K0s[i] * t[i] : 0.0;
    if (i < n0)
      x0[i] = -s;
    else
      eta0[i - n0] = -s;
  }
}
// last stage: multipliers of its equality rows = eta_K
__global__ void k_st_y_last(int e, const int *__restrict__ eq_rows, const double *__restrict__ eta, double *__restrict__ dy) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < e) dy[eq_rows[i]] = eta[i];
}
__global__ void k_st_gather(int e, const int *__restrict__ rows, const double *__restrict__ src, double *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < e) out[i] = src[rows[i]];
}
__global__ void k_st_negate(int n, const double *__restrict__ s, double *__restrict__ dx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dx[i] = -s[i];
}
// d = s, d2 = s + a2
__global__ void k_st_copy_add(int n, const double *__restrict__ s, double *__restrict__ d, const double *__restrict__ a2,
                              double *__restrict__ d2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) d[i] = s[i], d2[i] = s[i] + a2[i];
}
__global__ void k_st_copy(int n, const double *__restrict__ s, double *__restrict__ d) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) d[i] = s[i];
}

// ---------------------------------------------------------------------------------------
// One system over several ranks (staged_plan.hpp).  The blocks of G_xx this rank has computed in its row strip of the
// work block G go into its slot of the second exchange buffer in LOWER orientation: a block of its own rows as it is,
// a block it computed for the partner's rows (rows = its own columns) transposed.  blockIdx.y = block.
struct PackRect {
  int r0, c0, rows, cols;  // where the block lies in G (work orientation) and its size there
  int transpose;           // the slot holds it transposed: cols x rows, leading dimension rows
  int pad;
  long long off;           // into this rank's slot
};
__global__ void __launch_bounds__(256) k_st_pack_rects(const PackRect *__restrict__ rects, const double *__restrict__ G, long long ldg,
                                                       double *__restrict__ slot) {
  __shared__ double t[32][33];
  const PackRect r = rects[blockIdx.y];
  double *dst = slot + r.off;
  if (!r.transpose) {
    for (int i = blockIdx.x; i < r.rows; i += gridDim.x) {
      const double *src = G + (long long)(r.r0 + i) * ldg + r.c0;
      for (int j = threadIdx.x; j < r.cols; j += blockDim.x) dst[(long long)i * r.cols + j] = src[j];
    }
    return;
  }
  // 32 x 32 tiles through LDS: rows of G are read, rows of the transposed block written
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int tr = (r.rows + 31) / 32, tc = (r.cols + 31) / 32;
  for (int tile = blockIdx.x; tile < tr * tc; tile += gridDim.x) {
    const int i0 = (tile / tc) * 32, j0 = (tile % tc) * 32;
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int i = i0 + ty + 8 * u, j = j0 + tx;
      t[ty + 8 * u][tx] = (i < r.rows && j < r.cols) ? G[(long long)(r.r0 + i) * ldg + r.c0 + j] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int j = j0 + ty + 8 * u, i = i0 + tx;
      if (i < r.rows && j < r.cols) dst[(long long)j * r.rows + i] = t[tx][ty + 8 * u];
    }
    __syncthreads();
  }
}
// rows x cols block copy (the own row strip of V_k out of the transient full block)
__global__ void __launch_bounds__(256) k_st_copy2d(const double *__restrict__ src, long long lds_, double *__restrict__ dst, long long ldd, int rows,
                                                   int cols) {
  for (int i = blockIdx.x; i < rows; i += gridDim.x)
    for (int j = threadIdx.x; j < cols; j += blockDim.x) dst[(long long)i * ldd + j] = src[(long long)i * lds_ + j];
}
// y = add + the ranks' partial vectors in their order (x+ = F s + f over the column strips)
__global__ void k_st_sum_slots(int n, int nslots, const double *__restrict__ slots, long long stride, const double *__restrict__ add,
                               double *__restrict__ y) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = add ? add[i] : 0.0;
  for (int p = 0; p < nslots; p++) s += slots[(long long)p * stride + i];
  y[i] = s;
}
__global__ void k_st_zero(long long n, double *__restrict__ y) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = 0.0;
}
// ---------------------------------------------------------------------------------------
// Dense dynamics (hqpkkt_set_values_staged): the products with the dynamics rows of A that
// residuum() needs, all stages in one launch (blockIdx.y = stage)
struct DynDesc {
  long long oF;
  int ldf, np, nz, col0, row0, ncur;  // F block, its sizes, first column / dynamics row, states of the stage
};
// out2[row0 + li] = F_k[li][:] dx_k - dx_{k+1}[li]
__global__ void __launch_bounds__(256) k_st_dyn_ax(const DynDesc *__restrict__ desc, const double *__restrict__ F,
                                                   const double *__restrict__ dx, double *__restrict__ out2) {
  const DynDesc d = desc[blockIdx.y];
  const int lane = threadIdx.x & 63;
  for (int li = blockIdx.x * 4 + (threadIdx.x >> 6); li < d.np; li += gridDim.x * 4) {
    const double *fr = F + d.oF + (long long)li * d.ldf;
    double s = 0.0;
    for (int j = lane; j < d.nz; j += 64) s += fr[j] * dx[d.col0 + j];
    s = kktdev::wave_sum(s);
    if (lane == 0) out2[d.row0 + li] = s - dx[d.col0 + d.nz + li];
  }
}
// out1[c] = sum_li F_k[li][lc] dy[row0 + li] - dy[row of x_k's dynamics equation]   (c in stage k;
// the last stage has no F: np = 0)
// Both products of residuum() with the dense dynamics rows in ONE pass over the F blocks (two passes: 80 GB at the
// headline size, 14.5 ms): workgroup (bx, k) takes 256 columns of stage k's block, thread = column.  Down the rows it
// accumulates its column of A_dyn' dy; the row sums of A_dyn dx over the workgroup's 256 columns go, 64 rows at a time,
// through a wavefront sum and LDS into `part` (one value per row and column block); k_st_dyn_ax_finish adds the
// column blocks in their order and the -x_{k+1} of the dynamics rows.
__global__ void __launch_bounds__(256) k_st_dyn_both(const DynDesc *__restrict__ desc, const double *__restrict__ F,
                                                     const double *__restrict__ dx, const double *__restrict__ dy,
                                                     double *__restrict__ out1, double *__restrict__ part, int nblk_cols) {
  // wavefront w takes the rows w, w + 4, ... of the block; lane l the columns l, l + 64, l + 128, l + 192 of the
  // workgroup's 256: a row's sum over those columns is ONE wavefront sum (eight rows' sums side by side), a column's
  // sum over the rows meets the other three wavefronts' in LDS at the end
  __shared__ double cs[4][256];
  const DynDesc d = desc[blockIdx.y];
  const int ncols = d.np > 0 ? d.nz : d.ncur;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = blockIdx.x * 256;
  if (c0 >= ncols) return;  // (uniform)
  bool in[4];
  double xj[4], s[4];
  const double *fc[4];
#pragma unroll
  for (int g = 0; g < 4; g++) {
    const int lc = c0 + lane + 64 * g;
    in[g] = lc < ncols;
    xj[g] = (in[g] && lc < d.nz) ? dx[d.col0 + lc] : 0.0;
    fc[g] = F + d.oF + (in[g] ? lc : 0);
    s[g] = 0.0;
  }
  for (int li0 = wave; li0 < d.np; li0 += 32) {  // eight rows of this wavefront per trip: li0, li0 + 4, ...
    double v[8][4], p[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int li = li0 + 4 * u;
#pragma unroll
      for (int g = 0; g < 4; g++) v[u][g] = (in[g] && li < d.np) ? fc[g][(long long)li * d.ldf] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int li = li0 + 4 * u;
      const double yv = li < d.np ? dy[d.row0 + li] : 0.0;
      p[u] = 0.0;
#pragma unroll
      for (int g = 0; g < 4; g++) {
        s[g] = fma(v[u][g], yv, s[g]);
        p[u] = fma(v[u][g], xj[g], p[u]);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int u = 0; u < 8; u++) p[u] += __shfl_xor(p[u], o);
    if (lane == 0) {
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (li0 + 4 * u < d.np) part[(long long)(d.row0 + li0 + 4 * u) * nblk_cols + blockIdx.x] = p[u];
    }
  }
#pragma unroll
  for (int g = 0; g < 4; g++) cs[wave][lane + 64 * g] = s[g];
  __syncthreads();
  const int lc = c0 + tid;
  if (lc < ncols) {
    double t = (cs[0][tid] + cs[1][tid]) + (cs[2][tid] + cs[3][tid]);
    // x_k is the state the previous stage's dynamics produce: -1.0 in that row
    if (blockIdx.y > 0 && lc < d.ncur) t -= dy[d.row0 - d.ncur + lc];
    out1[d.col0 + lc] = t;
  }
}
__global__ void k_st_dyn_ax_finish(const DynDesc *__restrict__ desc, const double *__restrict__ part, int nblk_cols,
                                   const double *__restrict__ dx, double *__restrict__ out2) {
  const DynDesc d = desc[blockIdx.y];
  const int li = blockIdx.x * blockDim.x + threadIdx.x;
  if (li >= d.np) return;
  const int nb = (d.nz + 255) / 256;
  double s = 0.0;
  for (int b = 0; b < nb; b++) s += part[(long long)(d.row0 + li) * nblk_cols + b];
  out2[d.row0 + li] = s - dx[d.col0 + d.nz + li];
}
__global__ void __launch_bounds__(256) k_st_dyn_aty(const DynDesc *__restrict__ desc, const double *__restrict__ F,
                                                    const double *__restrict__ dy, double *__restrict__ out1) {
  const DynDesc d = desc[blockIdx.y];
  const int ncols = d.np > 0 ? d.nz : d.ncur;
  for (int lc = blockIdx.x * blockDim.x + threadIdx.x; lc < ncols; lc += gridDim.x * blockDim.x) {
    double s = 0.0;
    const double *fc = F + d.oF + lc;
    for (int li = 0; li < d.np; li++) s += fc[(long long)li * d.ldf] * dy[d.row0 + li];
    // x_k is the state the previous stage's dynamics produce: -1.0 in that row
    if (blockIdx.y > 0 && lc < d.ncur) s -= dy[d.row0 - d.ncur + lc];
    out1[d.col0 + lc] = s;
  }
}

// The same two products when the F blocks are the LOCAL ones of a rank (Floc_k = [F_p | F_u], staged_plan.hpp): the rank
// adds what its state columns contribute; the control columns, the -x_{k+1} of the dynamics rows and the -dy of the
// rows that produce x_k are added once, by the rank that owns them (control columns: rank 0).  Both outputs start at
// zero and are summed over the ranks afterwards.
struct DynLoc {
  long long oF;
  int ldf, np, nz, col0, row0, ncur;  // as DynDesc (nz, ncur: of the whole stage)
  int c0, wd, m;                      // own state columns [c0, c0 + wd), controls of the stage
  int with_controls;                  // this rank adds the control columns' share
};
__global__ void __launch_bounds__(256) k_st_dynloc_ax(const DynLoc *__restrict__ desc, const double *__restrict__ F,
                                                      const double *__restrict__ dx, double *__restrict__ out2) {
  const DynLoc d = desc[blockIdx.y];
  const int lane = threadIdx.x & 63;
  const int nloc = d.wd + (d.with_controls ? d.m : 0);
  for (int li = blockIdx.x * 4 + (threadIdx.x >> 6); li < d.np; li += gridDim.x * 4) {
    const double *fr = F + d.oF + (long long)li * d.ldf;
    double s = 0.0;
    for (int j = lane; j < nloc; j += 64) s += fr[j] * dx[d.col0 + (j < d.wd ? d.c0 + j : d.ncur + (j - d.wd))];
    s = kktdev::wave_sum(s);
    // -x_{k+1}[li]: by the rank that owns row li of the next stage's states ... which only the host knows: rank 0 here
    if (lane == 0) out2[d.row0 + li] = s - (d.with_controls ? dx[d.col0 + d.nz + li] : 0.0);
  }
}
__global__ void __launch_bounds__(256) k_st_dynloc_aty(const DynLoc *__restrict__ desc, const double *__restrict__ F,
                                                       const double *__restrict__ dy, double *__restrict__ out1) {
  const DynLoc d = desc[blockIdx.y];
  const int nloc = d.wd + (d.with_controls && d.np > 0 ? d.m : 0);
  for (int lc = blockIdx.x * blockDim.x + threadIdx.x; lc < nloc; lc += gridDim.x * blockDim.x) {
    const int gc = lc < d.wd ? d.c0 + lc : d.ncur + (lc - d.wd);
    double s = 0.0;
    const double *fc = F + d.oF + lc;
    for (int li = 0; li < d.np; li++) s += fc[(long long)li * d.ldf] * dy[d.row0 + li];
    // x_k is the state the previous stage's dynamics produce: -1.0 in that row
    if (blockIdx.y > 0 && gc < d.ncur) s -= dy[d.row0 - d.ncur + gc];
    out1[d.col0 + gc] = s;
  }
}

}  // namespace stg
