// micro-benchmark: variants of the triangle-reading symmetric product (N = 5000, ld = 5008)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef double double2_t __attribute__((ext_vector_type(2)));
__device__ inline double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move0(double v) {
  const long long b = __double_as_longlong(v);
  int lo = (int)(b & 0xffffffffLL), hi = (int)(b >> 32);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
  v += dpp_move0<0xb1, 0xf>(v);
  v += dpp_move0<0x4e, 0xf>(v);
  v += dpp_move0<0x124, 0xf>(v);
  v += dpp_move0<0x128, 0xf>(v);
  v += dpp_move0<0x142, 0xa>(v);
  v += dpp_move0<0x143, 0xc>(v);
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), 63);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// R rows per tile (4 waves, R/4 rows each), C = 128 * P columns (P pairs of 16-byte loads per lane and row), U rows in flight
template <int R, int P, int U, int MODE>
__global__ void __launch_bounds__(256) k_tiles(const double *V, long long ldv, int N, const double *x, double *rowpart, double *colpart, int ratio) {
  constexpr int C = 128 * P;
  __shared__ double red[4][C];
  const int t = blockIdx.x;
  // ratio = C / R row tiles share a count of column tiles: g + 1 each for the group g
  int gq = (int)((sqrt(1.0 + 8.0 * t / ratio) - 1.0) * 0.5);
  while (ratio * gq * (gq + 1) / 2 > t) gq--;
  while (ratio * (gq + 1) * (gq + 2) / 2 <= t) gq++;
  const int rem = t - ratio * gq * (gq + 1) / 2, bi = ratio * gq + rem / (gq + 1), bj = rem % (gq + 1), r0 = bi * R, c0 = bj * C;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool diag = c0 + C - 1 >= r0;
  int jj[P];
  double xa[P], xb[P], ca[P], cb[P];
#pragma unroll
  for (int p = 0; p < P; p++) {
    jj[p] = c0 + 2 * lane + 128 * p;
    xa[p] = jj[p] < N ? x[jj[p]] : 0.0, xb[p] = jj[p] + 1 < N ? x[jj[p] + 1] : 0.0;
    ca[p] = cb[p] = 0.0;
  }
  const int ra = r0 + wave * (R / 4), rb = min(N, ra + R / 4);
  const double *row = V + (long long)ra * ldv;
  double *rp = rowpart + (long long)bj * N;
  for (int i = ra; i < rb; i += U) {
    double2_t v[U][P];
    double xi[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const bool live = i + u < rb;
#pragma unroll
      for (int p = 0; p < P; p++) v[u][p] = (live && jj[p] < N) ? *(const double2_t *)(row + (long long)u * ldv + jj[p]) : double2_t{0.0, 0.0};
      xi[u] = live ? x[i + u] : 0.0;
    }
    row += U * ldv;
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int ii = i + u;
      double s = 0.0;
#pragma unroll
      for (int p = 0; p < P; p++) {
        double a = v[u][p].x, b = jj[p] + 1 < N ? v[u][p].y : 0.0;
        if (diag) {
          if (jj[p] > ii) a = 0.0;
          if (jj[p] + 1 > ii) b = 0.0;
        }
        s += a * xa[p] + b * xb[p];
        if (diag) {
          if (jj[p] == ii) a = 0.0;
          if (jj[p] + 1 == ii) b = 0.0;
        }
        ca[p] += a * xi[u], cb[p] += b * xi[u];
      }
      if (MODE == 0) {
        s = wave_sum(s);
        if (lane == 0 && ii < rb) rp[ii] = s;
      } else if (MODE == 2) {
        s = wave_sum_dpp(s);
        if (lane == 0 && ii < rb) rp[ii] = s;
      } else
        ca[0] += s;
    }
  }
#pragma unroll
  for (int p = 0; p < P; p++) red[wave][128 * p + 2 * lane] = ca[p], red[wave][128 * p + 2 * lane + 1] = cb[p];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256)
    if (c0 + c < N) colpart[(long long)bi * N + c0 + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}
// reference: the rows form (one wavefront per row)
__global__ void __launch_bounds__(256) k_rows(const double *A, long long lda, int N, const double *x, double *y) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  const double2_t *a2 = (const double2_t *)(A + (long long)row * lda);
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  int j = lane;
  const int n2 = N >> 1;
  for (; j + 192 < n2; j += 256) {
    const double2_t v0 = a2[j], v1 = a2[j + 64], v2 = a2[j + 128], v3 = a2[j + 192];
    s0 += v0.x * x[2 * j] + v0.y * x[2 * j + 1];
    s1 += v1.x * x[2 * (j + 64)] + v1.y * x[2 * (j + 64) + 1];
    s2 += v2.x * x[2 * (j + 128)] + v2.y * x[2 * (j + 128) + 1];
    s3 += v3.x * x[2 * (j + 192)] + v3.y * x[2 * (j + 192) + 1];
  }
  for (; j < n2; j += 64) {
    const double2_t v0 = a2[j];
    s0 += v0.x * x[2 * j] + v0.y * x[2 * j + 1];
  }
  double s = wave_sum((s0 + s1) + (s2 + s3));
  if (lane == 0) y[row] = s;
}
template <int R, int P, int U, int MODE>
static void run(const char *name, const double *V, long long ld, int N, const double *x, double *rp, double *cp, int nmat) {
  constexpr int C = 128 * P;
  const int ratio = C / R, nrt = (N + R - 1) / R;
  long long tiles = 0;
  for (int bi = 0; bi < nrt; bi++) tiles += bi / ratio + 1;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const int reps = 20;
  for (int w = 0; w < 2; w++) {
    hipEventRecord(e0);
    for (int r = 0; r < reps; r++)
      k_tiles<R, P, U, MODE><<<(unsigned)tiles, 256>>>(V + (long long)(r % nmat) * ld * N, ld, N, x, rp, cp, ratio);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s tiles %5lld  %.1f us  (%.0f GB/s of the triangle's %.0f MB)\n", name, tiles, ms * 1e3 / reps,
         tiles * (double)R * C * 8 / (ms * 1e-3 / reps) / 1e9, tiles * (double)R * C * 8 / 1e6);
}
int main() {
  const int N = 5000, nmat = 6;
  const long long ld = 5008;
  double *V, *x, *rp, *cp, *y;
  hipMalloc(&V, sizeof(double) * ld * N * nmat), hipMalloc(&x, sizeof(double) * ld), hipMalloc(&y, sizeof(double) * ld);
  hipMalloc(&rp, sizeof(double) * 64 * ld), hipMalloc(&cp, sizeof(double) * 320 * ld);
  hipMemset(V, 0, sizeof(double) * ld * N * nmat), hipMemset(x, 0, sizeof(double) * ld);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int w = 0; w < 2; w++) {
    hipEventRecord(e0);
    for (int r = 0; r < 20; r++) k_rows<<<(N + 3) / 4, 256>>>(V + (long long)(r % nmat) * ld * N, ld, N, x, y);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("rows form (whole matrix)     %.1f us (%.0f GB/s)\n", ms * 1e3 / 20, 8.0 * N * N / (ms * 1e-3 / 20) / 1e9);
  run<64, 4, 4, 0>("R64 C512 U4", V, ld, N, x, rp, cp, nmat);
  run<64, 4, 4, 2>("R64 C512 U4 dpp", V, ld, N, x, rp, cp, nmat);
  run<64, 4, 4, 1>("R64 C512 U4 none", V, ld, N, x, rp, cp, nmat);
  run<64, 2, 4, 2>("R64 C256 U4 dpp", V, ld, N, x, rp, cp, nmat);
  run<64, 2, 8, 2>("R64 C256 U8 dpp", V, ld, N, x, rp, cp, nmat);
  run<32, 4, 4, 2>("R32 C512 U4 dpp", V, ld, N, x, rp, cp, nmat);
  run<32, 4, 8, 2>("R32 C512 U8 dpp", V, ld, N, x, rp, cp, nmat);
  run<64, 4, 2, 2>("R64 C512 U2 dpp", V, ld, N, x, rp, cp, nmat);
  run<64, 4, 8, 2>("R64 C512 U8 dpp", V, ld, N, x, rp, cp, nmat);
  run<96, 4, 4, 2>("R96 C512 U4 dpp", V, ld, N, x, rp, cp, nmat);
  return 0;
}
