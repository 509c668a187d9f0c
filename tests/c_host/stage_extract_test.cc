// CPU test of shim/stage_extract.h (tests/test_c_host.py compiles and runs it): (1) a small DOCP staircase against the
// dense blocks it was generated from; (2) BASELINE configs[3]'s size - K = 200 stages of 5000 states and 50 controls,
// 5.05e9 entries in the dynamics rows - walked through the same code with generated rows and a counting sink: the
// entry count and the largest block offset exceed 32 bits and must come out exactly.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../shim/stage_extract.h"

struct Generated {  // dynamics rows of a DOCP with K stages of nx states and nu controls, dense fx / fu
  int K, nx, nu;
  int len(long long) const { return nx + nu + 1; }
  int col(long long i, int j) const {
    const int k = (int)(i / nx), li = (int)(i % nx);
    return j < nx + nu ? k * (nx + nu) + j : (k + 1) * (nx + nu) + li;
  }
  double val(long long i, int j) const { return j < nx + nu ? 1e-3 * (double)((i * 31 + j * 17) % 1000) + 0.5 : -1.0; }
};
struct CountSink {
  long long count = 0, max_off = 0, ld = 0;
  double sum = 0.0;
  void operator()(int li, int c, double v) {
    const long long off = (long long)li * ld + c;
    if (off > max_off) max_off = off;
    count++, sum += v;
  }
};

int main(int argc, char **argv) {
  {
    Generated A{4, 5, 2};
    const int nz = 7;
    for (int k = 0; k < 4; k++) {
      std::vector<double> blk(5 * 8, 0.0);
      hqpshim::DenseSink s{blk.data(), 8};
      if (hqpshim::stage_rows(A, (long long)k * 5, 0, 5, k * nz, nz, (k + 1) * nz, s) != 35) return 1;
      for (int li = 0; li < 5; li++)
        for (int c = 0; c < nz; c++)
          if (blk[li * 8 + c] != A.val((long long)k * 5 + li, c)) return 2;
    }
    // a row whose last entry is not the -1.0 of the staircase
    struct Bad : Generated {
      double val(long long i, int j) const { return (i == 7 && j == nx + nu) ? -2.0 : Generated::val(i, j); }
    } B;
    B.K = 4, B.nx = 5, B.nu = 2;
    hqpshim::DenseSink s{nullptr, 8};
    std::vector<double> blk(5 * 8, 0.0);
    s.dst = blk.data();
    if (hqpshim::stage_rows(B, 5, 0, 5, nz, nz, 2 * nz, s) != -1) return 3;
    std::vector<int> len(20), last(20), prev(20);
    hqpshim::staircase_keys(A, 20, len.data(), last.data(), prev.data());
    if (len[0] != 8 || last[0] != 7 || prev[0] != 6 || last[19] != 4 * 7 + 4) return 4;
  }
  const bool full = argc > 1 && atoi(argv[1]) != 0;
  const int K = full ? 200 : 200, nx = full ? 5000 : 5000, nu = 50, stages = full ? K : 3;
  Generated A{K, nx, nu};
  long long total = 0, max_off = 0;
  for (int k = K - stages; k < K; k++) {  // (the LAST stages: their row indices and columns are the largest)
    CountSink s;
    s.ld = nx + nu;
    const long long got = hqpshim::stage_rows(A, (long long)k * nx, 0, nx, k * (nx + nu), nx + nu, (k + 1) * (nx + nu), s);
    if (got != (long long)nx * (nx + nu) || s.count != got) return 5;
    total += got;
    if (s.max_off > max_off) max_off = s.max_off;
  }
  const long long want = (long long)stages * nx * (nx + nu);
  if (total != want) return 6;
  if (max_off != (long long)(nx - 1) * (nx + nu) + nx + nu - 1) return 7;
  // offsets of the blocks in one arena, as the library lays them out (multiples of 16 elements)
  long long off = 0;
  for (int k = 0; k < K; k++) off += ((long long)nx * ((nx + nu + 7) / 8 * 8) + 15) / 16 * 16;
  if (off <= 0x7fffffffLL) return 8;  // 5.06e9 elements: the test is pointless if this fits 32 bits
  printf("entries %lld (%s 2^32), arena elements %lld\n", total, total > 0xffffffffLL ? ">" : "<=", off);
  return 0;
}
