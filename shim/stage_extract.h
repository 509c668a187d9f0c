/*
 * stage_extract.h -- the dense per-stage blocks [fx_k fu_k] of a DOCP straight from the row lists of A, as
 * Hqp_IpLQDOCP::update takes them with sp_extract_mat (hqp/Hqp_IpLQDOCP.C:748-755), without ever forming a CSR
 * copy of the dynamics rows (5*10^9 entries at K = 200, nx = 5000: beyond 32-bit counters and row pointers).
 * Header-only and free of Meschach types: `Rows` is any type with
 *     int    len(long long i)            entries of row i
 *     int    col(long long i, int j)     column of its j-th entry (ascending in j)
 *     double val(long long i, int j)
 * (shim/Hqp_IpSpBKPHip.C wraps an SPMAT; tests/c_host/stage_extract_test.cc a generated matrix of C4 size.)
 * Every count and offset on the path is a long long.
 */
#ifndef HQP_STAGE_EXTRACT_H
#define HQP_STAGE_EXTRACT_H

namespace hqpshim {

/* per row of A what hqpkkt_detect_stages needs: length, column of the last entry, of the one before it */
template <class Rows>
inline void staircase_keys(const Rows &A, long long rows, int *row_len, int *last_col, int *prev_col) {
  for (long long i = 0; i < rows; i++) {
    const int n = A.len(i);
    row_len[i] = n;
    last_col[i] = n > 0 ? A.col(i, n - 1) : -1;
    prev_col[i] = n > 1 ? A.col(i, n - 2) : -1;
  }
}

/* Rows [lo, hi) of stage block k: the dynamics rows row0 + li of A, li = local row, all entries but the last go to
 * sink(li, local column, value); the last one must be the -1.0 at column next_col0 + li.  Returns the number of
 * entries delivered, or -1: an entry outside the stage's columns / not the staircase (HQPKKT_E_FORMAT). */
template <class Rows, class Sink>
inline long long stage_rows(const Rows &A, long long row0, int lo, int hi, int col0, int nz, int next_col0, Sink &sink) {
  long long count = 0;
  for (int li = lo; li < hi; li++) {
    const long long i = row0 + li;
    const int n = A.len(i);
    if (n < 2 || A.col(i, n - 1) != next_col0 + li || A.val(i, n - 1) != -1.0) return -1;
    for (int j = 0; j < n - 1; j++) {
      const int c = A.col(i, j) - col0;
      if (c < 0 || c >= nz) return -1;
      sink(li, c, A.val(i, j));
      count++;
    }
  }
  return count;
}

/* sink that writes a row-major block with leading dimension ld (the block must have been zeroed) */
struct DenseSink {
  double *dst;
  long long ld;
  void operator()(int li, int c, double v) { dst[(long long)li * ld + c] = v; }
};

}  // namespace hqpshim
#endif
