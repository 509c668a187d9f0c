/* Plain-C host of the C ABI (include/hqpkkt.h): no Python, no C++, no HIP headers.
 * Builds a small banded QP, runs analyze / set_values / factor / solve / residual and
 * the device-resident Mehrotra loop, prints one line that the test parses.
 * On a machine without a GPU every numeric entry must fail with HQPKKT_E_DEVICE (100):
 * the process prints the status and exits 3. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hqpkkt.h"

static double urand(unsigned long long *s) {
  *s ^= *s << 13, *s ^= *s >> 7, *s ^= *s << 17;
  return (double)(*s >> 11) / 9007199254740992.0;
}

int main(void) {
  const int n = 600, band = 6, me = n / 2, m = n;
  unsigned long long seed = 88172645463325252ULL;
  /* Q: upper band, diagonally dominant; A: rows of width band; C = I */
  int *Qp = malloc(sizeof(int) * (n + 1)), *Qi = malloc(sizeof(int) * n * (band + 1));
  double *Qx = malloc(sizeof(double) * n * (band + 1));
  int nq = 0;
  for (int i = 0; i < n; i++) {
    Qp[i] = nq;
    Qi[nq] = i, Qx[nq++] = 4.0 * band;
    for (int j = i + 1; j <= i + band && j < n; j++) Qi[nq] = j, Qx[nq++] = urand(&seed) - 0.5;
  }
  Qp[n] = nq;
  int *Ap = malloc(sizeof(int) * (me + 1)), *Ai = malloc(sizeof(int) * me * band);
  double *Ax = malloc(sizeof(double) * me * band);
  int na = 0;
  for (int r = 0; r < me; r++) {
    Ap[r] = na;
    for (int j = 2 * r; j < 2 * r + band && j < n; j++) Ai[na] = j, Ax[na++] = urand(&seed) - 0.5;
  }
  Ap[me] = na;
  int *Cp = malloc(sizeof(int) * (m + 1)), *Ci = malloc(sizeof(int) * m);
  double *Cx = malloc(sizeof(double) * m);
  for (int r = 0; r < m; r++) Cp[r] = r, Ci[r] = r, Cx[r] = 1.0;
  Cp[m] = m;

  hqpkkt_opts o;
  hqpkkt_t *h = NULL;
  int sbw = -1, st;
  hqpkkt_default_opts(&o);
  o.mode = HQPKKT_MODE_FULL;
  if ((st = hqpkkt_create(&o, &h)) || (st = hqpkkt_analyze(h, n, me, m, Qp, Qi, Ap, Ai, Cp, Ci, &sbw))) {
    printf("C_ABI status %d in create/analyze: %s\n", st, hqpkkt_strerror(st));
    return 2;
  }
  if ((st = hqpkkt_set_values(h, Qx, Ax, Cx))) {
    printf("C_ABI status %d in set_values: %s\n", st, hqpkkt_strerror(st));
    return st == HQPKKT_E_DEVICE ? 3 : 2;
  }
  double *z = malloc(sizeof(double) * m), *w = malloc(sizeof(double) * m);
  double *r1 = malloc(sizeof(double) * n), *r2 = malloc(sizeof(double) * me), *r3 = malloc(sizeof(double) * m),
         *r4 = malloc(sizeof(double) * m);
  double *dx = calloc(n, sizeof(double)), *dy = calloc(me, sizeof(double)), *dz = calloc(m, sizeof(double)),
         *dw = calloc(m, sizeof(double));
  for (int i = 0; i < m; i++) z[i] = 0.1 + urand(&seed), w[i] = 0.1 + urand(&seed), r3[i] = urand(&seed) - 0.5, r4[i] = urand(&seed) - 0.5;
  for (int i = 0; i < n; i++) r1[i] = urand(&seed) - 0.5;
  for (int i = 0; i < me; i++) r2[i] = urand(&seed) - 0.5;
  double res = -1.0, res2 = -1.0;
  if ((st = hqpkkt_factor(h, z, w)) || (st = hqpkkt_solve(h, z, w, r1, r2, r3, r4, dx, dy, dz, dw, &res)) ||
      (st = hqpkkt_residual(h, z, w, r1, r2, r3, r4, dx, dy, dz, dw, &res2))) {
    printf("C_ABI status %d in factor/solve: %s\n", st, hqpkkt_strerror(st));
    return 2;
  }
  /* the whole QP: min 1/2 x'Qx + c'x  s.t.  Ax + b = 0,  Cx + d >= 0 */
  double *c = malloc(sizeof(double) * n), *b = calloc(me, sizeof(double)), *d = malloc(sizeof(double) * m);
  double *x = calloc(n, sizeof(double)), *y = calloc(me, sizeof(double));
  for (int i = 0; i < n; i++) c[i] = urand(&seed) - 0.5;
  for (int i = 0; i < m; i++) d[i] = 1.0;
  hqpkkt_ip_opts io;
  hqpkkt_ip_result ir;
  hqpkkt_default_ip_opts(&io);
  io.norm_data = 4.0 * band + band;  /* a bound on the data norms is enough for the test */
  if ((st = hqpkkt_mehrotra(h, &io, c, b, d, x, y, z, w, &ir))) {
    printf("C_ABI status %d in mehrotra: %s\n", st, hqpkkt_strerror(st));
    return 2;
  }
  double zmin = 1e300, cmin = 1e300;
  for (int i = 0; i < m; i++) zmin = fmin(zmin, z[i]), cmin = fmin(cmin, x[i] + d[i]);
  hqpkkt_stats s;
  hqpkkt_get_stats(h, &s);
  printf("C_ABI ok sbw %d dim %d res %.3e res2 %.3e ip_result %d ip_iters %d mu %.3e zmin %.3e cmin %.3e\n", sbw,
         s.dim, res, res2, ir.result, ir.iters, ir.mu, zmin, cmin);
  hqpkkt_destroy(h);

  /* ---- the multistage plugin's dense hand-over (Hqp_IpLQDOCP semantics, HQPKKT_MODE_STAGED): K stages of nx states
   * and nu controls, the dynamics as K row-major blocks [fx fu], x_0 fixed by nx one-entry equality rows, -1 <= u <= 1 */
  {
    enum { K = 6, NX = 40, NU = 3, NZ = NX + NU };
    const int nt = K * NZ + NX, me_rest = NX, mm = 2 * K * NU, met = K * NX + me_rest;
    int nxs[K + 1], nus[K];
    for (int k = 0; k <= K; k++) nxs[k] = NX;
    for (int k = 0; k < K; k++) nus[k] = NU;
    int *qp = malloc(sizeof(int) * (nt + 1)), *qi = malloc(sizeof(int) * nt);
    double *qx = malloc(sizeof(double) * nt);
    for (int i = 0; i < nt; i++) qp[i] = i, qi[i] = i, qx[i] = (i < K * NZ && i % NZ >= NX) ? 0.1 : 1.0;
    qp[nt] = nt;
    int ep[NX + 1], ei[NX];
    double ex[NX];
    for (int i = 0; i < NX; i++) ep[i] = i, ei[i] = i, ex[i] = 1.0;
    ep[NX] = NX;
    int *cp = malloc(sizeof(int) * (mm + 1)), *ci = malloc(sizeof(int) * mm);
    double *cx = malloc(sizeof(double) * mm);
    for (int r = 0; r < mm; r++) {
      const int u = r % (K * NU);
      cp[r] = r, ci[r] = (u / NU) * NZ + NX + u % NU, cx[r] = r < K * NU ? 1.0 : -1.0;
    }
    cp[mm] = mm;
    double *Fb = malloc(sizeof(double) * K * NX * NZ);
    const double *Fp[K];
    long long ldF[K];
    for (int k = 0; k < K; k++) {
      Fp[k] = Fb + (size_t)k * NX * NZ, ldF[k] = NZ;
      for (int i = 0; i < NX * NZ; i++) Fb[(size_t)k * NX * NZ + i] = (urand(&seed) - 0.5) * ((i % NZ) < NX ? 0.25 : 1.0);
    }
    hqpkkt_t *hs = NULL;
    hqpkkt_default_opts(&o);
    o.mode = HQPKKT_MODE_STAGED;
    if ((st = hqpkkt_create(&o, &hs)) ||
        (st = hqpkkt_analyze_staged(hs, K, nxs, nus, nt, me_rest, mm, qp, qi, ep, ei, cp, ci)) ||
        (st = hqpkkt_set_values_staged(hs, qx, Fp, ldF, ex, cx))) {
      printf("C_ABI status %d in the staged hand-over: %s\n", st, hqpkkt_strerror(st));
      return 2;
    }
    double *zs = malloc(sizeof(double) * mm), *ws = malloc(sizeof(double) * mm), *s3 = malloc(sizeof(double) * mm),
           *s4 = malloc(sizeof(double) * mm), *s1 = malloc(sizeof(double) * nt), *s2 = malloc(sizeof(double) * met);
    double *ex1 = calloc(nt, sizeof(double)), *ey = calloc(met, sizeof(double)), *ez = calloc(mm, sizeof(double)),
           *ew = calloc(mm, sizeof(double));
    for (int i = 0; i < mm; i++) zs[i] = 0.1 + urand(&seed), ws[i] = 0.1 + urand(&seed), s3[i] = urand(&seed) - 0.5, s4[i] = urand(&seed) - 0.5;
    for (int i = 0; i < nt; i++) s1[i] = urand(&seed) - 0.5;
    for (int i = 0; i < met; i++) s2[i] = urand(&seed) - 0.5;
    double sres = -1.0, sres2 = -1.0;
    if ((st = hqpkkt_factor(hs, zs, ws)) || (st = hqpkkt_solve(hs, zs, ws, s1, s2, s3, s4, ex1, ey, ez, ew, &sres)) ||
        (st = hqpkkt_residual(hs, zs, ws, s1, s2, s3, s4, ex1, ey, ez, ew, &sres2))) {
      printf("C_ABI status %d in the staged factor/solve: %s\n", st, hqpkkt_strerror(st));
      return 2;
    }
    hqpkkt_get_stats(hs, &s);
    printf("C_ABI staged ok stages %d dim %d res %.3e res2 %.3e\n", s.n_levels - 1, s.dim, sres, sres2);
    hqpkkt_destroy(hs);
  }
  return 0;
}
