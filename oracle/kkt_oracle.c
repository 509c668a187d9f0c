/*
 * TEST INFRASTRUCTURE -- CPU oracle, never part of the product path.
 *
 * Plain-C restatement of the interior-point KKT linear-system path of
 * omuses/hqp, written against dense-upper storage with row extents instead of
 * the reference's row-list SPMAT, so that it shares no code with it:
 *
 *   rcm_*          hqp/sprcm.C:62-211 (scan), :226-384 (order), :391-420 (sbw)
 *   assemble_*     hqp/Hqp_IpSpBKP.C:117-136, hqp/Hqp_IpRedSpBKP.C:268-278,
 *                  meschach/addon2_hqp.c:1044-1110 (block scatter, upper only)
 *   kkto_factor    hqp/Hqp_IpSpBKP.C:139-180, hqp/Hqp_IpRedSpBKP.C:104-181,
 *                  :281-320 (w/z insertion, C'ZW^-1C, symmetric scaling)
 *   bkp_factor     hqp/spBKP.C:369-645 (Bunch-Kaufman-Parlett pivot rule,
 *                  interchange, 1x1 / 2x2 elimination, pivot encoding)
 *   bkp_solve      hqp/spBKP.C:647-797
 *   kkto_step      hqp/Hqp_IpSpBKP.C:183-218, hqp/Hqp_IpRedSpBKP.C:323-368
 *   kkto_residuum  hqp/Hqp_IpMatrix.C:131-178
 *   kkto_solve     hqp/Hqp_IpMatrix.C:65-128 (<= 5 refinement rounds, back-off)
 *
 * PARITY PIN: the reference's own tests hold no golden vectors for this path
 * (SURVEY.md section 8(c)); this oracle is pinned against the reference itself
 * run in the build container (oracle/_ref/libhqpref.so, tests/test_oracle_vs_ref.py)
 * and against tests/golden/ fixtures generated from it by tests/golden/make_golden.py.
 *
 * Storage is O(dim^2): meant for dim <= ~6000 (test sizes).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define KKTO_FULL 0    /* Hqp_IpSpBKP    : dim = n + me + m */
#define KKTO_REDUCED 1 /* Hqp_IpRedSpBKP : dim = n + me     */
#define KKTO_E_SING 4  /* meschach/err.h:88 */

typedef struct {
  int mode, n, me, m, dim, sbw;
  double tol, eps;
  /* QP blocks (CSR copies) */
  int *Qp, *Qi, *Ap, *Ai, *Cp, *Ci;
  double *Qx, *Ax, *Cx;
  int *qp2j, *j2qp; /* _QP2J / _J2QP */
  double *Jraw, *J; /* dense dim x dim, row-major, upper triangle used */
  int *last_raw, *last; /* row extents: largest column that may be non-zero */
  int *pivot;
  double *scale; /* m (FULL) or n (REDUCED) */
  double *zw;    /* REDUCED: z./w */
  double *rhs, *sol;
  double *t1, *t2, *t3, *t4, *e1, *e2, *e3, *e4; /* _r1.._r4, _dx.._dw */
} kkto;

/* ---------------------------------------------------------------- helpers */
static int *icopy(const int *s, long k) {
  int *d = (int *)malloc(sizeof(int) * (size_t)(k > 0 ? k : 1));
  if (k > 0) memcpy(d, s, sizeof(int) * (size_t)k);
  return d;
}
static double *dcopy(const double *s, long k) {
  double *d = (double *)malloc(sizeof(double) * (size_t)(k > 0 ? k : 1));
  if (k > 0) memcpy(d, s, sizeof(double) * (size_t)k);
  return d;
}
static double *dzero(long k) {
  return (double *)calloc((size_t)(k > 0 ? k : 1), sizeof(double));
}
static double dmin(double a, double b) { return a < b ? a : b; }

/* ------------------------------------------------------------------- RCM */
typedef struct {
  int node, deg, deg2;
} lvl_node;

static int lvl_less_eq(const lvl_node *a, const lvl_node *b) {
  /* ordering of hqp/sprcm.C:42-50: by deg, then deg2 */
  if (a->deg != b->deg) return a->deg < b->deg;
  return a->deg2 <= b->deg2;
}
/* stable merge sort; glibc's qsort (used by hqp/sprcm.C:333) is a stable
 * merge sort whenever its scratch buffer can be allocated, so ties resolve
 * identically */
static void lvl_sort(lvl_node *a, int k, lvl_node *tmp) {
  if (k < 2) return;
  int h = k / 2;
  lvl_sort(a, h, tmp);
  lvl_sort(a + h, k - h, tmp);
  int i = 0, j = h, o = 0;
  while (i < h && j < k) tmp[o++] = lvl_less_eq(&a[i], &a[j]) ? a[i++] : a[j++];
  while (i < h) tmp[o++] = a[i++];
  while (j < k) tmp[o++] = a[j++];
  memcpy(a, tmp, sizeof(lvl_node) * (size_t)k);
}

/* adjacency of the graph of [Q A' C'; A; C] (C absent when Cp == NULL);
 * neighbour lists are filled in the visiting order of hqp/sprcm.C:151-208 */
static void rcm_scan(int n, const int *Qp, const int *Qi, int me, const int *Ap,
                     const int *Ai, int m, const int *Cp, const int *Ci,
                     int **start_out, int **neigh_out) {
  int dim = n + me + (Cp ? m : 0);
  int *deg = (int *)calloc((size_t)dim + 1, sizeof(int));
  int *start = (int *)malloc(sizeof(int) * ((size_t)dim + 1));
  for (int i = 0; i < n; i++)
    for (int k = Qp[i]; k < Qp[i + 1]; k++)
      if (Qi[k] > i) deg[i]++, deg[Qi[k]]++;
  for (int r = 0; r < me; r++)
    for (int k = Ap[r]; k < Ap[r + 1]; k++) deg[n + r]++, deg[Ai[k]]++;
  if (Cp)
    for (int r = 0; r < m; r++)
      for (int k = Cp[r]; k < Cp[r + 1]; k++) deg[n + me + r]++, deg[Ci[k]]++;
  int off = 0;
  for (int i = 0; i < dim; i++) start[i] = off, off += deg[i], deg[i] = 0;
  start[dim] = off;
  int *neigh = (int *)malloc(sizeof(int) * (size_t)(off > 0 ? off : 1));
#define LINK(a, b) (neigh[start[a] + deg[a]++] = (b), neigh[start[b] + deg[b]++] = (a))
  for (int i = 0; i < n; i++)
    for (int k = Qp[i]; k < Qp[i + 1]; k++)
      if (Qi[k] > i) LINK(i, Qi[k]);
  for (int r = 0; r < me; r++)
    for (int k = Ap[r]; k < Ap[r + 1]; k++) LINK(n + r, Ai[k]);
  if (Cp)
    for (int r = 0; r < m; r++)
      for (int k = Cp[r]; k < Cp[r + 1]; k++) LINK(n + me + r, Ci[k]);
#undef LINK
  free(deg);
  *start_out = start;
  *neigh_out = neigh;
}

/* reverse Cuthill-McKee with the root iteration of hqp/sprcm.C:262-360:
 * repeat the level structure from the min-degree node of the last level while
 * the number of levels grows; "degree" counts not-yet-numbered neighbours */
static void rcm_order(int dim, const int *start, const int *neigh, int *order) {
  char *marks = (char *)malloc((size_t)dim + 1);
  char *glob = (char *)malloc((size_t)dim + 1);
  lvl_node *lv = (lvl_node *)calloc((size_t)dim + 1, sizeof(lvl_node));
  lvl_node *tmp = (lvl_node *)malloc(sizeof(lvl_node) * ((size_t)dim + 1));
  int *deg = (int *)malloc(sizeof(int) * ((size_t)dim + 1));
  for (int i = 0; i < dim; i++) glob[i] = 1, deg[i] = start[i + 1] - start[i];
  int root = 0, count = 0;
  while (root < dim) {
    int nlevels = 0, nlevels_old, cluster = count, lb = 0, le = 0;
    do {
      count = cluster;
      memcpy(marks, glob, (size_t)dim);
      nlevels_old = nlevels;
      nlevels = 0;
      lb = le = count;
      lv[count++].node = root;
      marks[root] = 0;
      for (int k = start[root]; k < start[root + 1]; k++) deg[neigh[k]]--;
      do {
        lb = le;
        le = count;
        nlevels++;
        for (int i = lb; i < le; i++) {
          int v = lv[i].node;
          for (int j = start[v]; j < start[v + 1]; j++) {
            int u = neigh[j];
            if (marks[u]) {
              lv[count++].node = u;
              marks[u] = 0;
              for (int k = start[u]; k < start[u + 1]; k++) deg[neigh[k]]--;
            }
          }
          /* the reference re-sorts the whole new level after each parent */
          if (count - le > 1) {
            for (int k = le; k < count; k++) {
              int u = lv[k].node, d2 = 0;
              lv[k].deg = deg[u];
              for (int j = start[u]; j < start[u + 1]; j++)
                if (marks[neigh[j]]) d2 += deg[neigh[j]];
              lv[k].deg2 = d2;
            }
            lvl_sort(lv + le, count - le, tmp);
          }
        }
      } while (count > le);
      root = lv[lb].node;
      int d = deg[root];
      for (int i = lb + 1; i < le; i++)
        if (deg[lv[i].node] < d) root = lv[i].node, d = deg[root];
      for (int i = 0; i < dim; i++) deg[i] = start[i + 1] - start[i];
    } while (nlevels > nlevels_old);
    root = dim;
    for (int i = 0; i < dim; i++) {
      glob[i] = marks[i];
      if (marks[i]) root = i;
    }
  }
  for (int i = 0; i < dim; i++) order[lv[i].node] = dim - i - 1;
  free(marks);
  free(glob);
  free(lv);
  free(tmp);
  free(deg);
}

static int rcm_sbw(int dim, const int *start, const int *neigh, const int *order) {
  int ub = 0;
  for (int v = 0; v < dim; v++) {
    int far = order[v];
    for (int k = start[v]; k < start[v + 1]; k++)
      if (order[neigh[k]] > far) far = order[neigh[k]];
    if (far - order[v] > ub) ub = far - order[v];
  }
  return ub;
}

/* ------------------------------------------------------------- lifecycle */
kkto *kkto_create(int mode, double tol, double eps) {
  kkto *h = (kkto *)calloc(1, sizeof(kkto));
  h->mode = mode;
  h->tol = tol;
  h->eps = eps;
  h->sbw = -1;
  return h;
}

static void free_all(kkto *h) {
  void *p[] = {h->Qp, h->Qi, h->Ap, h->Ai, h->Cp, h->Ci, h->Qx, h->Ax, h->Cx, h->qp2j,
               h->j2qp, h->Jraw, h->J, h->last_raw, h->last, h->pivot, h->scale, h->zw,
               h->rhs, h->sol, h->t1, h->t2, h->t3, h->t4, h->e1, h->e2, h->e3, h->e4};
  for (unsigned k = 0; k < sizeof(p) / sizeof(p[0]); k++) free(p[k]);
}

void kkto_destroy(kkto *h) {
  if (!h) return;
  free_all(h);
  free(h);
}

static void put(kkto *h, int a, int b, double v) {
  /* store into the upper triangle of the permuted matrix */
  int r = a < b ? a : b, c = a < b ? b : a;
  h->Jraw[(size_t)r * h->dim + c] = v;
  if (c > h->last_raw[r]) h->last_raw[r] = c;
}

int kkto_update(kkto *h, const double *Qx, const double *Ax, const double *Cx) {
  int n = h->n, me = h->me, m = h->m, dim = h->dim;
  const int *P = h->qp2j;
  if (Qx != h->Qx) memcpy(h->Qx, Qx, sizeof(double) * (size_t)h->Qp[n]);
  if (Ax != h->Ax) memcpy(h->Ax, Ax, sizeof(double) * (size_t)h->Ap[me]);
  if (Cx != h->Cx) memcpy(h->Cx, Cx, sizeof(double) * (size_t)h->Cp[m]);
  memset(h->Jraw, 0, sizeof(double) * (size_t)dim * dim);
  for (int i = 0; i < dim; i++) h->last_raw[i] = i;
  /* -Q, upper triangle of Q only (meschach/addon2_hqp.c:1066-1087) */
  for (int i = 0; i < n; i++)
    for (int k = h->Qp[i]; k < h->Qp[i + 1]; k++)
      if (h->Qi[k] >= i) put(h, P[i], P[h->Qi[k]], -1.0 * h->Qx[k]);
  /* A and A' each contribute the entries that land in the upper triangle;
   * together: every entry exactly once */
  for (int r = 0; r < me; r++)
    for (int k = h->Ap[r]; k < h->Ap[r + 1]; k++) put(h, P[n + r], P[h->Ai[k]], h->Ax[k]);
  if (h->mode == KKTO_FULL) {
    for (int r = 0; r < m; r++)
      for (int k = h->Cp[r]; k < h->Cp[r + 1]; k++)
        put(h, P[n + me + r], P[h->Ci[k]], h->Cx[k]);
    /* placeholder on the slack diagonal (hqp/Hqp_IpSpBKP.C:129-132) */
    for (int j = 0; j < m; j++) put(h, P[n + me + j], P[n + me + j], 1.0);
  }
  return 0;
}

int kkto_init(kkto *h, int n, int me, int m, const int *Qp, const int *Qi, const double *Qx,
              const int *Ap, const int *Ai, const double *Ax, const int *Cp, const int *Ci,
              const double *Cx) {
  free_all(h);
  h->n = n, h->me = me, h->m = m;
  int dim = h->dim = (h->mode == KKTO_FULL) ? n + me + m : n + me;
  h->Qp = icopy(Qp, n + 1), h->Qi = icopy(Qi, Qp[n]), h->Qx = dcopy(Qx, Qp[n]);
  h->Ap = icopy(Ap, me + 1), h->Ai = icopy(Ai, Ap[me]), h->Ax = dcopy(Ax, Ap[me]);
  h->Cp = icopy(Cp, m + 1), h->Ci = icopy(Ci, Cp[m]), h->Cx = dcopy(Cx, Cp[m]);
  int *start, *neigh;
  if (h->mode == KKTO_FULL) {
    rcm_scan(n, Qp, Qi, me, Ap, Ai, m, Cp, Ci, &start, &neigh);
  } else {
    /* pattern of Q + C'C (hqp/Hqp_IpRedSpBKP.C:203-234): two x-columns are
     * linked when they share a row of C */
    char *hit = (char *)calloc((size_t)n * (size_t)n, 1);
    for (int i = 0; i < n; i++)
      for (int k = Qp[i]; k < Qp[i + 1]; k++)
        if (Qi[k] > i) hit[(size_t)i * n + Qi[k]] = 1;
    for (int r = 0; r < m; r++)
      for (int a = Cp[r]; a < Cp[r + 1]; a++)
        for (int b = a + 1; b < Cp[r + 1]; b++) {
          int lo = Ci[a] < Ci[b] ? Ci[a] : Ci[b], hi = Ci[a] < Ci[b] ? Ci[b] : Ci[a];
          if (lo != hi) hit[(size_t)lo * n + hi] = 1;
        }
    int *Hp = (int *)calloc((size_t)n + 1, sizeof(int));
    long nn = 0;
    for (int i = 0; i < n; i++) {
      for (int j = i + 1; j < n; j++) nn += hit[(size_t)i * n + j];
      Hp[i + 1] = (int)nn;
    }
    int *Hi = (int *)malloc(sizeof(int) * (size_t)(nn > 0 ? nn : 1));
    nn = 0;
    for (int i = 0; i < n; i++)
      for (int j = i + 1; j < n; j++)
        if (hit[(size_t)i * n + j]) Hi[nn++] = j;
    rcm_scan(n, Hp, Hi, me, Ap, Ai, 0, NULL, NULL, &start, &neigh);
    free(hit), free(Hp), free(Hi);
  }
  h->qp2j = (int *)malloc(sizeof(int) * (size_t)(dim + 1));
  h->j2qp = (int *)malloc(sizeof(int) * (size_t)(dim + 1));
  rcm_order(dim, start, neigh, h->qp2j);
  h->sbw = rcm_sbw(dim, start, neigh, h->qp2j);
  for (int i = 0; i < dim; i++) h->j2qp[h->qp2j[i]] = i;
  free(start), free(neigh);
  h->Jraw = dzero((long)dim * dim), h->J = dzero((long)dim * dim);
  h->last_raw = (int *)calloc((size_t)dim + 1, sizeof(int));
  h->last = (int *)calloc((size_t)dim + 1, sizeof(int));
  h->pivot = (int *)calloc((size_t)dim + 1, sizeof(int));
  h->scale = dzero(h->mode == KKTO_FULL ? m : n);
  h->zw = dzero(m);
  h->rhs = dzero(dim), h->sol = dzero(dim);
  h->t1 = dzero(n), h->t2 = dzero(me), h->t3 = dzero(m), h->t4 = dzero(m);
  h->e1 = dzero(n), h->e2 = dzero(me), h->e3 = dzero(m), h->e4 = dzero(m);
  return kkto_update(h, h->Qx, h->Ax, h->Cx);
}

/* --------------------------------------------------- BKP factor and solve */
#define AT(i, j) a[(size_t)(i) * n + (j)]

/* symmetric interchange of p < q inside the reduced matrix whose first row is
 * i0; rows above i0 (finished multiplier rows) stay in the numbering of their
 * own elimination step, as in hqp/spBKP.C:205-366 */
static void sym_swap(double *a, int n, int *last, int i0, int p, int q) {
  double t;
  for (int k = i0; k < p; k++) {
    t = AT(k, p), AT(k, p) = AT(k, q), AT(k, q) = t;
    if (last[k] < q && AT(k, q) != 0.0) last[k] = q;
  }
  t = AT(p, p), AT(p, p) = AT(q, q), AT(q, q) = t;
  for (int k = p + 1; k < q; k++) {
    t = AT(p, k), AT(p, k) = AT(k, q), AT(k, q) = t;
    if (last[k] < q) last[k] = q;
  }
  int lp = last[p], lq = last[q], hi = lp > lq ? lp : lq;
  for (int k = q + 1; k <= hi; k++) t = AT(p, k), AT(p, k) = AT(q, k), AT(q, k) = t;
  last[p] = hi > q ? hi : q;
  last[q] = hi;
}

static int bkp_factor(double *a, int n, int *last, int *pivot, double tol) {
  const double alpha = tol * 0.6403882032022076; /* tol (1+sqrt 17)/8 */
  for (int i = 0; i < n; i++) pivot[i] = i;
  int i = 0;
  while (i < n - 1) {
    int ip1 = i + 1, j = i, one = 0;
    double aii = fabs(AT(i, i)), lambda = aii, sigma, t;
    for (int c = ip1; c <= last[i]; c++)
      if ((t = fabs(AT(i, c))) > lambda) lambda = t, j = c; /* first max wins */
    if (aii >= alpha * lambda) one = 1;
    if (!one) {
      sigma = lambda;
      for (int k = ip1; k < j; k++)
        if (j <= last[k] && (t = fabs(AT(k, j))) > sigma) sigma = t;
      for (int c = j + 1; c <= last[j]; c++)
        if ((t = fabs(AT(j, c))) > sigma) sigma = t;
      if (sigma * aii >= alpha * lambda * lambda)
        one = 1;
      else if (fabs(AT(j, j)) >= alpha * sigma) {
        sym_swap(a, n, last, i, i, j);
        pivot[i] = j;
        one = 1;
      }
    }
    if (one) {
      double d = AT(i, i);
      if (d != 0.0) {
        int li = last[i];
        for (int c = ip1; c <= li; c++) {
          double s = AT(i, c) / d;
          if (s != 0.0) {
            for (int k = c; k <= li; k++) AT(c, k) = AT(c, k) + (-s) * AT(i, k);
            if (last[c] < li) last[c] = li;
          }
          AT(i, c) = s;
        }
      }
      i = ip1;
      continue;
    }
    /* 2x2 pivot on (i, i+1) after moving j to i+1 (hqp/spBKP.C:489-599) */
    if (j > ip1) sym_swap(a, n, last, i, ip1, j);
    pivot[i] = j;
    pivot[ip1] = 0;
    {
      double d11 = AT(i, i), d12 = AT(i, ip1), d22 = AT(ip1, ip1);
      double det = d11 * d22 - d12 * d12;
      d11 /= det, d12 /= det, d22 /= det;
      int hi = last[i] > last[ip1] ? last[i] : last[ip1];
      for (int c = i + 2; c <= hi; c++) {
        double e = AT(i, c), e1 = AT(ip1, c);
        double s = -d12 * e1 + d22 * e, u = -d12 * e + d11 * e1;
        for (int k = c; k <= hi; k++) {
          double v = AT(c, k) + (-s) * AT(i, k);
          AT(c, k) = v + (-u) * AT(ip1, k);
        }
        if (last[c] < hi) last[c] = hi;
        AT(i, c) = s;
        AT(ip1, c) = u;
      }
      last[i] = hi, last[ip1] = hi;
    }
    i += 2;
  }
  return 0;
}

static int bkp_solve(const double *a, int n, const int *last, const int *pivot, double *x) {
  int i = 0;
  while (i < n - 1) {
    int ip1 = i + 1;
    double save = x[pivot[i]], aii = AT(i, i);
    if (pivot[ip1] > 0) { /* 1x1 */
      x[pivot[i]] = x[i];
      if (aii == 0.0) return KKTO_E_SING;
      x[i] = save / aii;
      for (int c = ip1; c <= last[i]; c++) x[c] -= save * AT(i, c);
      i = ip1;
    } else { /* 2x2 */
      double tmp = x[i], a12 = AT(i, ip1), a22 = AT(ip1, ip1);
      x[pivot[i]] = x[ip1];
      double det = aii * a22 - a12 * a12;
      if (det == 0.0) return KKTO_E_SING;
      x[i] = (tmp * a22 - save * a12) / det;
      x[ip1] = (save * aii - tmp * a12) / det;
      for (int c = i + 2; c <= last[i]; c++) x[c] -= tmp * AT(i, c);
      for (int c = i + 2; c <= last[ip1]; c++) x[c] -= save * AT(ip1, c);
      i += 2;
    }
  }
  if (i == n - 1) {
    if (AT(i, i) == 0.0) return KKTO_E_SING;
    x[i] /= AT(i, i);
    i = n - 2;
  } else
    i = n - 3;
  while (i >= 0) {
    int ii = (pivot[i] > 0 || i == 0) ? i : i - 1;
    for (int k = ii; k <= i; k++) {
      double t = x[k];
      for (int c = i + 1; c <= last[k]; c++) t -= AT(k, c) * x[c];
      x[k] = t;
    }
    if (i != pivot[ii]) {
      double t = x[i];
      x[i] = x[pivot[ii]];
      x[pivot[ii]] = t;
    }
    i = ii - 1;
  }
  return 0;
}
#undef AT

/* ---------------------------------------------------------------- factor */
int kkto_factor(kkto *h, const double *z, const double *w) {
  int n = h->n, me = h->me, m = h->m, dim = h->dim, nme = n + me;
  const int *P = h->qp2j, *IP = h->j2qp;
  memcpy(h->J, h->Jraw, sizeof(double) * (size_t)dim * dim);
  memcpy(h->last, h->last_raw, sizeof(int) * (size_t)dim);
  for (int j = 0; j < m; j++)
    if (z[j] == 0.0 || w[j] == 0.0) return KKTO_E_SING;
  if (h->mode == KKTO_FULL) {
    for (int j = 0; j < m; j++) {
      double wz = w[j] / z[j];
      int k = P[nme + j];
      h->J[(size_t)k * dim + k] = wz;
      h->scale[j] = dmin(1.0, sqrt(1.0 / wz));
    }
    for (int r = 0; r < dim; r++) {
      int qr = IP[r];
      double sr = qr >= nme ? h->scale[qr - nme] : 1.0;
      for (int c = r; c <= h->last[r]; c++) {
        double v = h->J[(size_t)r * dim + c] * sr;
        int qc = IP[c];
        if (qc >= nme) v *= h->scale[qc - nme];
        h->J[(size_t)r * dim + c] = v;
      }
    }
  } else {
    /* J11 -= C' diag(z/w) C (hqp/Hqp_IpRedSpBKP.C:104-181); for each pair of
     * x-columns the sum runs over the shared C rows in ascending order */
    for (int j = 0; j < m; j++) h->zw[j] = z[j] / w[j];
    double *S = dzero((long)n * n); /* lower incl. diagonal, in QP numbering */
    for (int r = 0; r < m; r++)
      for (int a = h->Cp[r]; a < h->Cp[r + 1]; a++)
        for (int b = h->Cp[r]; b < h->Cp[r + 1]; b++) {
          int ci = h->Ci[a], cj = h->Ci[b];
          if (cj <= ci) S[(size_t)ci * n + cj] += h->Cx[a] * h->zw[r] * h->Cx[b];
        }
    for (int i = 0; i < n; i++)
      for (int j = 0; j <= i; j++) {
        double s = S[(size_t)i * n + j];
        if (s == 0.0 && i != j) continue;
        int a = P[i], b = P[j], r = a < b ? a : b, c = a < b ? b : a;
        h->J[(size_t)r * dim + c] -= s;
        if (c > h->last[r]) h->last[r] = c;
        if (i == j) h->scale[i] = dmin(1.0, sqrt(-1.0 / h->J[(size_t)r * dim + c]));
      }
    free(S);
    for (int r = 0; r < dim; r++) {
      int qr = IP[r];
      double sr = qr < n ? h->scale[qr] : 1.0;
      for (int c = r; c <= h->last[r]; c++) {
        double v = h->J[(size_t)r * dim + c] * sr;
        int qc = IP[c];
        if (qc < n) v *= h->scale[qc];
        h->J[(size_t)r * dim + c] = v;
      }
    }
  }
  return bkp_factor(h->J, dim, h->last, h->pivot, h->tol);
}

/* ----------------------------------------------------------------- SpMVs */
static void sym_mv(int n, const int *p, const int *ix, const double *x, const double *v,
                   double *out) {
  /* upper-stored symmetric product, meschach/addon2_hqp.c:866-914 */
  for (int i = 0; i < n; i++) out[i] = 0.0;
  for (int i = 0; i < n; i++) {
    double sum = 0.0, vi = v[i];
    for (int k = p[i]; k < p[i + 1]; k++) {
      int c = ix[k];
      if (c == i)
        sum += x[k] * vi;
      else if (c > i)
        sum += x[k] * v[c], out[c] += x[k] * vi;
    }
    out[i] += sum;
  }
}
static void mv(int rows, const int *p, const int *ix, const double *x, const double *v,
               double *out) {
  for (int r = 0; r < rows; r++) {
    double s = 0.0;
    for (int k = p[r]; k < p[r + 1]; k++) s += x[k] * v[ix[k]];
    out[r] = s;
  }
}
static void vm_mltadd(int rows, const int *p, const int *ix, const double *x,
                      const double *v, double s, double *out) {
  /* out += s * M' v  (meschach/addon2_hqp.c:920-953) */
  for (int r = 0; r < rows; r++) {
    double t = s * v[r];
    for (int k = p[r]; k < p[r + 1]; k++) out[ix[k]] += t * x[k];
  }
}

/* ------------------------------------------------------------------ step */
int kkto_step(kkto *h, const double *z, const double *w, const double *r1, const double *r2,
              const double *r3, const double *r4, double *dx, double *dy, double *dz,
              double *dw) {
  int n = h->n, me = h->me, m = h->m, dim = h->dim, nme = n + me, e;
  const int *P = h->qp2j, *IP = h->j2qp;
  if (h->mode == KKTO_FULL) {
    for (int i = 0; i < n; i++) h->sol[i] = r1[i];
    for (int i = 0; i < me; i++) h->sol[n + i] = r2[i];
    for (int j = 0; j < m; j++) {
      if (z[j] == 0.0) return KKTO_E_SING;
      h->sol[nme + j] = (r4[j] / z[j] + r3[j]) * h->scale[j];
    }
    for (int k = 0; k < dim; k++) h->rhs[k] = h->sol[IP[k]];
    if ((e = bkp_solve(h->J, dim, h->last, h->pivot, h->rhs))) return e;
    for (int k = 0; k < dim; k++) h->sol[k] = h->rhs[P[k]];
    for (int i = 0; i < n; i++) dx[i] = h->sol[i];
    for (int i = 0; i < me; i++) dy[i] = h->sol[n + i];
    for (int j = 0; j < m; j++) dz[j] = h->sol[nme + j] * h->scale[j];
  } else {
    for (int j = 0; j < m; j++) {
      if (w[j] == 0.0) return KKTO_E_SING;
      dw[j] = r4[j] / w[j];
      dz[j] = h->zw[j] * r3[j];
      dz[j] = dw[j] + dz[j];
    }
    /* rhs1 = (r1 - C' dz) .* scale */
    for (int i = 0; i < n; i++) h->sol[i] = 0.0;
    vm_mltadd(m, h->Cp, h->Ci, h->Cx, dz, 1.0, h->sol);
    for (int i = 0; i < n; i++) h->sol[i] = (r1[i] - h->sol[i]) * h->scale[i];
    for (int i = 0; i < me; i++) h->sol[n + i] = r2[i];
    for (int k = 0; k < dim; k++) h->rhs[k] = h->sol[IP[k]];
    if ((e = bkp_solve(h->J, dim, h->last, h->pivot, h->rhs))) return e;
    for (int k = 0; k < dim; k++) h->sol[k] = h->rhs[P[k]];
    for (int i = 0; i < n; i++) dx[i] = h->sol[i] * h->scale[i];
    for (int i = 0; i < me; i++) dy[i] = h->sol[n + i];
    mv(m, h->Cp, h->Ci, h->Cx, dx, dw);
    for (int j = 0; j < m; j++) dz[j] = dz[j] - h->zw[j] * dw[j];
  }
  /* dw = C dx - r3 */
  for (int j = 0; j < m; j++) dw[j] = -1.0 * r3[j];
  for (int r = 0; r < m; r++) {
    double s = 0.0;
    for (int k = h->Cp[r]; k < h->Cp[r + 1]; k++) s += h->Cx[k] * dx[h->Ci[k]];
    dw[r] += 1.0 * s;
  }
  return 0;
}

/* -------------------------------------------------------------- residuum */
static double norm_inf(const double *v, int k) {
  double r = 0.0;
  for (int i = 0; i < k; i++)
    if (fabs(v[i]) > r) r = fabs(v[i]);
  return r;
}

double kkto_residuum(kkto *h, const double *z, const double *w, const double *r1,
                     const double *r2, const double *r3, const double *r4, const double *dx,
                     const double *dy, const double *dz, const double *dw) {
  int n = h->n, me = h->me, m = h->m;
  double *t1 = h->t1, *t2 = h->t2, *t3 = h->t3, *t4 = h->t4;
  sym_mv(n, h->Qp, h->Qi, h->Qx, dx, t1);
  vm_mltadd(me, h->Ap, h->Ai, h->Ax, dy, -1.0, t1);
  vm_mltadd(m, h->Cp, h->Ci, h->Cx, dz, -1.0, t1);
  for (int i = 0; i < n; i++) t1[i] = r1[i] + t1[i];
  mv(me, h->Ap, h->Ai, h->Ax, dx, t2);
  for (int i = 0; i < me; i++) t2[i] = r2[i] - t2[i];
  for (int j = 0; j < m; j++) t4[j] = r4[j] - (z[j] * dw[j] + w[j] * dz[j]);
  mv(m, h->Cp, h->Ci, h->Cx, dx, t3);
  for (int j = 0; j < m; j++) t3[j] = r3[j] - (t3[j] - dw[j]);
  double res = norm_inf(t1, n), p;
  if ((p = norm_inf(t2, me)) > res) res = p;
  if ((p = norm_inf(t3, m)) > res) res = p;
  if ((p = norm_inf(t4, m)) > res) res = p;
  return res;
}

/* ----------------------------------------------------------------- solve */
static void axpy4(kkto *h, double a, double *dx, double *dy, double *dz, double *dw) {
  for (int i = 0; i < h->n; i++) dx[i] = dx[i] + a * h->e1[i];
  for (int i = 0; i < h->me; i++) dy[i] = dy[i] + a * h->e2[i];
  for (int i = 0; i < h->m; i++) dz[i] = dz[i] + a * h->e3[i];
  for (int i = 0; i < h->m; i++) dw[i] = dw[i] + a * h->e4[i];
}

int kkto_solve(kkto *h, const double *z, const double *w, const double *r1, const double *r2,
               const double *r3, const double *r4, double *dx, double *dy, double *dz,
               double *dw, double *res_out, int *rounds_out) {
  int e, rounds = 0;
  if ((e = kkto_step(h, z, w, r1, r2, r3, r4, dx, dy, dz, dw))) return e;
  double res = kkto_residuum(h, z, w, r1, r2, r3, r4, dx, dy, dz, dw), last;
  for (int it = 0; it < 5 && res > h->eps; it++) {
    last = res;
    /* kkto_residuum left the residual vectors in t1..t4 (the reference's _r1.._r4) */
    if ((e = kkto_step(h, z, w, h->t1, h->t2, h->t3, h->t4, h->e1, h->e2, h->e3, h->e4)))
      return e;
    rounds++;
    double alpha = 1.0;
    do {
      axpy4(h, alpha, dx, dy, dz, dw);
      res = kkto_residuum(h, z, w, r1, r2, r3, r4, dx, dy, dz, dw);
      if (res > last) {
        axpy4(h, -alpha, dx, dy, dz, dw);
        alpha -= 0.3;
      }
    } while (res > last && alpha > 0.0);
    if (alpha <= 0.0) break;
  }
  *res_out = res;
  if (rounds_out) *rounds_out = rounds;
  return 0;
}

/* --------------------------------------------------------------- getters */
int kkto_sbw(const kkto *h) { return h->sbw; }
int kkto_dim(const kkto *h) { return h->dim; }
void kkto_get_perm(const kkto *h, int *out) { memcpy(out, h->qp2j, sizeof(int) * (size_t)h->dim); }
void kkto_get_pivot(const kkto *h, int *out) { memcpy(out, h->pivot, sizeof(int) * (size_t)h->dim); }
/* which: 0 = assembled raw matrix, 1 = factored; dense row-major dim x dim */
void kkto_get_dense(const kkto *h, int which, double *out) {
  memcpy(out, which ? h->J : h->Jraw, sizeof(double) * (size_t)h->dim * h->dim);
}
