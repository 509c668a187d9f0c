// Symbolic phase (host).  See analysis.hpp.
//
// Reference semantics followed here:
//   KKT pattern  hqp/Hqp_IpSpBKP.C:117-136  (-Q upper, A, C, slack diagonal)
//                hqp/Hqp_IpRedSpBKP.C:203-262 (-(Q + C'C) pattern, A)
//   RCM          hqp/sprcm.C:62-211 (graph), :226-384 (order), :391-420 (sbw)
// Everything after the RCM order (nested dissection of the band, supernodes,
// fronts) has no counterpart in the reference, whose factorisation is a
// sequential row-list elimination (hqp/spBKP.C:406-636).
#include "analysis.hpp"

#include <algorithm>
#include <functional>
#include <cstring>
#include <numeric>
#include <thread>

#include <chrono>
#include <cstdio>
#include <cstdlib>
static std::chrono::steady_clock::time_point g_t0;
#define TMARK(x) do { if (getenv("HQPKKT_TIMING")) { auto t=std::chrono::steady_clock::now(); fprintf(stderr, "phase before %s: %.2f s\n", x, std::chrono::duration<double>(t-g_t0).count()); g_t0=t; } } while(0)
namespace kktdev {
namespace {

// fn(lo, hi) over [0, n) on up to 16 host threads (the symbolic phase is otherwise single-threaded, as the
// reference's init() is; only loops without dependences between their iterations use this)
template <class F>
void parallel_for(long long n, F fn) {
  unsigned nt = std::thread::hardware_concurrency();
  nt = nt < 1 ? 1 : (nt > 16 ? 16 : nt);
  if (n < 200000 || nt == 1) {
    fn(0, n);
    return;
  }
  std::vector<std::thread> th;
  const long long chunk = (n + nt - 1) / nt;
  for (unsigned t = 0; t < nt; t++) {
    const long long lo = t * chunk, hi = std::min(n, lo + chunk);
    if (lo < hi) th.emplace_back([=, &fn]() { fn(lo, hi); });
  }
  for (auto &x : th) x.join();
}

struct LevelItem {
  int node, deg, deg2;
};

// stable LSD radix sort of (key, payload) pairs by key, 16 bits per pass: the entry
// lists of dense stage blocks hold 10^7..10^8 items
static void radix_sort_pairs(std::vector<std::pair<long long, int>> &v) {
  if (v.size() < 2) return;
  long long mx = 0;
  for (auto &e : v) mx = std::max(mx, e.first);
  std::vector<std::pair<long long, int>> tmp(v.size());
  for (int shift = 0; shift < 64 && (mx >> shift) > 0; shift += 16) {
    std::vector<size_t> cnt(65537, 0);
    for (auto &e : v) cnt[((e.first >> shift) & 0xffff) + 1]++;
    for (int d = 0; d < 65536; d++) cnt[d + 1] += cnt[d];
    for (auto &e : v) tmp[cnt[(e.first >> shift) & 0xffff]++] = e;
    v.swap(tmp);
  }
}

// Reverse Cuthill-McKee exactly as the reference runs it: the start node of
// each component is the first unnumbered node; the level structure is rebuilt
// from the minimum-degree node of the last level while the number of levels
// grows; inside a level, nodes are kept sorted by (live degree, sum of live
// degrees of unnumbered neighbours) with a stable sort, the whole new level
// being re-sorted after each parent has been expanded.
void rcm_order(int dim, const std::vector<int> &start, const std::vector<int> &neigh,
               std::vector<int> &order) {
  std::vector<char> marks(dim), glob(dim, 1);
  std::vector<LevelItem> lv(dim + 1);
  // live[v]: neighbours of v that are not numbered yet; lum[v] = marks[v] ? live[v] : 0, so that the
  // second sort key of a node (sum of the live degrees of its unnumbered neighbours) is ONE gather
  std::vector<int> live(dim), lum(dim);
  auto reset_live = [&]() {
    for (int i = 0; i < dim; i++) live[i] = start[i + 1] - start[i];
  };
  auto number = [&](int v, int &count) {
    lv[count++].node = v;
    marks[v] = 0;
    lum[v] = 0;
    for (int k = start[v]; k < start[v + 1]; k++) {
      const int w = neigh[k];
      live[w]--;
      if (marks[w]) lum[w]--;
    }
  };
  reset_live();
  int root = 0, count = 0;
  while (root < dim) {
    int nlev = 0, nlev_old, first = count, lb = 0, le = 0;
    do {
      count = first;
      marks = glob;
      for (int i = 0; i < dim; i++) lum[i] = marks[i] ? live[i] : 0;
      nlev_old = nlev;
      nlev = 0;
      lb = le = count;
      number(root, count);
      do {
        lb = le;
        le = count;
        nlev++;
        for (int i = lb; i < le; i++) {
          const int v = lv[i].node;
          for (int j = start[v]; j < start[v + 1]; j++)
            if (marks[neigh[j]]) number(neigh[j], count);
          if (count - le > 1) {
            for (int k = le; k < count; k++) {
              const int u = lv[k].node;
              int d2 = 0;
              for (int j = start[u]; j < start[u + 1]; j++) d2 += lum[neigh[j]];
              lv[k].deg = live[u];
              lv[k].deg2 = d2;
            }
            std::stable_sort(lv.begin() + le, lv.begin() + count,
                             [](const LevelItem &a, const LevelItem &b) {
                               return a.deg != b.deg ? a.deg < b.deg : a.deg2 < b.deg2;
                             });
          }
        }
      } while (count > le);
      root = lv[lb].node;
      for (int i = lb + 1; i < le; i++)
        if (live[lv[i].node] < live[root]) root = lv[i].node;
      reset_live();
    } while (nlev > nlev_old);
    glob = marks;
    root = dim;
    for (int i = 0; i < dim; i++)
      if (marks[i]) root = i;
  }
  order.assign(dim, 0);
  for (int i = 0; i < dim; i++) order[lv[i].node] = dim - 1 - i;
}

// Plain reverse Cuthill-McKee (one breadth-first pass per component from its first node, the new
// neighbours of a node in the order of their degrees): ordering 2 does not report the reference's
// permutation, and the reference-faithful pass above re-sorts a whole level after every parent, which is
// quadratic in the level width (seconds for a 10^6-node mesh)
void cm_order_plain(int dim, const std::vector<int> &start, const std::vector<int> &neigh, std::vector<int> &order) {
  std::vector<int> seq;
  seq.reserve(dim);
  std::vector<char> seen(dim, 0);
  std::vector<int> nb;
  for (int root = 0; root < dim; root++) {
    if (seen[root]) continue;
    seen[root] = 1;
    size_t head = seq.size();
    seq.push_back(root);
    while (head < seq.size()) {
      const int v = seq[head++];
      nb.clear();
      for (int k = start[v]; k < start[v + 1]; k++)
        if (!seen[neigh[k]]) seen[neigh[k]] = 1, nb.push_back(neigh[k]);
      std::stable_sort(nb.begin(), nb.end(), [&](int a, int b) { return start[a + 1] - start[a] < start[b + 1] - start[b]; });
      seq.insert(seq.end(), nb.begin(), nb.end());
    }
  }
  order.assign(dim, 0);
  for (int i = 0; i < dim; i++) order[seq[i]] = dim - 1 - i;
}

bool csr_ok(int rows, int cols, const int *p, const int *ix) {
  if (!p) return rows == 0;
  if (p[0] != 0) return false;
  for (int r = 0; r < rows; r++) {
    if (p[r + 1] < p[r]) return false;
    for (int k = p[r]; k < p[r + 1]; k++) {
      if (ix[k] < 0 || ix[k] >= cols) return false;
      if (k > p[r] && ix[k] <= ix[k - 1]) return false;
    }
  }
  return true;
}

struct RawEntry {
  long long key;
  int a, b;
  Term t;
};

void transpose(const Analysis::Csr &M, int cols, Analysis::Csr &T) {
  T.rows = cols;
  T.ptr.assign(cols + 1, 0);
  for (int c : M.col) T.ptr[c + 1]++;
  for (int c = 0; c < cols; c++) T.ptr[c + 1] += T.ptr[c];
  T.col.resize(M.col.size());
  T.src.resize(M.col.size());
  std::vector<int> fill(T.ptr.begin(), T.ptr.end() - 1);
  for (int r = 0; r < M.rows; r++)
    for (int k = M.ptr[r]; k < M.ptr[r + 1]; k++) {
      int d = fill[M.col[k]]++;
      T.col[d] = r;
      T.src[d] = M.src[k];
    }
}

}  // namespace

int Analysis::setup_blocks(int mode_, int n_, int me_, int m_, const int *Qp, const int *Qi,
                           const int *Ap, const int *Ai, const int *Cp, const int *Ci) {
  mode = mode_, n = n_, me = me_, m = m_;
  if (n < 0 || me < 0 || m < 0 || n + me + m == 0) return 1;
  if (!csr_ok(n, n, Qp, Qi) || !csr_ok(me, n, Ap, Ai) || !csr_ok(m, n, Cp, Ci)) return 6;
  dim = mode == 0 ? n + me + m : n + me;
  nq = n ? Qp[n] : 0, na = me ? Ap[me] : 0, nc = m ? Cp[m] : 0;
  Qfull = Csr(), A = Csr(), AT = Csr(), C = Csr(), CT = Csr();
  // ---------------------------------------------------------- SpMV blocks
  {
    std::vector<std::vector<std::pair<int, int>>> rows(n);
    for (int i = 0; i < n; i++)
      for (int k = Qp[i]; k < Qp[i + 1]; k++)
        if (Qi[k] >= i) {
          rows[i].push_back({Qi[k], k});
          if (Qi[k] != i) rows[Qi[k]].push_back({i, k});
        }
    Qfull.rows = n;
    Qfull.ptr.assign(n + 1, 0);
    for (int i = 0; i < n; i++) {
      std::sort(rows[i].begin(), rows[i].end());
      Qfull.ptr[i + 1] = Qfull.ptr[i] + (int)rows[i].size();
      for (auto &e : rows[i]) Qfull.col.push_back(e.first), Qfull.src.push_back(e.second);
    }
    A.rows = me;
    A.ptr.assign(Ap ? Ap : (const int *)nullptr, Ap ? Ap + me + 1 : nullptr);
    if (A.ptr.empty()) A.ptr.assign(1, 0);
    A.col.assign(Ai, Ai + na);
    A.src.resize(na);
    std::iota(A.src.begin(), A.src.end(), nq);
    C.rows = m;
    C.ptr.assign(Cp ? Cp : (const int *)nullptr, Cp ? Cp + m + 1 : nullptr);
    if (C.ptr.empty()) C.ptr.assign(1, 0);
    C.col.assign(Ci, Ci + nc);
    C.src.resize(nc);
    std::iota(C.src.begin(), C.src.end(), nq + na);
    transpose(A, n, AT);
    transpose(C, n, CT);
  }
  return 0;
}

int Analysis::run(int mode_, int n_, int me_, int m_, const int *Qp, const int *Qi,
                  const int *Ap, const int *Ai, const int *Cp, const int *Ci, int leaf_size,
                  int max_pivots, int zd_policy) {
  TMARK("0");
  if (int e0 = setup_blocks(mode_, n_, me_, m_, Qp, Qi, Ap, Ai, Cp, Ci)) return e0;
  const int ONE = nq + na + nc, WONE = m;
  if (max_pivots <= 0) max_pivots = 128;
  if (max_pivots > 192) max_pivots = 192;  // k_factor_blk: 12 blocks of 16 (k_factor_diag of rounds 1-3: 128)
  if (leaf_size <= 0) leaf_size = 0;  // decided below from sbw

  TMARK("1");
  // ------------------------------------------------------------- entries
  std::vector<RawEntry> raw;
  if (mode == 0) {
    // FULL: every entry is a single term and the sorted order (row-major over the upper
    // triangle: x row i holds its Q entries, then column i of A, then column i of C; the
    // slack rows hold their diagonal) can be written down directly - no 48-byte records,
    // no sort (dense stage blocks: 10^7..10^8 entries)
    const size_t tot = (size_t)na + nc + m + (size_t)nq;
    ent_a.clear(), ent_b.clear(), term_ptr.assign(1, 0), terms.clear();
    ent_a.reserve(tot), ent_b.reserve(tot), terms.reserve(tot), term_ptr.reserve(tot + 1);
    auto put = [&](int a, int b, Term t) {
      ent_a.push_back(a), ent_b.push_back(b), terms.push_back(t), term_ptr.push_back((int)terms.size());
    };
    for (int i = 0; i < n; i++) {
      for (int k = Qp[i]; k < Qp[i + 1]; k++)
        if (Qi[k] >= i) put(i, Qi[k], {k, ONE, WONE, -1.0});
      for (int t = AT.ptr[i]; t < AT.ptr[i + 1]; t++) put(i, n + AT.col[t], {AT.src[t], ONE, WONE, 1.0});
      for (int t = CT.ptr[i]; t < CT.ptr[i + 1]; t++) put(i, n + me + CT.col[t], {CT.src[t], ONE, WONE, 1.0});
    }
    for (int j = 0; j < m; j++) put(n + me + j, n + me + j, {ONE, ONE, j, 1.0});
  } else {
  raw.reserve((size_t)nq + na + 4 * (size_t)nc);
  auto add = [&](int a, int b, Term t) {
    int lo = std::min(a, b), hi = std::max(a, b);
    raw.push_back({(long long)lo * dim + hi, lo, hi, t});
  };
  for (int i = 0; i < n; i++)
    for (int k = Qp[i]; k < Qp[i + 1]; k++)
      if (Qi[k] >= i) add(i, Qi[k], {k, ONE, WONE, -1.0});
  for (int r = 0; r < me; r++)
    for (int k = Ap[r]; k < Ap[r + 1]; k++) add(n + r, Ai[k], {nq + k, ONE, WONE, 1.0});
  for (int r = 0; r < m; r++)
    for (int a = Cp[r]; a < Cp[r + 1]; a++)
      for (int b = Cp[r]; b <= a; b++)
        add(Ci[a], Ci[b], {nq + na + a, nq + na + b, r, -1.0});
  // entries with the same (row, col) are merged into one entry with several terms; the
  // order is (key, order of generation).  Sorting (key, index) pairs instead of the
  // 48-byte records keeps this phase short for the 10^7..10^8 entries of dense stage blocks.
  {
    std::vector<std::pair<long long, int>> order(raw.size());
    for (size_t k = 0; k < raw.size(); k++) order[k] = {raw[k].key, (int)k};
    radix_sort_pairs(order);
    ent_a.clear(), ent_b.clear(), term_ptr.assign(1, 0), terms.clear();
    ent_a.reserve(raw.size()), ent_b.reserve(raw.size()), terms.reserve(raw.size());
    for (size_t k = 0; k < order.size(); k++) {
      const RawEntry &r = raw[order[k].second];
      if (k == 0 || order[k].first != order[k - 1].first) {
        if (k) term_ptr.push_back((int)terms.size());
        ent_a.push_back(r.a), ent_b.push_back(r.b);
      }
      terms.push_back(r.t);
    }
    term_ptr.push_back((int)terms.size());
  }
  }
  { std::vector<RawEntry>().swap(raw); }
  const int nent = (int)ent_a.size();
  diag_ent.assign(n, -1);
  for (int e = 0; e < nent; e++)
    if (ent_a[e] == ent_b[e] && ent_a[e] < n) diag_ent[ent_a[e]] = e;

  TMARK("2");
  // ---------------------------------------------------------- RCM graph
  // neighbour lists in the reference's visiting order: x-x couplings row by
  // row (upper triangle), then the rows of A, then the rows of C
  std::vector<int> gdeg(dim + 1, 0), gstart(dim + 1, 0);
  auto count_edge = [&](int a, int b) { gdeg[a]++, gdeg[b]++; };
  for (int e = 0; e < nent; e++)
    if (ent_b[e] < n && ent_a[e] != ent_b[e]) count_edge(ent_a[e], ent_b[e]);
  for (int r = 0; r < me; r++)
    for (int k = Ap[r]; k < Ap[r + 1]; k++) count_edge(n + r, Ai[k]);
  if (mode == 0)
    for (int r = 0; r < m; r++)
      for (int k = Cp[r]; k < Cp[r + 1]; k++) count_edge(n + me + r, Ci[k]);
  for (int i = 0; i < dim; i++) gstart[i + 1] = gstart[i] + gdeg[i];
  std::vector<int> gneigh(gstart[dim]), gfill(gstart.begin(), gstart.end() - 1);
  auto link = [&](int a, int b) { gneigh[gfill[a]++] = b, gneigh[gfill[b]++] = a; };
  for (int e = 0; e < nent; e++)
    if (ent_b[e] < n && ent_a[e] != ent_b[e]) link(ent_a[e], ent_b[e]);
  for (int r = 0; r < me; r++)
    for (int k = Ap[r]; k < Ap[r + 1]; k++) link(n + r, Ai[k]);
  if (mode == 0)
    for (int r = 0; r < m; r++)
      for (int k = Cp[r]; k < Cp[r + 1]; k++) link(n + me + r, Ci[k]);

  if (ordering == 2)
    cm_order_plain(dim, gstart, gneigh, qp2j);
  else
    rcm_order(dim, gstart, gneigh, qp2j);
  sbw = 0;
  std::vector<int> reach(dim);  // by band position: farthest coupled position
  for (int v = 0; v < dim; v++) {
    int far = qp2j[v];
    for (int k = gstart[v]; k < gstart[v + 1]; k++) far = std::max(far, qp2j[gneigh[k]]);
    reach[qp2j[v]] = far;
    sbw = std::max(sbw, far - qp2j[v]);
  }

  TMARK("3");
  // ------------------------------------------- nested dissection of the band
  // Logical nodes: leaves = the still unassigned positions of an interval of the
  // band order, inner nodes = vertex separators.  Cutting the interval at `mid`,
  // the separator is the set of positions >= mid that are coupled to a position
  // < mid (all of them lie within sbw of the cut); positions of the window that
  // are not coupled across the cut (e.g. slack rows whose x sits on the right)
  // stay in the right part, so a separator is usually well below sbw rows.  Each
  // logical node becomes a chain of supernodes of <= max_pivots pivots below.
  if (leaf_size <= 0) leaf_size = ordering >= 1 ? 32 : std::max(3 * std::max(sbw, 1) / 2, 32);
  struct Tmp {
    std::vector<int> verts;  // band positions
    std::vector<int> kids;
  };
  std::vector<Tmp> tmp;
  std::vector<int> pos2q_nd(dim);
  for (int q = 0; q < dim; q++) pos2q_nd[qp2j[q]] = q;
  std::vector<char> taken(dim, 0);  // position already belongs to a separator
  auto make = [&](std::vector<int> verts, std::vector<int> kids) {
    tmp.push_back(Tmp{std::move(verts), std::move(kids)});
    return (int)tmp.size() - 1;
  };
  auto free_positions = [&](int lo, int hi) {
    std::vector<int> v;
    for (int r = lo; r < hi; r++)
      if (!taken[r]) v.push_back(r);
    return v;
  };
  std::vector<int> roots;
  if (ordering >= 1) {
    // ---- general graphs: nested dissection by level structures (George's automatic nested
    // dissection).  Per connected piece: breadth-first levels from a pseudo-peripheral vertex, the
    // thinnest level of the middle half is the separator (only its vertices that touch the next
    // level: the others join the near side), both sides recursively.  A mesh-like sparsity
    // (discretised problems, the CUTE collection's typical structure) gets separators that shrink
    // with the piece; the cut of the RCM band above keeps them at the band width on every level.
    const int leaf = std::max(leaf_size, 8);
    std::vector<int> in_set(dim, -1), lvl(dim, -1), queue;
    int set_id = 0;
    // breadth-first levels of the piece that holds `start` inside set `sid`; returns the level count
    auto bfs = [&](int start, int sid, std::vector<int> &order, std::vector<int> &lptr) {
      order.clear(), lptr.clear();
      order.push_back(start), lvl[start] = 0;
      lptr.push_back(0);
      size_t head = 0;
      int cur = 0;
      while (head < order.size()) {
        const int q = order[head];
        if (lvl[q] != cur) cur = lvl[q], lptr.push_back((int)head);
        head++;
        for (int k = gstart[q]; k < gstart[q + 1]; k++) {
          const int x = gneigh[k];
          if (in_set[x] == sid && lvl[x] < 0) lvl[x] = cur + 1, order.push_back(x);
        }
      }
      lptr.push_back((int)order.size());
      return (int)lptr.size() - 1;
    };
    // the caller's numbering as a linear order of ALL vertices: x_i at i; a multiplier or slack at the mean index of the
    // x variables it is coupled to (scaled by 2, so that it falls between them)
    std::vector<long long> natkey(dim);
    for (int q = 0; q < dim; q++) {
      if (q < n) {
        natkey[q] = 2LL * q;
        continue;
      }
      long long sum = 0, cnt = 0;
      for (int k = gstart[q]; k < gstart[q + 1]; k++)
        if (gneigh[k] < n) sum += gneigh[k], cnt++;
      natkey[q] = cnt ? 2 * sum / cnt + 1 : 2LL * q;
    }
    std::function<std::vector<int>(std::vector<int> &)> dissect = [&](std::vector<int> &S) -> std::vector<int> {
      std::vector<int> out;
      const int sid = set_id++;
      for (int q : S) in_set[q] = sid, lvl[q] = -1;
      std::vector<int> order, lptr, best_order, best_lptr;
      for (int q0 : S) {
        if (lvl[q0] >= 0) continue;  // piece already handled
        // pseudo-peripheral start: repeat from a minimum-degree vertex of the last level while the depth grows
        int start = q0, depth = bfs(start, sid, order, lptr);
        for (int round = 0; round < 4; round++) {
          int cand = -1, cdeg = 0x7fffffff;
          for (int t = lptr[depth - 1]; t < lptr[depth]; t++) {
            const int q = order[t], dg = gstart[q + 1] - gstart[q];
            if (dg < cdeg) cdeg = dg, cand = q;
          }
          best_order = order, best_lptr = lptr;
          for (int q : order) lvl[q] = -1;
          const int d2 = bfs(cand, sid, order, lptr);
          if (d2 <= depth) {  // no deeper: keep the previous structure
            for (int q : order) lvl[q] = -1;
            order = best_order, lptr = best_lptr;
            for (int l = 0; l < depth; l++)
              for (int t = lptr[l]; t < lptr[l + 1]; t++) lvl[order[t]] = l;
            break;
          }
          depth = d2, start = cand;
        }
        const int len = (int)order.size();
        auto as_leaf = [&]() {
          std::vector<int> v;
          for (int q : order) v.push_back(qp2j[q]);
          std::sort(v.begin(), v.end());
          out.push_back(make(std::move(v), {}));
        };
        if (len <= leaf || depth < 3) {
          as_leaf();
          continue;
        }
        // thinnest level with at least a quarter of the piece on either side; none: the one next to the median
        int cut = -1;
        long long bestw = 0;
        for (int l = 1; l + 1 < depth; l++) {
          const int before = lptr[l], after = len - lptr[l + 1], w = lptr[l + 1] - lptr[l];
          if (4LL * before >= len && 4LL * after >= len && (cut < 0 || w < bestw)) cut = l, bestw = w;
        }
        if (cut < 0) {
          long long bestd = 0;
          for (int l = 1; l + 1 < depth; l++) {
            const long long dd = std::llabs(2LL * lptr[l] + (lptr[l + 1] - lptr[l]) - len);
            if (cut < 0 || dd < bestd) cut = l, bestd = dd;
          }
        }
        std::vector<int> sep, left, right;
        for (int t = lptr[cut]; t < lptr[cut + 1]; t++) {
          const int q = order[t];
          bool touches = false;
          for (int k = gstart[q]; k < gstart[q + 1] && !touches; k++)
            touches = in_set[gneigh[k]] == sid && lvl[gneigh[k]] == cut + 1;
          (touches ? sep : left).push_back(q);
        }
        left.insert(left.end(), order.begin(), order.begin() + lptr[cut]);
        right.assign(order.begin() + lptr[cut + 1], order.end());
        {
          // Further candidates: the piece cut in two halves by a linear ORDER of its vertices, the separator a vertex
          // cover of the edges that cross (greedy: the vertex with the most crossing edges first).  A level structure is
          // the wrong tool where a few far couplings tie distant parts of a band or mesh together - every level then
          // reaches across the whole piece (a band of 10^5 variables with 1000 far couplings: separators of thousands, a
          // root front of 11 724 rows) - while a cut of the right order pays one end of each crossing coupling and the
          // band width.  Two orders are tried: the Cuthill-McKee numbering (itself a breadth-first order: good for
          // meshes, scrambled by far couplings) and the caller's own numbering of the variables (multipliers and slacks
          // at the mean index of their x: programs from discretisations and multistage problems come banded in it).
          // The smallest separator of the three wins.
          // (only the two halves are needed, not the order inside them: nth_element, linear in the piece - two full sorts
          // per piece doubled the time of the symbolic phase of a 10^6-cell mesh.  Not tried where the level structure's
          // separator is as small as a planar mesh's: below the square root of the piece's size.)
          static const bool levels_only = getenv("HQPKKT_ND_LEVELS_ONLY") != nullptr;  // (round 4's separators: comparisons)
          const bool suspicious = !levels_only && (double)sep.size() * sep.size() > 1.0 * len;
          std::vector<int> saved(suspicious ? len : 0), byp, best_sep, best_left, best_right;
          if (suspicious) byp = order;
          for (int t = 0; t < len && suspicious; t++) saved[t] = lvl[order[t]];
          std::vector<int> &side = lvl;  // scratch marks for this piece: -3 left, -4 right, -5 in the cover (in_set stays sid)
          for (int pass = 0; pass < 2 && suspicious; pass++) {
            const int half = len / 2;
            if (pass == 0)
              std::nth_element(byp.begin(), byp.begin() + half, byp.end(), [&](int a, int b) { return qp2j[a] < qp2j[b]; });
            else
              std::nth_element(byp.begin(), byp.begin() + half, byp.end(),
                               [&](int a, int b) { return natkey[a] != natkey[b] ? natkey[a] < natkey[b] : a < b; });
            for (int t = 0; t < len; t++) side[byp[t]] = t < half ? -3 : -4;
            std::vector<std::pair<int, int>> cand;  // (crossing degree, vertex)
            for (int q : byp) {
              int c = 0;
              for (int k = gstart[q]; k < gstart[q + 1]; k++) {
                const int x = gneigh[k];
                c += in_set[x] == sid && side[x] != side[q];
              }
              if (c) cand.push_back({c, q});
            }
            std::sort(cand.begin(), cand.end(), [&](const std::pair<int, int> &a, const std::pair<int, int> &b) {
              return a.first != b.first ? a.first > b.first : a.second < b.second;
            });
            std::vector<int> sep2;
            for (auto &cq : cand) {  // taken if one of its crossing edges is still uncovered
              const int q = cq.second;
              bool open = false;
              for (int k = gstart[q]; k < gstart[q + 1] && !open; k++) {
                const int x = gneigh[k];
                open = in_set[x] == sid && (side[x] == -3 || side[x] == -4) && side[x] != side[q];
              }
              if (open) side[q] = -5, sep2.push_back(q);
            }
            if (sep2.size() < (best_sep.empty() ? sep.size() : best_sep.size())) {
              best_sep = sep2;
              best_left.clear(), best_right.clear();
              for (int t = 0; t < len; t++) {
                const int q = byp[t];
                if (side[q] == -3) best_left.push_back(q);
                if (side[q] == -4) best_right.push_back(q);
              }
            }
          }
          if (!best_sep.empty() && best_sep.size() < sep.size()) sep.swap(best_sep), left.swap(best_left), right.swap(best_right);
          for (int t = 0; t < len && suspicious; t++) lvl[order[t]] = saved[t];
        }
        if (3LL * (long long)sep.size() >= len) {
          as_leaf();
          continue;
        }
        std::vector<int> sv;
        for (int q : sep) sv.push_back(qp2j[q]);
        std::sort(sv.begin(), sv.end());
        // (the recursion re-labels in_set / lvl of its vertices: nothing of this piece is read afterwards)
        std::vector<int> kids = dissect(left);
        std::vector<int> kr = dissect(right);
        kids.insert(kids.end(), kr.begin(), kr.end());
        // mark the piece as handled for the loop over S (lvl >= 0) under this frame's set id again
        for (int q : order) in_set[q] = -2, lvl[q] = 0;
        out.push_back(make(std::move(sv), std::move(kids)));
      }
      return out;
    };
    std::vector<int> all(dim);
    for (int q = 0; q < dim; q++) all[q] = q;
    roots = dissect(all);
  } else {
  struct Frame {
    int lo, hi, stage, mid;
    std::vector<int> sep, left, right;
  };
  {
    std::vector<Frame> st;
    std::vector<std::vector<int>> ret;  // return values stack
    st.push_back({0, dim, 0, 0, {}, {}, {}});
    while (!st.empty()) {
      Frame &f = st.back();
      if (f.stage == 0) {
        std::vector<int> fr = free_positions(f.lo, f.hi);
        const int len = (int)fr.size();
        if (len == 0) {
          ret.push_back({});
          st.pop_back();
          continue;
        }
        if (len <= leaf_size) {
          ret.push_back({make(fr, {})});
          st.pop_back();
          continue;
        }
        // cut so that both sides keep about (len - separator)/2 free rows
        f.mid = fr[std::max(1, (len - std::min(sbw, len / 3)) / 2)];
        std::vector<int> sep;
        for (int r = f.mid; r < f.hi && r <= f.mid + sbw; r++) {
          if (taken[r]) continue;
          const int q = pos2q_nd[r];
          bool coupled = false;
          for (int k = gstart[q]; k < gstart[q + 1] && !coupled; k++) {
            const int u = qp2j[gneigh[k]];
            coupled = u >= f.lo && u < f.mid && !taken[u];
          }
          if (coupled) sep.push_back(r);
        }
        if ((int)sep.size() * 3 >= len) {  // separator would dominate: keep as one node
          ret.push_back({make(fr, {})});
          st.pop_back();
          continue;
        }
        for (int r : sep) taken[r] = 1;
        f.sep = sep;
        f.stage = 1;
        const int lo = f.lo, mid = f.mid;
        st.push_back({lo, mid, 0, 0, {}, {}, {}});
        continue;
      }
      if (f.stage == 1) {
        f.left = ret.back();
        ret.pop_back();
        f.stage = 2;
        const int mid = f.mid, hi = f.hi;
        st.push_back({mid, hi, 0, 0, {}, {}, {}});
        continue;
      }
      f.right = ret.back();
      ret.pop_back();
      std::vector<int> kids = f.left;
      kids.insert(kids.end(), f.right.begin(), f.right.end());
      if (!f.sep.empty())
        ret.push_back({make(f.sep, kids)});
      else
        ret.push_back(kids);
      st.pop_back();
    }
    roots = ret.back();
  }

  }

  // postorder of the logical nodes
  const int nlog = (int)tmp.size();
  std::vector<int> lid(nlog, -1), lorder;
  lorder.reserve(nlog);
  {
    std::vector<std::pair<int, size_t>> st;
    for (int r : roots) {
      st.push_back({r, 0});
      while (!st.empty()) {
        auto &top = st.back();
        if (top.second < tmp[top.first].kids.size()) {
          int c = tmp[top.first].kids[top.second++];
          st.push_back({c, 0});
        } else {
          lid[top.first] = (int)lorder.size();
          lorder.push_back(top.first);
          st.pop_back();
        }
      }
    }
  }
  std::vector<std::vector<int>> lverts(nlog);  // pivot sets by band position
  std::vector<int> lnode_of_pos(dim);
  for (int id = 0; id < nlog; id++) {
    const Tmp &t = tmp[lorder[id]];
    for (int r : t.verts) lverts[id].push_back(r), lnode_of_pos[r] = id;
  }
  // Variables with a structurally zero diagonal (equality multipliers; x_i without
  // Q_ii) can only be pivoted together with, or after, a neighbour that carries a
  // diagonal.  A maximum matching assigns every such variable a DISTINCT partner
  // (which makes every subtree's matrix structurally non-singular); a variable
  // whose partner lives in a proper ancestor is moved up into that node, and
  // inside a node it is ordered behind its partner, so that a 2x2 pivot with the
  // partner is always available inside the pivot block.  Neighbours are either
  // in the variable's subtree or in ancestors (separator property), and node ids
  // are a postorder, so "partner id > own id" means "proper ancestor"; moving a
  // vertex up its own root path keeps the assembly tree valid.
  std::vector<int> partner(dim, -1);
  {
    std::vector<char> has_diag(dim, 0);
    for (int e = 0; e < nent; e++)
      if (ent_a[e] == ent_b[e]) has_diag[ent_a[e]] = 1;
    auto node_of_q = [&](int q) { return lnode_of_pos[qp2j[q]]; };
    std::vector<int> used_by(dim, -1), pos2q(dim);
    for (int q = 0; q < dim; q++) pos2q[qp2j[q]] = q;
    // pass 1: free partner inside the own subtree (largest node id <= own)
    // pass 2: free partner in the lowest ancestor
    for (int pass = 1; pass <= 2; pass++)
      for (int r = 0; r < dim; r++) {
        const int q = pos2q[r];
        if (has_diag[q] || partner[q] >= 0) continue;
        const int sn = node_of_q(q);
        int best = -1, bestn = pass == 1 ? -1 : nlog;
        for (int k = gstart[q]; k < gstart[q + 1]; k++) {
          const int x = gneigh[k];
          if (!has_diag[x] || used_by[x] >= 0) continue;
          const int t = node_of_q(x);
          if (pass == 1 ? (t <= sn && t > bestn) : (t > sn && t < bestn)) best = x, bestn = t;
        }
        if (best >= 0) partner[q] = best, used_by[best] = q;
      }
    // pass 3: augmenting paths (Kuhn) for what is still unmatched
    {
      std::vector<int> seen(dim, -1), stack_z, stack_k, via(dim, -1);
      for (int r = 0; r < dim; r++) {
        const int q0 = pos2q[r];
        if (has_diag[q0] || partner[q0] >= 0 || gstart[q0 + 1] == gstart[q0]) continue;
        stack_z.assign(1, q0);
        stack_k.assign(1, gstart[q0]);
        int found = -1;
        while (!stack_z.empty() && found < 0) {
          const int zq = stack_z.back();
          int &k = stack_k.back();
          if (k >= gstart[zq + 1]) {
            stack_z.pop_back(), stack_k.pop_back();
            continue;
          }
          const int x = gneigh[k++];
          if (!has_diag[x] || seen[x] == q0) continue;
          seen[x] = q0;
          via[x] = zq;
          if (used_by[x] < 0)
            found = x;
          else {
            stack_z.push_back(used_by[x]);
            stack_k.push_back(gstart[used_by[x]]);
          }
        }
        while (found >= 0) {  // flip the path back to q0
          const int zq = via[found], prev = partner[zq];
          partner[zq] = found, used_by[found] = zq;
          found = (zq == q0) ? -1 : prev;
        }
      }
    }
    for (int r = 0; r < dim; r++) {
      const int q = pos2q[r];
      if (has_diag[q]) continue;
      const int sn = lnode_of_pos[r];
      int t = sn;
      bool all_diag = true;
      for (int k = gstart[q]; k < gstart[q + 1]; k++) all_diag = all_diag && has_diag[gneigh[k]];
      if (zd_policy == 2 && all_diag) {
        // behind ALL its neighbours: the node of the last eliminated neighbour.  The
        // pivot is then the complete Schur complement (e.g. A Q^-1 A' for an equality
        // multiplier), never a partial sum that is tiny against the border entries.
        for (int k = gstart[q]; k < gstart[q + 1]; k++) t = std::max(t, node_of_q(gneigh[k]));
      } else if (partner[q] >= 0) {
        t = node_of_q(partner[q]);
      }
      if (t <= sn) continue;  // (a node emptied by these moves is spliced out of the tree below)
      lverts[sn].erase(std::find(lverts[sn].begin(), lverts[sn].end(), r));
      lverts[t].push_back(r);
      lnode_of_pos[r] = t;
    }
    // order inside a node: band order; a zero-diagonal variable right behind its
    // partner (policy 0) or behind every variable that carries a diagonal (1, 2)
    std::vector<std::vector<int>> followers(dim);
    for (int id = 0; id < nlog; id++) {
      std::vector<int> &v = lverts[id];
      std::sort(v.begin(), v.end());
      std::vector<int> heads, tail;
      for (int r : v) {
        const int q = pos2q[r];
        const int pq = has_diag[q] ? -1 : partner[q];
        if (zd_policy != 0 && !has_diag[q])
          tail.push_back(r);
        else if (pq >= 0 && lnode_of_pos[qp2j[pq]] == id)
          followers[qp2j[pq]].push_back(r);
        else
          heads.push_back(r);
      }
      // FULL mode, optional: the slack rows (diagonal w/z, which goes to zero for active
      // constraints) behind the x variables of the node, so that -Q_ii is pivoted first
      // without a run-time interchange.  Inside a node the front is dense, so the order
      // is free; the fill of the chain of supernodes grows by ~10 %.
      if (slack_policy == 1 && mode == 0)
        std::stable_partition(heads.begin(), heads.end(), [&](int r) { return pos2q[r] < n + me; });
      // FULL mode, default: a slack row that stands in FRONT of one of its own x variables
      // (same node) is moved right behind the last of them.  The band order leaves that
      // to chance; with the slack row first and w/z small the Bunch-Kaufman test
      // interchanges the two at run time, which costs a trip through the slow path of
      // k_factor_diag for ~2 % of the pivots (C2).  ~3 % more fill: the slack row picks up
      // its x variable's pattern inside the chain of supernodes.
      if (slack_policy == 2 && mode == 0) {
        for (long t = 0; t < (long)heads.size(); t++) {
          const int q = pos2q[heads[t]];
          if (q < n + me) continue;
          long last = t;
          for (long u = t + 1; u < (long)heads.size(); u++) {
            const int qu = pos2q[heads[u]];
            if (qu >= n) continue;
            for (int k = gstart[q]; k < gstart[q + 1]; k++)
              if (gneigh[k] == qu) last = u;
          }
          if (last > t) {  // rotate heads[t] behind heads[last]
            const int r = heads[t];
            for (long u = t; u < last; u++) heads[u] = heads[u + 1];
            heads[last] = r;
            t--;  // the element that moved into place t has not been looked at
          }
        }
      }
      std::vector<int> out;
      for (int r : heads) {
        out.push_back(r);
        for (int f : followers[r]) out.push_back(f);
      }
      out.insert(out.end(), tail.begin(), tail.end());
      v.swap(out);
    }
  }

  TMARK("4");
  // chains of supernodes: each logical node is cut into pieces of <= max_pivots
  // pivots; a partner pair is never cut with the zero-diagonal variable first
  std::vector<std::vector<int>> nverts;
  std::vector<int> ltop(nlog, -1);
  std::vector<std::vector<int>> nkids;
  std::vector<std::vector<int>> lkids(nlog);  // children by logical id
  for (int id = 0; id < nlog; id++)
    for (int c : tmp[lorder[id]].kids) lkids[id].push_back(lid[c]);
  // a node whose rows all moved up to ancestors (small leaves of multipliers) leaves the tree:
  // its children hang on its parent, in its place
  std::vector<char> absorbed(nlog, 0);
  for (int id = 0; id < nlog; id++) {  // postorder: the children's lists are final
    std::vector<int> kids;
    for (int c : lkids[id])
      if (lverts[c].empty())
        kids.insert(kids.end(), lkids[c].begin(), lkids[c].end());
      else
        kids.push_back(c);
    lkids[id].swap(kids);
    if (lverts[id].empty()) absorbed[id] = 1;
  }
  // Narrow bands: a separator holds a handful of rows, and what a tree level costs there is
  // the latency of its launches, not its arithmetic.  A separator absorbs its child
  // separators while the merged pivot set still fits a small front: their rows first, each
  // node's rows in the order settled above, so the elimination order does not change; the
  // child separators are not coupled with each other, the merged pivot block has explicit
  // zeros.  The levels above the leaves shrink to a half or a third.
  if (amalgamation && small_fronts) {
    for (int id = nlog - 1; id >= 0; id--) {  // ids are a postorder: parents first this way
      if (absorbed[id]) continue;
      for (bool again = true; again;) {
        again = false;
        size_t tot = lverts[id].size();
        bool internal = false;
        for (int c : lkids[id])
          if (!lkids[c].empty()) tot += lverts[c].size(), internal = true;
        if (!internal || tot > (size_t)SMALL_PIVOTS) break;
        std::vector<int> verts, kids;
        for (int c : lkids[id]) {
          if (lkids[c].empty()) {
            kids.push_back(c);
            continue;
          }
          verts.insert(verts.end(), lverts[c].begin(), lverts[c].end());
          kids.insert(kids.end(), lkids[c].begin(), lkids[c].end());
          absorbed[c] = 1;
        }
        verts.insert(verts.end(), lverts[id].begin(), lverts[id].end());
        lverts[id].swap(verts), lkids[id].swap(kids);
        again = true;
      }
    }
  }
  for (int id = 0; id < nlog; id++) {
    if (absorbed[id]) continue;
    const std::vector<int> &v = lverts[id];
    // (separators in the high hundreds and beyond - the top of an irregular graph's tree - in longer pieces: every piece rewrites the
    //  whole update block of its front, 8 bytes per entry each way for 2 * pivots flops)
    const int len = (int)v.size(), mp = (long_chain_pivots > 0 && len >= LONG_CHAIN_VERTS) ? long_chain_pivots : max_pivots;
    const int parts = std::max(1, (len + mp - 1) / mp);
    const int size = (len + parts - 1) / parts;
    int top = -1;
    for (int s = 0; s < len || (len == 0 && s == 0); s += std::max(size, 1)) {
      std::vector<int> piece(v.begin() + s, v.begin() + std::min(s + size, len));
      std::vector<int> kids;
      if (top < 0)
        for (int c : lkids[id]) kids.push_back(ltop[c]);
      else
        kids.push_back(top);
      nverts.push_back(std::move(piece));
      nkids.push_back(std::move(kids));
      top = (int)nverts.size() - 1;
      if (len == 0) break;
    }
    ltop[id] = top;
  }
  nnodes = (int)nverts.size();
  piv_start.assign(nnodes, 0), npiv.assign(nnodes, 0), parent.assign(nnodes, -1);
  level.assign(nnodes, 0), child_slot.assign(nnodes, 0);
  std::vector<int> j2e(dim);
  {
    int e = 0;
    for (int id = 0; id < nnodes; id++) {
      piv_start[id] = e;
      npiv[id] = (int)nverts[id].size();
      for (int r : nverts[id]) j2e[r] = e++;
      for (size_t s = 0; s < nkids[id].size(); s++) {
        int c = nkids[id][s];
        parent[c] = id;
        child_slot[c] = (int)s;
        level[id] = std::max(level[id], level[c] + 1);
      }
    }
  }
  q2e.resize(dim), e2q.resize(dim);
  for (int q = 0; q < dim; q++) q2e[q] = j2e[qp2j[q]], e2q[q2e[q]] = q;
  std::vector<int> owner(dim);
  for (int id = 0; id < nnodes; id++)
    for (int k = 0; k < npiv[id]; k++) owner[piv_start[id] + k] = id;

  TMARK("5");
  // ------------------------------------------------------ symbolic fronts
  ent_er.resize(nent), ent_ec.resize(nent);
  std::vector<std::vector<int>> node_hi(nnodes);
  for (int e = 0; e < nent; e++) {
    int ea = q2e[ent_a[e]], eb = q2e[ent_b[e]];
    ent_ec[e] = std::min(ea, eb), ent_er[e] = std::max(ea, eb);
    int o = owner[ent_ec[e]];
    if (ent_er[e] >= piv_start[o] + npiv[o]) node_hi[o].push_back(ent_er[e]);
  }
  child_ptr.assign(nnodes + 1, 0);
  for (int id = 0; id < nnodes; id++)
    if (parent[id] >= 0) child_ptr[parent[id] + 1]++;
  for (int id = 0; id < nnodes; id++) child_ptr[id + 1] += child_ptr[id];
  child_idx.assign(child_ptr[nnodes], 0);
  {
    std::vector<int> fill(child_ptr.begin(), child_ptr.end() - 1);
    for (int id = 0; id < nnodes; id++)
      if (parent[id] >= 0) child_idx[child_ptr[parent[id]] + child_slot[id]] = id, fill[parent[id]]++;
  }
  bptr.assign(nnodes + 1, 0);
  bidx.clear();
  nbor.assign(nnodes, 0);
  {
    std::vector<int> stamp(dim, -1), cur;
    for (int id = 0; id < nnodes; id++) {
      cur.clear();
      const int pend = piv_start[id] + npiv[id];
      auto take = [&](int e) {
        if (e >= pend && stamp[e] != id) stamp[e] = id, cur.push_back(e);
      };
      for (int e : node_hi[id]) take(e);
      for (int k = child_ptr[id]; k < child_ptr[id + 1]; k++) {
        int c = child_idx[k];
        for (long long t = bptr[c]; t < bptr[c + 1]; t++) take(bidx[t]);
      }
      std::sort(cur.begin(), cur.end());
      nbor[id] = (int)cur.size();
      bidx.insert(bidx.end(), cur.begin(), cur.end());
      bptr[id + 1] = (long long)bidx.size();
      if (parent[id] < 0 && !cur.empty()) return 17;  // a root must have no border
    }
  }
  rel.assign(bidx.size(), -1);
  for (int id = 0; id < nnodes; id++) {
    int q = parent[id];
    if (q < 0) continue;
    const int qs = piv_start[q], qe = qs + npiv[q];
    const int *qb = bidx.data() + bptr[q];
    for (long long t = bptr[id]; t < bptr[id + 1]; t++) {
      int e = bidx[t];
      if (e < qs) return 17;
      if (e < qe)
        rel[t] = e - qs;
      else {
        const int *pos = std::lower_bound(qb, qb + nbor[q], e);
        if (pos == qb + nbor[q] || *pos != e) return 17;
        rel[t] = npiv[q] + (int)(pos - qb);
      }
    }
  }

  TMARK("6");
  // inverse of `rel`, per child: position in the parent's front -> position in the child's
  // border (or -1).  The consumers of a front (pivot block, panel solve, Schur update)
  // pull the children's update blocks through it - there is no extend-add pass.
  pinv_off.assign(nnodes, 0);
  {
    long long tot = 0;
    for (int id = 0; id < nnodes; id++)
      if (parent[id] >= 0) pinv_off[id] = tot, tot += npiv[parent[id]] + nbor[parent[id]];
    pinv.assign(tot, -1);
    for (int id = 0; id < nnodes; id++)
      if (parent[id] >= 0)
        for (long long t = bptr[id]; t < bptr[id + 1]; t++) pinv[pinv_off[id] + rel[t]] = (int)(t - bptr[id]);
  }

  // ------------------------------------------- shard plan (one system, P ranks)
  // The top of the assembly tree is replicated on every rank, the subtrees below
  // it are dealt to the ranks: walk down from the roots, always opening the
  // heaviest remaining subtree, until a longest-processing-time assignment of the
  // open subtrees is balanced.  With one rank everything belongs to rank 0.
  std::vector<long long> nflops(nnodes);
  for (int id = 0; id < nnodes; id++) {
    const long long p = npiv[id], b = nbor[id];
    nflops[id] = p * p * p / 3 + b * p * p + b * b * p;  // diag block, panel solve, update (lower half)
  }
  node_owner.assign(nnodes, 0);
  xroots.clear();
  if (shard_count > 1) {
    std::vector<long long> work(nflops);
    for (int id = 0; id < nnodes; id++)
      if (parent[id] >= 0) work[parent[id]] += work[id];
    std::vector<int> open;
    for (int id = 0; id < nnodes; id++)
      if (parent[id] < 0) open.push_back(id);
    std::vector<char> is_top(nnodes, 0);
    std::vector<int> assign;
    auto lpt = [&](std::vector<int> &who) -> double {  // returns max load / mean load
      std::vector<int> ord(open.size());
      std::iota(ord.begin(), ord.end(), 0);
      std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return work[open[x]] > work[open[y]]; });
      std::vector<long long> load(shard_count, 0);
      who.assign(open.size(), 0);
      long long tot = 0;
      for (int t : ord) {
        int q = (int)(std::min_element(load.begin(), load.end()) - load.begin());
        who[t] = q, load[q] += work[open[t]], tot += work[open[t]];
      }
      const long long mx = *std::max_element(load.begin(), load.end());
      return tot > 0 ? (double)mx * shard_count / (double)tot : 1.0;
    };
    for (;;) {
      std::sort(open.begin(), open.end());
      const double imb = lpt(assign);
      if ((int)open.size() >= shard_count && imb <= 1.10) break;
      if ((int)open.size() >= 16 * shard_count) break;
      int best = -1;
      for (size_t t = 0; t < open.size(); t++)
        if (child_ptr[open[t] + 1] > child_ptr[open[t]] && (best < 0 || work[open[t]] > work[open[best]]))
          best = (int)t;
      if (best < 0) break;  // only leaves left
      const int r = open[best];
      open.erase(open.begin() + best);
      is_top[r] = 1;
      for (int k = child_ptr[r]; k < child_ptr[r + 1]; k++) open.push_back(child_idx[k]);
    }
    std::vector<int> root_owner(nnodes, -2);
    for (size_t t = 0; t < open.size(); t++) root_owner[open[t]] = assign[t];
    for (int id = nnodes - 1; id >= 0; id--)
      node_owner[id] = is_top[id] ? -1 : (root_owner[id] != -2 ? root_owner[id] : node_owner[parent[id]]);
    // exchanged blocks: subtree roots that feed a replicated parent, grouped by owner
    for (int q = 0; q < shard_count; q++)
      for (int id : open)
        if (node_owner[id] == q && parent[id] >= 0) xroots.push_back(id);
  }

  TMARK("7");
  // --------------------------------------------------- storage + schedules
  panel_off.assign(nnodes, 0), upd_off.assign(nnodes, 0), x_off.assign(nnodes, 0);
  cb_off.assign(nnodes, 0);
  panel_elems = upd_elems = x_elems = cb_elems = 0;
  max_front = max_npiv = max_nbor = 0;
  nnz_factor = flops_factor = 0;
  nlevels = 0;
  std::vector<char> is_xroot(nnodes, 0);
  for (int id : xroots) is_xroot[id] = 1;
  // Update blocks: by default every supernode keeps its own b x b block for the whole
  // factorisation (one memset, nothing to manage).  When that would take more than
  // upd_pingpong_bytes, the blocks of a tree level live only until the next level has
  // consumed them: levels are re-assigned "as late as possible" (every child exactly one
  // level below its parent) and the blocks of even / odd levels alternate between two
  // halves of the arena, each as large as its fullest level.
  upd_pingpong = false;
  {
    long long tot = 0;
    for (int id = 0; id < nnodes; id++) tot += (long long)nbor[id] * nbor[id];
    upd_pingpong = shard_count <= 1 && upd_pingpong_bytes > 0 && 8 * tot > upd_pingpong_bytes;
  }
  if (upd_pingpong) {
    for (int id = nnodes - 1; id >= 0; id--)
      if (parent[id] >= 0) level[id] = level[parent[id]] - 1;
  }
  upd_level_off.clear(), upd_level_len.clear();
  std::vector<long long> lvl_fill;
  if (upd_pingpong) {
    int nl = 0;
    for (int id = 0; id < nnodes; id++) nl = std::max(nl, level[id] + 1);
    std::vector<long long> used(nl, 0);
    for (int id = 0; id < nnodes; id++) used[level[id]] += (long long)nbor[id] * nbor[id];
    long long cap[2] = {0, 0};
    for (int l = 0; l < nl; l++) cap[l & 1] = std::max(cap[l & 1], used[l]);
    upd_level_off.assign(nl, 0), upd_level_len = used;
    for (int l = 0; l < nl; l++) upd_level_off[l] = (l & 1) ? cap[0] : 0;
    lvl_fill = upd_level_off;
    upd_elems = cap[0] + cap[1];
  }
  for (int id = 0; id < nnodes; id++) {
    const long long p = npiv[id], b = nbor[id], F = p + b;
    panel_off[id] = panel_elems, panel_elems += F * p;
    if (upd_pingpong) {
      upd_off[id] = lvl_fill[level[id]], lvl_fill[level[id]] += b * b;
      cb_off[id] = cb_elems, cb_elems += b;
    } else if (!is_xroot[id]) {
      upd_off[id] = upd_elems, upd_elems += b * b;
      cb_off[id] = cb_elems, cb_elems += b;
    }
    x_off[id] = x_elems, x_elems += b * p;
    max_front = std::max<int>(max_front, (int)F);
    max_npiv = std::max<int>(max_npiv, (int)p);
    max_nbor = std::max<int>(max_nbor, (int)b);
    nlevels = std::max(nlevels, level[id] + 1);
    nnz_factor += p * (p + 1) / 2 + b * p;
    flops_factor += nflops[id];
  }
  // exchange regions behind the ordinary blocks: one slot per rank, a rank's
  // subtree roots packed inside its slot
  upd_x_off = upd_elems, cb_x_off = cb_elems, upd_x_slot = cb_x_slot = 0;
  if (!xroots.empty()) {
    std::vector<long long> ufill(shard_count, 0), cfill(shard_count, 0);
    for (int id : xroots) {
      const long long b = nbor[id];
      ufill[node_owner[id]] += b * b, cfill[node_owner[id]] += b;
    }
    upd_x_slot = *std::max_element(ufill.begin(), ufill.end());
    cb_x_slot = *std::max_element(cfill.begin(), cfill.end());
    upd_x_slot = (upd_x_slot + 1) & ~1LL, cb_x_slot = (cb_x_slot + 1) & ~1LL;
    std::fill(ufill.begin(), ufill.end(), 0), std::fill(cfill.begin(), cfill.end(), 0);
    for (int id : xroots) {
      const int q = node_owner[id];
      const long long b = nbor[id];
      upd_off[id] = upd_x_off + q * upd_x_slot + ufill[q], ufill[q] += b * b;
      cb_off[id] = cb_x_off + q * cb_x_slot + cfill[q], cfill[q] += b;
    }
    upd_elems += shard_count * upd_x_slot, cb_elems += shard_count * cb_x_slot;
  }
  linv_off.assign(nnodes, 0);
  linv_elems = 0;
  for (int id = 0; id < nnodes; id++) linv_off[id] = linv_elems, linv_elems += (long long)npiv[id] * npiv[id];

  // per-level work lists: [0] this rank's subtrees, [1] the replicated top
  for (int which = 0; which < 2; which++) {
    Sched &S = sched[which];
    S = Sched();
    auto keep = [&](int id) { return which == 0 ? node_owner[id] == shard_rank : node_owner[id] < 0; };
    S.level_ptr.assign(nlevels + 1, 0);
    for (int id = 0; id < nnodes; id++)
      if (keep(id)) S.level_ptr[level[id] + 1]++, S.nnodes++, S.flops += nflops[id];
    for (int l = 0; l < nlevels; l++) S.level_ptr[l + 1] += S.level_ptr[l];
    S.level_nodes.assign(S.nnodes, 0);
    // a "small front" (few pivots AND few border rows, the fronts of narrow-band
    // systems) is processed by one wavefront per supernode that does extend-add,
    // pivot block, panel and update in one kernel (and the solves likewise)
    auto fsmall = [&](int id) { return small_fronts && npiv[id] <= SMALL_PIVOTS && nbor[id] <= SMALL_BORDER; };
    // The one-wavefront pivot-block kernel pays where a level holds many blocks of a few pivots; a handful of them
    // (the remainder pieces of a cut separator next to blocks of 150 pivots) would be a launch of their own on the
    // critical path: they go with the general blocks of the level.
    std::vector<int> nsmall_level(nlevels, 0);
    for (int id = 0; id < nnodes; id++)
      if (keep(id) && !fsmall(id) && npiv[id] <= SMALL_PIVOTS) nsmall_level[level[id]]++;
    auto smallblk = [&](int id) { return !fsmall(id) && npiv[id] <= SMALL_PIVOTS && nsmall_level[level[id]] >= 64; };
    {
      // order inside a level: small fronts, then the other supernodes with at most
      // SMALL_PIVOTS pivots (one-wavefront pivot-block kernel), then the rest
      std::vector<int> fill(S.level_ptr.begin(), S.level_ptr.end() - 1);
      for (int id = 0; id < nnodes; id++)
        if (keep(id) && fsmall(id)) S.level_nodes[fill[level[id]]++] = id;
      S.level_fsmall.assign(nlevels, 0);
      for (int l = 0; l < nlevels; l++) S.level_fsmall[l] = fill[l] - S.level_ptr[l];
      for (int id = 0; id < nnodes; id++)
        if (keep(id) && smallblk(id)) S.level_nodes[fill[level[id]]++] = id;
      S.level_small.assign(nlevels, 0);
      for (int l = 0; l < nlevels; l++) S.level_small[l] = fill[l] - S.level_ptr[l] - S.level_fsmall[l];
      for (int id = 0; id < nnodes; id++)
        if (keep(id) && !fsmall(id) && !smallblk(id)) S.level_nodes[fill[level[id]]++] = id;
      // the general fronts of a level by falling number of pivots: a level with more fronts than the chip has CUs (one
      // pivot-block workgroup per CU) then starts its largest fronts first and fills up with the small ones
      for (int l = 0; l < nlevels; l++)
        std::stable_sort(S.level_nodes.begin() + S.level_ptr[l] + S.level_fsmall[l] + S.level_small[l], S.level_nodes.begin() + S.level_ptr[l + 1],
                         [&](int a, int b) { return npiv[a] != npiv[b] ? npiv[a] > npiv[b] : nbor[a] > nbor[b]; });
      S.level_fs_p.assign(nlevels, 1), S.level_fs_b.assign(nlevels, 1), S.level_sm_p.assign(nlevels, 1);
      for (int id = 0; id < nnodes; id++) {
        if (!keep(id) || npiv[id] > SMALL_PIVOTS) continue;
        const int l = level[id];
        if (fsmall(id))
          S.level_fs_p[l] = std::max(S.level_fs_p[l], npiv[id]), S.level_fs_b[l] = std::max(S.level_fs_b[l], nbor[id]);
        else if (smallblk(id))
          S.level_sm_p[l] = std::max(S.level_sm_p[l], npiv[id]);
      }
    }
    S.upd_tile_ptr.assign(nlevels + 1, 0), S.slab_ptr.assign(nlevels + 1, 0);
    S.gslab_ptr.assign(nlevels + 1, 0), S.cblk_ptr.assign(nlevels + 1, 0);
    S.upd_big_ptr.assign(nlevels, 0);
    // (HQPKKT_SCHUR_BIG_B: a test hook - 1 sends every front through k_schur_update_big)
    const int big_b = getenv("HQPKKT_SCHUR_BIG_B") ? atoi(getenv("HQPKKT_SCHUR_BIG_B")) : UPD_BIG_BORDER;
    for (int l = 0; l < nlevels; l++) {
      // the level's 64-tiles first, then the 128-tiles of its fronts with large borders
      for (int big = 0; big < 2; big++) {
        if (big) S.upd_big_ptr[l] = (int)S.upd_tiles.size() / 3;
        for (int t = S.level_ptr[l] + S.level_fsmall[l]; t < S.level_ptr[l + 1]; t++) {
          int id = S.level_nodes[t], b = nbor[id];
          if ((b >= big_b) != (big == 1)) continue;
          const int edge = big ? 2 * UPD_TILE : UPD_TILE;
          int nt = (b + edge - 1) / edge;
          for (int ti = 0; ti < nt; ti++)
            for (int tj = 0; tj <= ti; tj++) S.upd_tiles.insert(S.upd_tiles.end(), {id, ti, tj});
        }
      }
      for (int t = S.level_ptr[l] + S.level_fsmall[l]; t < S.level_ptr[l + 1]; t++) {
        int id = S.level_nodes[t], b = nbor[id];
        int ns = (b + SLAB_ROWS - 1) / SLAB_ROWS;
        for (int sl = 0; sl < ns; sl++) S.slabs.insert(S.slabs.end(), {id, sl});
        for (int sl = 0; sl < std::max(1, (b + 63) / 64); sl++) S.gslabs.insert(S.gslabs.end(), {id, sl});
        for (int c = 0; c < (npiv[id] + 15) / 16; c++) S.cblks.insert(S.cblks.end(), {id, c});
      }
      S.upd_tile_ptr[l + 1] = (int)S.upd_tiles.size() / 3;
      S.slab_ptr[l + 1] = (int)S.slabs.size() / 2;
      S.gslab_ptr[l + 1] = (int)S.gslabs.size() / 2;
      S.cblk_ptr[l + 1] = (int)S.cblks.size() / 2;
    }
  }
  // arena ranges this rank writes (zeroed at the start of every factorisation)
  // and the solution entries it contributes to the all-reduce of a sharded solve
  zero_panel.clear();
  keep_e.assign(dim, 0);
  {
    auto push = [](std::vector<long long> &v, long long off, long long len) {
      if (len <= 0) return;
      if (!v.empty() && v[v.size() - 2] + v.back() == off)
        v.back() += len;
      else
        v.push_back(off), v.push_back(len);
    };
    for (int id = 0; id < nnodes; id++) {
      const bool mine = node_owner[id] == shard_rank, top = node_owner[id] < 0;
      if (!mine && !top) continue;
      const long long p = npiv[id], b = nbor[id];
      push(zero_panel, panel_off[id], (p + b) * p);
      if (mine || shard_rank == 0)
        for (int k = 0; k < p; k++) keep_e[piv_start[id] + k] = 1;
    }
  }

  TMARK("8");
  // ------------------------------------------------------- assembly map
  ent_dst.resize(nent);
  for (int e = 0; e < nent; e++) {
    int o = owner[ent_ec[e]];
    const long long F = npiv[o] + nbor[o];
    int lc = ent_ec[e] - piv_start[o], lr;
    if (ent_er[e] < piv_start[o] + npiv[o])
      lr = ent_er[e] - piv_start[o];
    else {
      const int *b0 = bidx.data() + bptr[o];
      lr = npiv[o] + (int)(std::lower_bound(b0, b0 + nbor[o], ent_er[e]) - b0);
    }
    ent_dst[e] = panel_off[o] + (long long)lc * F + lr;
  }
  TMARK("9");
  // store the entries in destination order: the numeric scatter then writes
  // (mostly) consecutive addresses from consecutive threads
  {
    std::vector<int> perm(nent);
    {
      // destinations are distinct and ascend with (elimination column, row): a counting sort
      // by column, then each column's handful of entries by row
      std::vector<int> cptr(dim + 1, 0);
      for (int k = 0; k < nent; k++) cptr[ent_ec[k] + 1]++;
      for (int c = 0; c < dim; c++) cptr[c + 1] += cptr[c];
      {
        std::vector<int> fill(cptr.begin(), cptr.end() - 1);
        for (int k = 0; k < nent; k++) perm[fill[ent_ec[k]]++] = k;
      }
      for (int c = 0; c < dim; c++)
        if (cptr[c + 1] - cptr[c] > 1)
          std::sort(perm.begin() + cptr[c], perm.begin() + cptr[c + 1],
                    [&](int x, int y) { return ent_dst[x] < ent_dst[y]; });
    }
    TMARK("10");
    // (random gathers over 10^6..10^8 entries: cache misses are what they cost, so spread them over threads)
    auto apply_i = [&](std::vector<int> &v) {
      std::vector<int> t(nent);
      parallel_for(nent, [&](long long lo, long long hi) {
        for (long long k = lo; k < hi; k++) t[k] = v[perm[k]];
      });
      v.swap(t);
    };
    std::vector<int> new_ptr((size_t)nent + 1, 0);
    for (int k = 0; k < nent; k++) new_ptr[k + 1] = new_ptr[k] + (term_ptr[perm[k] + 1] - term_ptr[perm[k]]);
    std::vector<Term> new_terms(terms.size());
    parallel_for(nent, [&](long long lo, long long hi) {
      for (long long k = lo; k < hi; k++) {
        int o = new_ptr[k];
        for (int t = term_ptr[perm[k]]; t < term_ptr[perm[k] + 1]; t++) new_terms[o++] = terms[t];
      }
    });
    term_ptr.swap(new_ptr), terms.swap(new_terms);
    std::vector<long long> nd(nent);
    parallel_for(nent, [&](long long lo, long long hi) {
      for (long long k = lo; k < hi; k++) nd[k] = ent_dst[perm[k]];
    });
    ent_dst.swap(nd);
    apply_i(ent_a), apply_i(ent_b), apply_i(ent_er), apply_i(ent_ec);
    std::vector<int> inv(nent);
    parallel_for(nent, [&](long long lo, long long hi) {
      for (long long k = lo; k < hi; k++) inv[perm[k]] = (int)k;
    });
    for (int i = 0; i < n; i++)
      if (diag_ent[i] >= 0) diag_ent[i] = inv[diag_ent[i]];
  }
  TMARK("11");
  return 0;
}

}  // namespace kktdev
