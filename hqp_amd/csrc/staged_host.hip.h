// Host side of the STAGED engine: device residency of the stage blocks and the kernel
// sequences of factor (backward recursion over the stages) and step (backward vector sweep,
// initial state, forward sweep).  Included by hqpkkt.hip after struct hqpkkt.
// Reference counterpart: Hqp_IpLQDOCP::update / factor / step (hqp/Hqp_IpLQDOCP.C:722-976).
#pragma once

struct StagedDev {
  kktdev::StagedPlan plan;
  DBuf<double> F, V, misc;
  DBuf<int> dyn, eq_rows, fix_rows, fix_src, h_tptr, chk_idx, chk_kind;
  DBuf<long long> h_dst, a_dst;
  DBuf<stg::HTerm> h_terms;
  DBuf<stg::DynDesc> dyn_desc;  // dense dynamics: per stage (K+1) what k_st_dyn_ax / _aty need
  DBuf<double> dyn_x1, dyn_x2;  // A_dyn' dy (n), A_dyn dx (ndyn)
  DBuf<double> dyn_part;        // row sums of A_dyn dx per block of 256 columns (k_st_dyn_both): ndyn x dyn_part_cols
  int dyn_part_cols = 0;
  double *hblk[2] = {nullptr, nullptr};  // pinned staging of one stage block each (hqpkkt_stage_staging)
  long long hblk_elems = 0;
  hipEvent_t hblk_ev[2] = {nullptr, nullptr};
  std::vector<char> blocks_set;         // dense hand-over block by block: which stages have arrived since the analysis
  // one system over several ranks (staged_plan.hpp): per stage where the ranks' strips of W / blocks of G_xx lie in the
  // exchange buffers, this rank's tiles of its blocks' products and its blocks to pack; the local dynamics blocks of
  // residuum()'s products and their summed results (A_dyn' dy: n, A_dyn dx: ndyn)
  DBuf<stg::StripTab> wtabs;  // (the strips of the gathered F: offsets inside one of the two buffers)
  DBuf<stg::RectTab> rtabs;
  DBuf<int> gtile;
  DBuf<unsigned> gowned;             // per stage: bitmap of the 128 x 128 tiles of G this rank computes (k_st_add_h_owned)
  std::vector<long long> gowned_off;  // stage k's words start at gowned_off[k]
  DBuf<stg::PackRect> prects;
  std::vector<int> prect_ptr;
  DBuf<stg::DynLoc> dyn_loc;
  DBuf<double> dyn_sum;
  long long dyn_sum_x2 = 0;  // offset of A_dyn dx in dyn_sum
  // the exchanges of a stage in the stream-ordered form (RCCL) go to a stream of their own, so that the gather of the
  // NEXT stage's F blocks travels beside this stage's products: ev_w[i] "the first stream is ready for exchange i",
  // ev_x[i] "exchange i has arrived" (i = 0, 1: the gathered F in buffer i, 2: the blocks of G_xx)
  hipStream_t stream_x = nullptr;
  hipEvent_t ev_x1 = nullptr, ev_w[3] = {nullptr, nullptr, nullptr}, ev_x[3] = {nullptr, nullptr, nullptr};
  // The solve's products with V that stand outside its two chains, many stages per launch (k_st_symv_*_batch; not
  // sharded): [0] g_k = V_{k+1} f_k ahead of the backward sweep, [1] the dynamics rows' multipliers behind the forward
  // sweep.  A launch holds the stages whose partial sums fit StagedPlan::symb_elems; stages that do not take the
  // triangle form (st_symv) go through k_st_gemv_rows one by one.
  struct SymvGroup {
    int first, count, tiles, fins;
    int kfirst, klast;  // the stages k (products with V_{k+1}) of the launch
  };
  DBuf<stg::SymvItem> symv_items[2];
  std::vector<SymvGroup> symv_groups[2];
  std::vector<int> symv_rows_stage[2];
  DBuf<double> zeros;           // 256 zero doubles: the operand rows k >= K of the LDS-DMA staging (GemmArgs::zeros)
  int gemm_variant = stg::GEMM_DMA8;
  int cus = 0;
  DBuf<double> ks_ws2;          // the pieces of a thin product cut in k (k_dgemm_tn_ks) launched on the SECOND stream
  long long ks_ws2_elems = 0;
  DBuf<double> sk_ws;           // stream-K dgemm: two partial tiles per workgroup
  DBuf<unsigned> sk_cnt;
  int sk_grid = 0;              // workgroups of the stream-K grid (2 per CU); 0: not used
  int sk_tiles = 0;             // most tiles a product of this handle has (size of the counter array)
  // second stream: the control-sized chain of a stage (G_u strip, H's control part, carried rows, K^-1, Y, Rm)
  // runs beside the large product G_xx = fx'W_x instead of behind it (fork / join by events; inside a
  // captured sequence these are parallel branches of the graph)
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool overlap = false;
  int overlap_mode = 0;  // 0 never, 1 every stage, 2 stages of 1280 .. 4096 states
  // order of the tiles of a lower-triangular product with T tile rows (GemmArgs::tile_map), by T
  std::vector<std::pair<int, DBuf<int> *>> tri_maps;
  const int *tri_map(int T, bool create = false) {
    for (auto &e : tri_maps)
      if (e.first == T) return e.second->p;
    if (!create) return nullptr;  // (made at upload time: no allocation inside a captured sequence)
    std::vector<int> m;
    m.reserve((size_t)T * (T + 1) / 2);
    const int S = 8;  // super-blocks of 8 x 8 tiles, row by row; inside a block column by column
    for (int I = 0; I < (T + S - 1) / S; I++)
      for (int J = 0; J <= I; J++)
        for (int tn = J * S; tn < std::min(T, (J + 1) * S); tn++)
          for (int tm = std::max(I * S, tn); tm < std::min(T, (I + 1) * S); tm++) m.push_back(tm << 16 | tn);
    DBuf<int> *b = new (std::nothrow) DBuf<int>;
    if (!b || b->upload(m)) {
      delete b;
      return nullptr;
    }
    tri_maps.push_back({T, b});
    return b->p;
  }
  // work lists of the cut form of the large products (stg::gemm_split_table: unequal shares for the two workgroups of a
  // CU), by (tiles, k-slabs); made at upload time for the shapes of the recursion - a shape without one (or
  // HQPKKT_SK_TABLE=0) runs the equal-share plan
  struct SkTab {
    long long tiles, nslab, pieces;
    int stride;
    DBuf<stg::SkUnit> *units;
  };
  std::vector<SkTab> sk_tabs;
  bool sk_tables_on = true;
  const SkTab *sk_tab(long long tiles, long long nslab, bool create = false) {
    if (!sk_tables_on || sk_grid <= 0) return nullptr;
    for (auto &e : sk_tabs)
      if (e.tiles == tiles && e.nslab == nslab) return e.units ? &e : nullptr;
    if (!create) return nullptr;  // (no allocation inside a captured sequence)
    stg::SplitTable t;
    SkTab e{tiles, nslab, 0, 0, nullptr};
    if (stg::gemm_split_table(tiles, nslab, sk_grid, t) && t.pieces * 128LL * 128 <= sk_ws_elems && tiles <= sk_tiles) {
      DBuf<stg::SkUnit> *b = new (std::nothrow) DBuf<stg::SkUnit>;
      if (b && !b->upload(t.units))
        e.units = b, e.stride = t.stride, e.pieces = t.pieces;
      else
        delete b;
    }
    sk_tabs.push_back(e);
    return e.units ? &sk_tabs.back() : nullptr;
  }
  // (the shape of a product as st_gemm launches it)
  void sk_tab_prepare(int M, int N, int K, int lower) {
    if (M <= 0 || N <= 0 || K <= 0 || sk_grid <= 0 || (lower && M < N)) return;
    if (!stg::gemm_use_split(M, N, K, lower, sk_grid) || (!plan.sharded && stg::gemm_use_frac(M, N, K, lower, sk_grid))) return;
    (void)sk_tab(stg::gemm_tiles(M, N, 128, lower), (K + stg::GEMM_BK - 1) / stg::GEMM_BK, true);
  }
  size_t lds_small = 0, lds_small_big = 0, lds_init = 0, lds_x0 = 0;
  long long sk_ws_elems = 0, sk_cnt_elems = 0;
  void release() {
    F.release(), V.release(), misc.release();
    dyn.release(), eq_rows.release(), fix_rows.release(), fix_src.release(), h_tptr.release();
    chk_idx.release(), chk_kind.release(), h_dst.release(), a_dst.release(), h_terms.release();
    dyn_desc.release(), dyn_x1.release(), dyn_x2.release(), dyn_part.release(), sk_ws.release(), ks_ws2.release(), sk_cnt.release(), zeros.release();
    wtabs.release(), rtabs.release(), gtile.release(), gowned.release(), prects.release(), dyn_loc.release(), dyn_sum.release();
    if (stream_x) (void)hipStreamDestroy(stream_x), stream_x = nullptr;
    for (hipEvent_t *ev : {&ev_w[0], &ev_w[1], &ev_w[2], &ev_x[0], &ev_x[1], &ev_x[2]})
      if (*ev) (void)hipEventDestroy(*ev), *ev = nullptr;
    if (ev_x1) (void)hipEventDestroy(ev_x1), ev_x1 = nullptr;
    for (int b = 0; b < 2; b++) {
      if (hblk[b]) (void)hipHostFree(hblk[b]), hblk[b] = nullptr;
      if (hblk_ev[b]) (void)hipEventDestroy(hblk_ev[b]), hblk_ev[b] = nullptr;
    }
    hblk_elems = 0, blocks_set.clear();
    for (auto &e : tri_maps) e.second->release(), delete e.second;
    tri_maps.clear();
    for (auto &e : sk_tabs)
      if (e.units) e.units->release(), delete e.units;
    sk_tabs.clear();
    if (stream2) (void)hipStreamDestroy(stream2), stream2 = nullptr;
    if (ev_fork) (void)hipEventDestroy(ev_fork), ev_fork = nullptr;
    if (ev_join) (void)hipEventDestroy(ev_join), ev_join = nullptr;
  }
};

namespace {

// per-stage pointers into the arenas
struct StagePtr {
  double *F, *V, *Y, *Rm, *Kinv, *Kmat, *N, *BT, *T, *v, *beta, *eta, *rho;
  int *dyn;
  double *Vs;  // sharded: this rank's row strip of V_k (V: the transient full block; F: the local block [F_p | F_u])
};
inline StagePtr stage_ptr(StagedDev &d, int k) {
  const kktdev::StagedPlan &P = d.plan;
  StagePtr s{};
  double *M = d.misc.p;
  s.V = P.sharded ? M + P.oVf[k & 1] : d.V.p + P.oV[k];
  s.Vs = P.sharded ? d.V.p + P.oVs[k] : nullptr;
  s.BT = M + P.oBT[k], s.N = M + P.oN[k];
  const int capx = std::max(P.cap[k], 1);
  s.v = M + P.oVec[k], s.beta = s.v + P.nk[k], s.eta = s.beta + capx, s.rho = s.eta + capx;
  s.dyn = d.dyn.p + P.dyn_off[k];
  if (k < P.K) {
    s.F = d.F.p + (P.sharded ? P.oFl[k] : P.oF[k]);
    s.Y = M + P.oY[k], s.Rm = M + P.oR[k], s.Kinv = M + P.oK[k], s.Kmat = M + P.oKm[k], s.T = M + P.oT[k];
  }
  return s;
}

// C = alpha A'B + beta Cin on the handle's stream; 128 x 128 tiles for large products, 64 x 64 below
int st_gemm(hqpkkt_t *h, stg::GemmArgs g, int cls = KC_ST_GEMM, bool allow_sk = true) {
  if (g.M <= 0 || g.N <= 0) return 0;
  StagedDev *d = h->sd;
  // operands by LDS-DMA (global_load_lds_dwordx4) only from 16-byte aligned rows: an operand that starts at an odd
  // column (the control columns F + nn of a stage with an odd number of states) is staged through registers
  const bool al16 = ((((uintptr_t)g.A | (uintptr_t)g.B) & 15) == 0) && (((g.lda | g.ldb) & 1) == 0);
  // (allow_sk false: launches of the second stream - the workspace of the split form belongs to the first)
  const bool no_frac = false;
  // (not for one system over several ranks: there the strip product W_p = V+ F_p - 200 tiles at eight ranks - runs beside
  // the second stream's control-sized products, and a launch whose 512 workgroups hold every CU for its whole duration
  // starves them: 1.46 against 1.32 ms per stage, tools/shard_pieces.py 8 0)
  const bool frac = d && allow_sk && d->sk_grid > 0 && !no_frac && !d->plan.sharded && stg::gemm_use_frac(g.M, g.N, g.K, g.lower, d->sk_grid) &&
                    2LL * d->sk_grid * 128 * 128 <= d->sk_ws_elems;
  const bool split = frac || (d && allow_sk && d->sk_grid > 0 && stg::gemm_use_split(g.M, g.N, g.K, g.lower, d->sk_grid));
  const bool big = split || stg::gemm_big_tiles(g.M, g.N, g.lower, g.K);
  const int b = big ? 128 : 64;
  const long long tm = (g.M + b - 1) / b;
  const long long tiles = stg::gemm_tiles(g.M, g.N, b, g.lower);
  if (g.lower && g.M < g.N) return HQPKKT_E_INTERN;  // (lower: a triangle, or the column strip of one)
  if (g.lower && g.M == g.N && big && d && tm >= 16 && tm < 32768) g.tile_map = d->tri_map((int)tm);
  if (d && d->zeros.p && al16) g.zeros = d->zeros.p;
  if (split && tiles <= d->sk_tiles) {
    // tile count that does not fill the chip evenly: whole rounds, then the k ranges of the rest cut (k_dgemm_tn_sk)
    // (the arrival counters are zero between launches: the last arriver of a tile resets its counter)
    // (a few hundred tiles: the k-slabs of all tiles in one sequence, an equal share per workgroup - gemm_use_frac)
    stg::SplitPlan sk = frac ? stg::gemm_split_plan_frac(tiles, (g.K + stg::GEMM_BK - 1) / stg::GEMM_BK, d->sk_grid)
                             : stg::gemm_split_plan(tiles, (g.K + stg::GEMM_BK - 1) / stg::GEMM_BK, d->sk_grid);
    const StagedDev::SkTab *tab = frac ? nullptr : d->sk_tab(tiles, (g.K + stg::GEMM_BK - 1) / stg::GEMM_BK, !h->capturing);
    if (tab) sk.table = tab->units->p, sk.stride = tab->stride;
    if (frac || tab || stg::gemm_split_plan_pieces(sk) * 128LL * 128 <= d->sk_ws_elems) {
      sk.ws = d->sk_ws.p, sk.cnt = d->sk_cnt.p;
      KLAUNCH(h, cls, stg::gemm_launch_split(d->gemm_variant, d->sk_grid, h->stream, g, sk));
      return 0;
    }
  }
  if (big)
    KLAUNCH(h, cls, stg::gemm_launch_plain(d ? d->gemm_variant : stg::GEMM_REG4, (unsigned)tiles, h->stream, g, d ? d->cus : 0));
  else if (d && d->cus > 0 && !g.lower && !g.mirror && g.K >= 512 && tiles * 2 <= d->cus &&
           (long long)g.M * g.N * 4 <= (allow_sk ? d->sk_ws_elems : d->ks_ws2_elems)) {
    // a thin, deep product: its k range cut over the chip (k_dgemm_tn_ks), the pieces added in their order (the
    // launches of the second stream have a workspace of their own)
    double *ws = allow_sk ? d->sk_ws.p : d->ks_ws2.p;
    const long long wse = allow_sk ? d->sk_ws_elems : d->ks_ws2_elems;
    const int nslab = (g.K + stg::GEMM_BK - 1) / stg::GEMM_BK;
    int nsplit = (int)std::min<long long>(nslab / 4, std::max<long long>(1, (2LL * d->cus) / tiles));
    nsplit = (int)std::max<long long>(1, std::min<long long>(nsplit, wse / std::max<long long>(1, (long long)g.M * g.N)));
    KLAUNCH(h, cls, (stg::k_dgemm_tn_ks<64, 64><<<dim3((unsigned)tiles, nsplit), 256, stg::gemm_lds_bytes(64, 64), h->stream>>>(g, ws, nsplit)));
    KLAUNCH(h, cls, stg::k_dgemm_ks_finish<<<nblk((long long)g.M * g.N), 256, 0, h->stream>>>(g, ws, nsplit));
  } else {
    // few tiles of a deep rectangular product (W of a stage of ~1000 states: 272): 64 x 32 tiles, so that a CU holds two
    // workgroups and one multiplies while the other waits at its barrier: 81 -> 73 us
    if (stg::gemm_tiles_6432(g.M, g.N, g.K, g.lower, g.mirror, d ? d->cus : 0)) {
      const long long t2 = ((g.M + 63) / 64) * (long long)((g.N + 31) / 32);
      KLAUNCH(h, cls, (stg::k_dgemm_tn<64, 32><<<(unsigned)t2, 256, stg::gemm_lds_bytes(64, 32), h->stream>>>(g)));
      return 0;
    }
    KLAUNCH(h, cls, stg::k_dgemm_tn<64, 64><<<(unsigned)tiles, 256, stg::gemm_lds_bytes(64, 64), h->stream>>>(g));
  }
  return 0;
}

int st_gemv_rows(hqpkkt_t *h, stg::GemvRows g) {
  if (g.M <= 0) return 0;
  KLAUNCH(h, KC_ST_VEC, stg::k_st_gemv_rows<<<(g.M + 3) / 4, 256, 0, h->stream>>>(g));
  return 0;
}
// y = scale (add + V x + A2 x2) with the symmetric V of a stage: from 2048 states on only the tiles on and below the
// diagonal are read (k_st_symv_tiles + k_st_symv_finish; HQPKKT_NO_SYMV: the rows form throughout)
bool symv_tiles_form(const stg::GemvRows &g) {
  static const bool off = getenv("HQPKKT_NO_SYMV") != nullptr;
  static const int from = getenv("HQPKKT_SYMV_FROM") ? atoi(getenv("HQPKKT_SYMV_FROM")) : 2048;
  return !(off || g.M != g.N || g.N < from || (g.lda & 1) || (((size_t)g.A) & 15));
}
long long symv_tiles(int N) {
  const int nrt = (N + stg::SV_R - 1) / stg::SV_R;
  long long tiles = 0;
  for (int bi = 0; bi < nrt; bi++) tiles += bi / (stg::SV_C / stg::SV_R) + 1;
  return tiles;
}
int st_symv(hqpkkt_t *h, StagedDev &d, stg::GemvRows g) {
  if (!symv_tiles_form(g)) return st_gemv_rows(h, g);
  const kktdev::StagedPlan &P = d.plan;
  const int N = g.N, nct = (N + stg::SV_C - 1) / stg::SV_C;
  double *rowpart = d.misc.p + P.oSym, *colpart = rowpart + (long long)nct * N;
  static_assert(stg::SV_R == 64 && stg::SV_C == 512, "StagedPlan::oSym is sized for these tiles");
  const long long tiles = symv_tiles(N);
  KLAUNCH(h, KC_ST_VEC, stg::k_st_symv_tiles<<<(unsigned)tiles, 256, 0, h->stream>>>(stg::SymvArgs{g.A, g.lda, N, g.x, rowpart, colpart}));
  KLAUNCH(h, KC_ST_VEC, stg::k_st_symv_finish<<<(N + 63) / 64, 256, 0, h->stream>>>(
                            stg::SymvFinish{N, rowpart, colpart, g.add, g.A2, g.lda2, g.n2, g.x2, g.y, g.scale}));
  return 0;
}
// y = add + alpha A'x over a K x N row-major block
// (add2, y2: a second result y2 = y + add2)
int st_gemv_cols(hqpkkt_t *h, StagedDev &d, const double *A, long long lda, int K, int N, const double *x,
                 const double *add, double alpha, double *y, const double *add2 = nullptr, double *y2 = nullptr) {
  if (N <= 0) return 0;
  const kktdev::StagedPlan &P = d.plan;
  int chunks = std::max(1, std::min(P.part_chunks, K / 64));
  stg::GemvCols g{A, lda, K, N, x, add, alpha, y, d.misc.p + P.oPart, (K + chunks - 1) / chunks, add2, y2};
  KLAUNCH(h, KC_ST_VEC, stg::k_st_gemv_cols<<<dim3((N + 511) / 512, chunks), 256, 0, h->stream>>>(g));
  if (chunks > 1)
    KLAUNCH(h, KC_ST_VEC, stg::k_st_cols_finish<<<(N + 255) / 256, 256, 0, h->stream>>>(N, chunks, d.misc.p + P.oPart, add,
                                                                                       alpha, y, add2, y2));
  return 0;
}

// the two tables of StagedDev::symv_items (at upload time: the arenas' addresses are final)
int staged_build_symv_tables(StagedDev &d) {
  const kktdev::StagedPlan &P = d.plan;
  for (int dir = 0; dir < 2; dir++) d.symv_items[dir].release(), d.symv_groups[dir].clear(), d.symv_rows_stage[dir].clear();
  if (P.sharded) return 0;
  double *M = d.misc.p, *S = M + P.oS, *gv = M + P.oGv;
  auto up16 = [](long long x) { return (x + 15) / 16 * 16; };
  for (int dir = 0; dir < 2; dir++) {
    std::vector<stg::SymvItem> items;
    StagedDev::SymvGroup g{0, 0, 0, 0, 0, 0};
    long long used = 0;
    for (int k = 0; k < P.K; k++) {
      const StagePtr sn = stage_ptr(d, k + 1);
      const int np = P.nk[k + 1];
      if (np <= 0) continue;
      const stg::GemvRows gr{sn.V, P.ldv[k + 1], np, np, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 1.0};
      if (!symv_tiles_form(gr)) {
        d.symv_rows_stage[dir].push_back(k);
        continue;
      }
      const long long need = up16(kktdev::StagedPlan::symv_need(np));
      if (g.count > 0 && used + need > P.symb_elems) {
        d.symv_groups[dir].push_back(g);
        g = StagedDev::SymvGroup{(int)items.size(), 0, 0, 0, k, k}, used = 0;
      }
      if (g.count == 0) g.kfirst = k;
      g.klast = k;
      const int nct = (np + stg::SV_C - 1) / stg::SV_C;
      double *rowpart = M + P.oSymB + used, *colpart = rowpart + (long long)nct * np;
      stg::SymvItem it{};
      it.a = stg::SymvArgs{sn.V, P.ldv[k + 1], np, dir == 0 ? nullptr : S + P.nmk[k + 1], rowpart, colpart};
      if (dir == 0)
        it.f = stg::SymvFinish{np, rowpart, colpart, nullptr, nullptr, 0, nullptr, nullptr, gv + P.nks[k], 1.0};
      else
        it.f = stg::SymvFinish{np, rowpart, colpart, sn.v, P.cap[k + 1] > 0 ? sn.BT : nullptr, P.ldb[k + 1], sn.dyn + 1, sn.eta, nullptr, 1.0};
      it.tile0 = g.tiles, it.fin0 = g.fins;
      it.xrel = dir == 0, it.yrel = dir == 1, it.xoff = it.yoff = P.nks[k];
      items.push_back(it);
      g.tiles += (int)symv_tiles(np), g.fins += (np + 63) / 64, g.count++, used += need;
    }
    if (g.count > 0) d.symv_groups[dir].push_back(g);
    if (!items.empty())
      if (int e = d.symv_items[dir].upload(items)) return e;
  }
  return 0;
}
// one launch pair of StagedDev::symv_groups[dir] on h->stream.  dir 0: g_k = V_{k+1} f_k (f: the dynamics rows of r2) into
// the plan's oGv; dir 1: the dynamics rows of dy = V+ x+ + v+ + B+' eta+
int staged_symv_group(hqpkkt_t *h, StagedDev &d, int dir, int gi, const double *r2, double *dy) {
  const StagedDev::SymvGroup &g = d.symv_groups[dir][gi];
  KLAUNCH(h, KC_ST_VEC, stg::k_st_symv_tiles_batch<<<(unsigned)g.tiles, 256, 0, h->stream>>>(d.symv_items[dir].p + g.first, g.count, g.tiles, r2));
  KLAUNCH(h, KC_ST_VEC, stg::k_st_symv_finish_batch<<<(unsigned)g.fins, 256, 0, h->stream>>>(d.symv_items[dir].p + g.first, g.count, g.fins, dy));
  return 0;
}
// ... and the stages outside the groups (no triangle form: few states), one launch each
int staged_symv_rows(hqpkkt_t *h, StagedDev &d, int dir, const double *r2, double *dy) {
  const kktdev::StagedPlan &P = d.plan;
  double *M = d.misc.p, *S = M + P.oS, *gv = M + P.oGv;
  for (int k : d.symv_rows_stage[dir]) {
    const StagePtr sn = stage_ptr(d, k + 1);
    const int np = P.nk[k + 1];
    int e;
    if (dir == 0)
      e = st_gemv_rows(h, stg::GemvRows{sn.V, P.ldv[k + 1], np, np, r2 + P.nks[k], nullptr, nullptr, 0, nullptr, nullptr, gv + P.nks[k], 1.0});
    else
      e = st_gemv_rows(h, stg::GemvRows{sn.V, P.ldv[k + 1], np, np, S + P.nmk[k + 1], sn.v, P.cap[k + 1] > 0 ? sn.BT : nullptr, P.ldb[k + 1],
                                        sn.dyn + 1, sn.eta, dy + P.nks[k], 1.0});
    if (e) return e;
  }
  return 0;
}

}  // namespace

// The control-sized elimination of a stage whose matrices live in global memory: phase (A) and the scaled K by the
// one-workgroup kernel, the inverse by the blocked sweep on the whole chip (k_blk_*, staged.hip.h), its check against
// K, and the one-workgroup inverse behind it in case the blocked one gave up (decided on the device: flags[0]).
// the sweep over the pivot blocks of 64 down the diagonal of the scaled matrix in `scratch` (layout: stg::big_scratch)
static int st_blk_sweep(hqpkkt_t *h, double *scratch, int q, bool allow_sk) {
  const stg::BigScratch bs = stg::big_scratch(scratch, q);
  const int nb = (q + 63) / 64;
  int e;
  for (int j = 0; j < nb; j++) {
    const stg::BlkArgs ba{scratch, q, j};
    KLAUNCH(h, KC_ST_SMALL, stg::k_blk_pivot<<<1, 256, 0, h->stream>>>(ba));
    if ((e = st_gemm(h, stg::GemmArgs{bs.Pb, 64, bs.T0, bs.ldk, nullptr, 0, bs.R, bs.ldk, 64, q, 64, 1.0, 0.0, 0, 0}, KC_ST_GEMM_UPD, allow_sk)) ||
        (e = st_gemm(h, stg::GemmArgs{bs.T0, bs.ldk, bs.R, bs.ldk, bs.Ks, bs.ldk, bs.Ks, bs.ldk, q, q, 64, -1.0, 1.0, 0, 0}, KC_ST_GEMM_UPD, allow_sk)))
      return e;
    KLAUNCH(h, KC_ST_SMALL, stg::k_blk_fixup<<<nblk(64LL * q), 256, 0, h->stream>>>(ba));
  }
  return 0;
}
static int st_small_big(hqpkkt_t *h, StagedDev &d, stg::SmallArgs sa, bool allow_sk) {
  const bool legacy = false;  // (the one-workgroup inverse: the fall-back of the blocked sweep only)
  static const double tol = getenv("HQPKKT_BLOCK_GJ_TOL") ? atof(getenv("HQPKKT_BLOCK_GJ_TOL")) : 1e-6;
  if (legacy) {
    sa.mode = 0;
    KLAUNCH(h, KC_ST_SMALL, stg::k_st_small<1024><<<1, 1024, d.lds_small_big, h->stream>>>(sa));
    return 0;
  }
  int e;
  sa.mode = 1;
  KLAUNCH(h, KC_ST_SMALL, stg::k_st_small<1024><<<1, 1024, d.lds_small_big, h->stream>>>(sa));
  const stg::BigScratch bs = stg::big_scratch(sa.scratch, sa.qmax);
  const int q = sa.qmax;
  if ((e = st_blk_sweep(h, sa.scratch, q, allow_sk))) return e;
  KLAUNCH(h, KC_ST_SMALL, stg::k_blk_final<<<nblk((long long)q * q), 256, 0, h->stream>>>(sa));
  if ((e = st_gemm(h, stg::GemmArgs{sa.Kmat, sa.ldq, sa.Kinv, sa.ldq, nullptr, 0, bs.Ks, bs.ldk, q, q, q, 1.0, 0.0, 0, 0}, KC_ST_GEMM_UPD, allow_sk)))
    return e;
  KLAUNCH(h, KC_ST_SMALL, stg::k_blk_check<<<1, 1024, 0, h->stream>>>(sa, tol));
  sa.mode = 2;
  KLAUNCH(h, KC_ST_SMALL, stg::k_st_small<1024><<<1, 1024, d.lds_small_big, h->stream>>>(sa));
  return 0;
}

// Rm = K^-1 Y, refined against K: one launch for K of order <= 64 (k_st_rm), three products above
// (wa: the arguments of k_st_wide - Y and the carried rows B_k; with K of order 1 .. 64 they are formed inside k_st_rm)
static int st_rm(hqpkkt_t *h, StagedDev &d, const StagePtr &sp, int k, bool allow_sk, const stg::WideArgs &wa) {
  const kktdev::StagedPlan &P = d.plan;
  const int q = P.qmax[k], nn = P.nk[k];
  const long long ldy = P.ldy[k];
  if (q > 0 && q <= 64) {
    stg::RmArgs ra{sp.Kinv, sp.Kmat, P.ldq[k], sp.Y, sp.Rm, ldy, q, nn, 1, wa};
    KLAUNCH(h, KC_ST_GEMM_UPD, stg::k_st_rm<<<(nn + stg::RM_COLS - 1) / stg::RM_COLS, 256, stg::st_rm_lds(q), h->stream>>>(ra));
    return 0;
  }
  KLAUNCH(h, KC_ST_SMALL, stg::k_st_wide<<<nblk(nn), 256, 0, h->stream>>>(wa));
  if (q <= 0) return 0;
  // one round of refinement against K: Rm += K^-1 (Y - K Rm).  The product with an explicit inverse alone
  // leaves a residual of cond(K) eps |Y| where the reference's solve by Bunch-Kaufman factors
  // (hqp/Hqp_IpLQDOCP.C:1866-1869, 1911-1924) leaves eps |K| |Rm|; stiff stages need the latter
  double *Res = d.misc.p + P.oRes;
  int e;
  if ((e = st_gemm(h, stg::GemmArgs{sp.Kinv, P.ldq[k], sp.Y, ldy, nullptr, 0, sp.Rm, ldy, q, nn, q, 1.0, 0.0, 0, 0}, KC_ST_GEMM_UPD, allow_sk)) ||
      (e = st_gemm(h, stg::GemmArgs{sp.Kmat, P.ldq[k], sp.Rm, ldy, sp.Y, ldy, Res, ldy, q, nn, q, -1.0, 1.0, 0, 0}, KC_ST_GEMM_UPD, allow_sk)) ||
      (e = st_gemm(h, stg::GemmArgs{sp.Kinv, P.ldq[k], Res, ldy, sp.Rm, ldy, sp.Rm, ldy, q, nn, q, 1.0, 1.0, 0, 0}, KC_ST_GEMM_UPD, allow_sk)))
    return e;
  return 0;
}

static int staged_analyze(hqpkkt_t *h, int n, int me, int m, bool dense_dyn = false) {
  if (!h->sd) h->sd = new (std::nothrow) StagedDev;
  if (!h->sd) return HQPKKT_E_MEM;
  StagedDev &d = *h->sd;
  kktdev::StagedPlan &P = d.plan;
  std::vector<int> gnx = P.given_nx, gnu = P.given_nu;
  P = kktdev::StagedPlan();
  P.given_nx = gnx, P.given_nu = gnu;
  P.dense_dyn = dense_dyn;
  if (h->shard_count > 16) return HQPKKT_E_RANGE;
  P.shard_rank = h->shard_rank, P.shard_count = h->shard_count;
  P.sharded = h->shard_count > 1 || h->xchg_fn || h->xchg_sfn;
  h->an.shard_rank = h->shard_rank, h->an.shard_count = 1;  // (the tree engine's exchange plan is not used)
  int e = h->an.setup_blocks(1, n, me, m, h->pQp.data(), h->pQi.data(), h->pAp.data(), h->pAi.data(),
                             h->pCp.data(), h->pCi.data());
  if (e) return e;
  e = P.run(n, me, m, h->pQp.data(), h->pQi.data(), h->pAp.data(), h->pAi.data(), h->pCp.data(), h->pCi.data());
  if (e) return e;
  h->an.sbw = -1;
  h->analyzed = true;
  std::memset(&h->st, 0, sizeof(h->st));
  h->st.dim = n + me, h->st.sbw = -1;
  h->st.n_supernodes = P.K + 1, h->st.n_levels = P.K + 1;
  int mf = 0;
  for (int k = 0; k < P.K; k++) mf = std::max(mf, P.nk[k] + P.mk[k] + P.nk[k + 1]);
  h->st.max_front = mf;
  h->st.nnz_kkt = (long long)P.nq + P.na + P.nc;
  h->st.nnz_factor = P.v_elems + P.misc_elems;
  h->st.flops_factor = P.flops_factor;
  h->st.bytes_panels = (long long)sizeof(double) * (P.f_elems + P.v_elems);
  h->st.bytes_updates = (long long)sizeof(double) * P.misc_elems;
  h->st.shard_rank = P.shard_rank, h->st.shard_count = P.shard_count;
  if (P.sharded) {
    long long bytes = 0, fl = 0;
    const int NR = P.shard_count, RK = P.shard_rank;
    for (int k = 0; k < P.K; k++) {
      bytes += (long long)sizeof(double) * (P.fgslot[k] + P.xslot[k]) * NR;  // (the gathered F: static, requested a stage ahead)
      const int *cut = &P.xcut[(size_t)k * (NR + 1)];
      const long long wd = cut[RK + 1] - cut[RK];
      const long long np = P.nk[k + 1], mm = P.mk[k], nn = P.nk[k], q = P.qmax[k], cx = P.cap[k + 1];
      // own: the strip of W, its columns of the control rows of G and of the carried rows, the tiles of its blocks of
      // G_xx; by every rank: the control columns, the rank-q update of the whole block
      fl += 2 * np * np * wd + 2 * np * 128LL * 128 * (P.gtile_ptr[k + 1] - P.gtile_ptr[k]);
      fl += 2 * np * np * mm + 2 * np * (mm + cx) * (nn + mm) + q * nn * nn;
    }
    h->st.bytes_exchange_factor = bytes, h->st.flops_local = fl, h->st.n_exchange_blocks = 2 * P.K;
    // per solve: a state-sized vector per stage and direction, the partial sums of x+, the dynamics rows' multipliers
    long long sb = 0;
    for (int k = 0; k < P.K; k++) sb += (long long)sizeof(double) * (2LL * P.nk[k + 1] + (long long)NR * P.nk[k + 1]);
    h->st.bytes_exchange_step = sb + (long long)sizeof(double) * P.ndyn;
  }
  return 0;
}

static int staged_upload(hqpkkt_t *h) {
  int e = ensure_device(h);
  if (e) return e;
  Analysis &an = h->an;
  StagedDev &d = *h->sd;
  kktdev::StagedPlan &P = d.plan;
  const int n = an.n, me = an.me, m = an.m;
  if ((e = h->Qf.upload(an.Qfull)) || (e = h->A.upload(an.A)) || (e = h->AT.upload(an.AT)) ||
      (e = h->C.upload(an.C)) || (e = h->CT.upload(an.CT)))
    return e;
  const size_t nv = (size_t)an.nq + an.na + an.nc + 1;
  if ((e = h->vals.alloc(nv)) || (e = h->wt.alloc(m + 1)) || (e = h->flags.alloc(128)) ||
      (e = h->vin.alloc(2 * (size_t)m + n + me + 2 * (size_t)m)) ||
      (e = h->vout.alloc((size_t)n + me + 2 * (size_t)m)) || (e = h->vres.alloc((size_t)n + me + 2 * (size_t)m)) ||
      (e = h->vcor.alloc((size_t)n + me + 2 * (size_t)m)) || (e = h->tz.alloc(m)))
    return e;
  h->bits.p = (unsigned long long *)(h->flags.p + 120);
  if ((e = alloc_hpin(h))) return e;
  if (h->hstage) (void)hipHostFree(h->hstage), h->hstage = nullptr;
  h->hstage_in = h->hstage_out = 0;
  {
    const size_t nin = 4 * (size_t)m + n + me, nout = (size_t)n + me + 2 * (size_t)m;
    if ((nin + nout) * sizeof(double) <= (size_t)512 * 1024 && nin + nout > 0) {
      HIPCHK(hipHostMalloc((void **)&h->hstage, sizeof(double) * (nin + nout), hipHostMallocDefault));
      h->hstage_in = nin, h->hstage_out = nout;
    }
  }
  {
    const double one = 1.0;
    HIPCHK(hipMemcpy(h->vals.p + (nv - 1), &one, sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->wt.p + m, &one, sizeof(double), hipMemcpyHostToDevice));
    const double rows = 2.0 * n + me + m;
    const double nnz = (double)an.Qfull.col.size() + 2.0 * an.A.col.size() + 2.0 * an.C.col.size();
    h->short_rows = rows > 0 && nnz / rows < 8.0;
  }
  // (slack: 16-byte operand loads of column slices may read a tile's width past a block's last row)
  if ((e = d.F.alloc(P.f_elems + 8192)) || (e = d.V.alloc(P.v_elems + 8192)) || (e = d.misc.alloc(P.misc_elems + 8192)) ||
      (e = d.dyn.alloc(P.dyn_ints)) || (e = d.eq_rows.upload(P.eq_rows)) || (e = d.fix_rows.upload(P.fix_rows)) ||
      (e = d.fix_src.upload(P.fix_src)) || (e = d.h_tptr.upload(P.h_tptr)) || (e = d.chk_idx.upload(P.chk_idx)) ||
      (e = d.chk_kind.upload(P.chk_kind)) || (e = d.h_dst.upload(P.h_dst)) || (e = d.a_dst.upload(P.a_dst)))
    return e;
  {
    std::vector<stg::HTerm> t(P.h_terms.size());
    for (size_t k = 0; k < t.size(); k++) t[k] = stg::HTerm{P.h_terms[k].s1, P.h_terms[k].s2, P.h_terms[k].wi};
    if ((e = d.h_terms.upload(t))) return e;
  }
  if (P.dense_dyn && P.sharded) {
    // the local blocks: own state columns [c0, c0 + wd) and the control columns; rank 0 adds what belongs to nobody's strip
    const int NR = P.shard_count, RK = P.shard_rank;
    std::vector<stg::DynLoc> dl(P.K + 1);
    for (int k = 0; k <= P.K; k++) {
      stg::DynLoc &x = dl[k];
      x.oF = k < P.K ? P.oFl[k] : 0, x.ldf = k < P.K ? P.ldfl[k] : 0;
      x.np = k < P.K ? P.nk[k + 1] : 0, x.nz = k < P.K ? P.nk[k] + P.mk[k] : P.nk[k];
      x.col0 = P.nmk[k], x.row0 = k < P.K ? P.nks[k] : P.ndyn, x.ncur = P.nk[k];
      x.c0 = P.xcut[(size_t)k * (NR + 1) + RK], x.wd = P.xcut[(size_t)k * (NR + 1) + RK + 1] - x.c0;
      x.m = k < P.K ? P.mk[k] : 0, x.with_controls = RK == 0;
    }
    d.dyn_sum_x2 = ((long long)n + 15) / 16 * 16;
    if ((e = d.dyn_loc.upload(dl)) || (e = d.dyn_sum.alloc((size_t)(d.dyn_sum_x2 + P.ndyn + 16)))) return e;
  } else if (P.dense_dyn) {
    std::vector<stg::DynDesc> dd(P.K + 1);
    for (int k = 0; k <= P.K; k++) {
      dd[k].oF = k < P.K ? P.oF[k] : 0, dd[k].ldf = k < P.K ? P.ldf[k] : 0;
      dd[k].np = k < P.K ? P.nk[k + 1] : 0, dd[k].nz = k < P.K ? P.nk[k] + P.mk[k] : P.nk[k];
      dd[k].col0 = P.nmk[k], dd[k].row0 = k < P.K ? P.nks[k] : P.ndyn, dd[k].ncur = P.nk[k];
    }
    if ((e = d.dyn_desc.upload(dd)) || (e = d.dyn_x1.alloc(n)) || (e = d.dyn_x2.alloc(P.ndyn))) return e;
    {
      int nzmax = 1;
      for (int k = 0; k < P.K; k++) nzmax = std::max(nzmax, P.nk[k] + P.mk[k]);
      nzmax = std::max(nzmax, P.nk[P.K]);
      d.dyn_part_cols = (nzmax + 255) / 256;
      if ((e = d.dyn_part.alloc((size_t)std::max(P.ndyn, 1) * d.dyn_part_cols))) return e;
    }
  }
  d.gemm_variant = stg::gemm_variant_from_env();
  if (d.gemm_variant != stg::GEMM_REG4) {  // (the register-staged loop stays selectable for comparisons)
    if ((e = d.zeros.alloc(256))) return e;
    HIPCHK(hipMemset(d.zeros.p, 0, sizeof(double) * 256));
  } else
    d.zeros.release();
  HIPCHK(hipMemset(d.F.p, 0, sizeof(double) * std::max<long long>(P.f_elems, 1)));
  HIPCHK(hipMemset(d.V.p, 0, sizeof(double) * std::max<long long>(P.v_elems, 1)));
  HIPCHK(hipMemset(d.misc.p, 0, sizeof(double) * std::max<long long>(P.misc_elems, 1)));
  HIPCHK(hipMemset(d.dyn.p, 0, sizeof(int) * std::max(P.dyn_ints, 1)));
  {  // stream-K grid: two workgroups per CU, if some product of the recursion has more tiles than that
    int cus = 0;
    HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->opts.device));
    d.cus = cus;
    long long tmax = 0, pmax = 0;  // most tiles / most cut pieces of a product of this handle (column slices have fewer)
    for (int k = 0; k < P.K; k++) {
      const long long t1 = (P.nk[k + 1] + 127) / 128, t2 = (P.nk[k] + P.mk[k] + 127) / 128;
      tmax = std::max(tmax, t1 * t2);
      if (cus > 0)
        for (long long t : {t1 * t2, t2 * (t2 + 1) / 2, t1 * (t1 + 1) / 2})
          for (int grid : {2 * cus, cus})
            pmax = std::max(pmax, stg::gemm_split_plan_pieces(stg::gemm_split_plan(t, (std::max(P.nk[k + 1], P.nk[k]) + stg::GEMM_BK - 1) / stg::GEMM_BK, grid)));
    }
    if (P.sharded)
      for (int k = 0; k < P.K; k++) tmax = std::max<long long>(tmax, P.gtile_ptr[k + 1] - P.gtile_ptr[k]);
    // (a plan has at most two cut phases of at most one unit per workgroup of the grid each: gemm_split_plan; every launch
    // checks its pieces against the workspace.  Until round 5 the workspace was sized 16 pieces per tile of the largest
    // product - 3.4 GB at the headline width, per handle, sharded or not)
    pmax = std::max(pmax * 5 / 4 + 64, 4LL * cus + 64);
    d.sk_grid = 0, d.sk_tiles = (int)tmax;
    if (cus > 0) {
      d.sk_grid = stg::gemm_wgs_per_cu(stg::gemm_variant_from_env()) * cus;
      // (the cut form of the 64 x 64 tiles: at most two phases of one unit per workgroup, up to 3/4 of its grid in tiles)
      d.sk_ws_elems = std::max<long long>(pmax, 1) * 128 * 128;
      d.sk_cnt_elems = d.sk_tiles + 4;
      if ((e = d.sk_ws.alloc((size_t)d.sk_ws_elems)) || (e = d.sk_cnt.alloc((size_t)d.sk_cnt_elems))) return e;
      HIPCHK(hipMemset(d.sk_cnt.p, 0, sizeof(unsigned) * (size_t)d.sk_cnt_elems));
    }
  }
  // The control-sized chain of a stage on a second stream beside its large product G_xx.  Measured on one MI355X (same
  // box, tools/c4_bench.py): stages of 1500 / 2000 / 2500 / 3000 states + 2.7 / 2.5 / 3.5 / 2.7 %, 5000 states - 1.1 % (the
  // separate skinny product for the control rows of G and the contention cost more than the hidden chain), 1000 states
  // - 13 %.  So: on for stages of 1280 .. 4096 states.  When sharded the
  // separate product exists anyway (staged_stage_sharded).
  {
    d.overlap_mode = 2;  // by stage width
    bool any = d.overlap_mode == 1 || P.sharded;
    if (d.overlap_mode == 2)
      for (int k = 0; k < P.K; k++) any = any || (P.nk[k] >= 1280 && P.nk[k] <= 4096);
    if (!d.stream2 && any) {
      {
        // (the control-sized chain ahead of the large products' waiting workgroups: a chain of a few small kernels behind
        // a launch that fills every CU otherwise waits a whole tile time for each of its launches)
        int lo = 0, hi = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIPCHK(hipStreamCreateWithPriority(&d.stream2, hipStreamNonBlocking, hi));
      }
      HIPCHK(hipEventCreateWithFlags(&d.ev_fork, hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&d.ev_join, hipEventDisableTiming));
    }
    d.overlap = d.stream2 != nullptr && d.overlap_mode != 0;
    d.ks_ws2_elems = 0;
    if (d.stream2 && d.cus > 0) {
      d.ks_ws2_elems = 8LL << 20;
      if ((e = d.ks_ws2.alloc((size_t)d.ks_ws2_elems))) return e;
    }
  }
  if (P.sharded) {
    const int NR = P.shard_count, RK = P.shard_rank;
    std::vector<stg::StripTab> wt(P.K + 1);
    std::vector<stg::RectTab> rt(P.K + 1);
    std::vector<stg::PackRect> pr;
    d.prect_ptr.assign(P.K + 1, 0);
    for (int k = 0; k < P.K; k++) {
      const int *cut = &P.xcut[(size_t)k * (NR + 1)];
      stg::StripTab &t = wt[k];
      stg::RectTab &r = rt[k];
      t.nranks = r.nranks = NR;
      for (int p = 0; p <= NR; p++) {
        t.cut[p] = r.cut[p] = cut[p];
        if (p < NR) t.off[p] = (long long)p * P.fgslot[k], t.ld[p] = (cut[p + 1] - cut[p] + P.mk[k] + 7) / 8 * 8;
      }
      for (auto &b : r.blk) b.off[0] = b.off[1] = 0, b.rsplit = 1 << 30, b.pad = 0;
      d.prect_ptr[k] = (int)pr.size();
      for (int q = P.xrect_ptr[k]; q < P.xrect_ptr[k + 1]; q++) {
        const kktdev::StagedPlan::XRect &x = P.xrects[q];
        stg::RectTab::Block &b = r.blk[x.a * 16 + x.b];
        const long long off = (long long)x.owner * P.xslot[k] + x.off;
        if (x.r0 == cut[x.a])
          b.off[0] = off;
        else
          b.off[1] = off, b.rsplit = x.r0;
        if (x.owner == RK) {
          stg::PackRect pk{};
          if (x.mine_rows)
            pk.r0 = x.r0, pk.c0 = x.c0, pk.rows = x.r1 - x.r0, pk.cols = x.c1 - x.c0, pk.transpose = 0;
          else  // computed for the partner's rows in the own row strip of G: rows = own columns
            pk.r0 = x.c0, pk.c0 = x.r0, pk.rows = x.c1 - x.c0, pk.cols = x.r1 - x.r0, pk.transpose = 1;
          pk.off = x.off;
          pr.push_back(pk);
        }
      }
    }
    d.prect_ptr[P.K] = (int)pr.size();
    if (pr.empty()) pr.push_back(stg::PackRect{});
    std::vector<int> gt = P.gtile;
    if (gt.empty()) gt.push_back(0);
    // which tiles of the work block G of a stage this rank computes (rows: its strip from c0 on)
    std::vector<unsigned> ow;
    d.gowned_off.assign(P.K + 1, 0);
    for (int k = 0; k < P.K; k++) {
      d.gowned_off[k] = (long long)ow.size();
      const int ntc = (P.nk[k] + 127) / 128, r0 = P.xcut[(size_t)k * (NR + 1) + RK] / 128;
      ow.resize(ow.size() + ((size_t)ntc * ntc + 31) / 32, 0u);
      for (int t = P.gtile_ptr[k]; t < P.gtile_ptr[k + 1]; t++) {
        const int tr = r0 + (P.gtile[t] >> 16), tc = P.gtile[t] & 0xffff;
        if (tr < ntc && tc < ntc) {
          const size_t bit = (size_t)tr * ntc + tc;
          ow[d.gowned_off[k] + bit / 32] |= 1u << (bit % 32);
        }
      }
    }
    d.gowned_off[P.K] = (long long)ow.size();
    if (ow.empty()) ow.push_back(0u);
    if ((e = d.wtabs.upload(wt)) || (e = d.rtabs.upload(rt)) || (e = d.prects.upload(pr)) || (e = d.gtile.upload(gt)) ||
        (e = d.gowned.upload(ow)))
      return e;
    if (!d.ev_x1) HIPCHK(hipEventCreateWithFlags(&d.ev_x1, hipEventDisableTiming));
    if (h->xchg_sfn && !d.stream_x) {
      HIPCHK(hipStreamCreateWithFlags(&d.stream_x, hipStreamNonBlocking));
      for (hipEvent_t *ev : {&d.ev_w[0], &d.ev_w[1], &d.ev_w[2], &d.ev_x[0], &d.ev_x[1], &d.ev_x[2]})
        HIPCHK(hipEventCreateWithFlags(ev, hipEventDisableTiming));
    }
  }
  // orders of the tiles of the triangular products (G, V)
  for (int k = 0; k < P.K; k++)
    for (int sz : {P.nk[k] + P.mk[k], P.nk[k]}) {
      const int T = (sz + 127) / 128;
      if (T >= 16) (void)d.tri_map(T, true);
    }
  // work lists of the cut form of W = V+ F and G = F'W (staged_stage: the products over all columns, over the state
  // columns alone when the control-sized chain runs beside them)
  d.sk_tables_on = stg::gemm_sk_table_from_env();
  if (!P.sharded)
    for (int k = 0; k < P.K; k++) {
      const int np = P.nk[k + 1], nn = P.nk[k], nz = nn + P.mk[k];
      d.sk_tab_prepare(np, nz, np, 0), d.sk_tab_prepare(np, nn, np, 0);
      d.sk_tab_prepare(nz, nz, np, 1), d.sk_tab_prepare(nn, nn, np, 1);
    }
  d.lds_small = 0, d.lds_small_big = 0;
  for (int k = 0; k < P.K; k++) {
    if (P.big[k])
      d.lds_small_big = std::max(d.lds_small_big, stg::st_small_lds(P.mk[k], P.capn[k], true));
    else
      d.lds_small = std::max(d.lds_small, stg::st_small_lds(P.mk[k], P.capn[k]));
  }
  {
    const size_t q = (size_t)P.q0max;
    d.lds_init = (size_t)kktdev::gj_lds_bytes((long long)q) - (P.big0 ? q * (q | 1) * 8 : 0);
    d.lds_x0 = sizeof(double) * (3 * q + 64 * 65 + 8);
  }
  static std::mutex attr_mutex;  // function attributes are process state, shared by all handles - per DEVICE
  struct PerDev { size_t small = 0, small_big = 0, init = 0, init_big = 0, x0 = 0; bool gemm = false; };
  static PerDev per_dev[64];
  if (h->opts.device < 0 || h->opts.device >= 64) return HQPKKT_E_RANGE;
  {
    std::lock_guard<std::mutex> lk(attr_mutex);
    PerDev &pd = per_dev[h->opts.device];
    size_t &attr_small = pd.small, &attr_small_big = pd.small_big, &attr_init = pd.init, &attr_init_big = pd.init_big, &attr_x0 = pd.x0;
    bool &attr_gemm = pd.gemm;
    if (!attr_gemm) {
      HIPCHK(stg::gemm_set_attributes());
      HIPCHK(hipFuncSetAttribute((const void *)stg::k_st_rm, hipFuncAttributeMaxDynamicSharedMemorySize, (int)stg::st_rm_lds(64)));
      attr_gemm = true;
    }
    if (d.lds_small > attr_small) {
      HIPCHK(hipFuncSetAttribute((const void *)stg::k_st_small<256>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)d.lds_small));
      HIPCHK(hipFuncSetAttribute((const void *)stg::k_st_small<1024, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)d.lds_small));
      attr_small = d.lds_small;
    }
    if (d.lds_small_big > attr_small_big) {
      HIPCHK(hipFuncSetAttribute((const void *)stg::k_st_small<1024>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)d.lds_small_big));
      attr_small_big = d.lds_small_big;
    }
    if (!P.big0 && d.lds_init > attr_init) {
      HIPCHK(hipFuncSetAttribute((const void *)stg::k_st_init_factor<256>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)d.lds_init));
      attr_init = d.lds_init;
    }
    if (d.lds_x0 > attr_x0) {
      HIPCHK(hipFuncSetAttribute((const void *)stg::k_st_x0_free, hipFuncAttributeMaxDynamicSharedMemorySize, (int)d.lds_x0));
      attr_x0 = d.lds_x0;
    }
    if (P.big0 && d.lds_init > attr_init_big) {
      HIPCHK(hipFuncSetAttribute((const void *)stg::k_st_init_factor<1024>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)d.lds_init));
      attr_init_big = d.lds_init;
    }
  }
  if ((e = staged_build_symv_tables(d))) return e;
  h->uploaded = true;
  return 0;
}

// the dynamics block of stage k from the caller's F (n+ rows of n_k + m_k values) into the F arena: the whole block,
// or - one system over several ranks - this rank's state columns and the control columns
static int staged_copy_block(hqpkkt_t *h, int k, const double *F, long long ldF, hipMemcpyKind kind) {
  StagedDev &d = *h->sd;
  const kktdev::StagedPlan &P = d.plan;
  const int nn = P.nk[k], mm = P.mk[k], np = P.nk[k + 1];
  if (np <= 0) return 0;
  if (!P.sharded) {
    if (nn + mm > 0)
      HIPCHK(hipMemcpy2DAsync(d.F.p + P.oF[k], sizeof(double) * P.ldf[k], F, sizeof(double) * ldF, sizeof(double) * (nn + mm), np, kind, h->stream));
    return 0;
  }
  const int *cut = &P.xcut[(size_t)k * (P.shard_count + 1)];
  const int c0 = cut[P.shard_rank], wd = cut[P.shard_rank + 1] - c0;
  double *dst = d.F.p + P.oFl[k];
  if (wd > 0) HIPCHK(hipMemcpy2DAsync(dst, sizeof(double) * P.ldfl[k], F + c0, sizeof(double) * ldF, sizeof(double) * wd, np, kind, h->stream));
  if (mm > 0) HIPCHK(hipMemcpy2DAsync(dst + wd, sizeof(double) * P.ldfl[k], F + nn, sizeof(double) * ldF, sizeof(double) * mm, np, kind, h->stream));
  return 0;
}

static int staged_set_values(hqpkkt_t *h, const double *Qx, const double *Ax, const double *Cx,
                             const double *const *Fblk = nullptr, const long long *ldF = nullptr, bool dense = false) {
  Analysis &an = h->an;
  int e;
  if (!h->uploaded && (e = staged_upload(h))) return e;
  StagedDev &d = *h->sd;
  kktdev::StagedPlan &P = d.plan;
  if (P.dense_dyn != (dense || Fblk != nullptr)) return HQPKKT_E_INTERN;  // analysed for the other hand-over
  HIPCHK(hipSetDevice(h->opts.device));
  hipStream_t s = h->stream;
  hipMemcpyKind kind = h->opts.loc == HQPKKT_LOC_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  if (Fblk)
    for (int k = 0; k < P.K; k++) {
      const int nz = P.nk[k] + P.mk[k];
      if (!Fblk[k] || ldF[k] < nz) return HQPKKT_E_SIZES;
      if ((e = staged_copy_block(h, k, Fblk[k], ldF[k], kind))) return e;
    }
  if (an.nq) HIPCHK(hipMemcpyAsync(h->vals.p, Qx, sizeof(double) * an.nq, kind, s));
  if (an.na) HIPCHK(hipMemcpyAsync(h->vals.p + an.nq, Ax, sizeof(double) * an.na, kind, s));
  if (an.nc) HIPCHK(hipMemcpyAsync(h->vals.p + an.nq + an.na, Cx, sizeof(double) * an.nc, kind, s));
  for (CsrBuf *c : {&h->Qf, &h->A, &h->AT, &h->C, &h->CT})
    if (c->src.count)
      k_gather_values<<<nblk((long long)c->src.count), 256, 0, s>>>((int)c->src.count, c->src.p, h->vals.p, c->val.p);
  HIPCHK(hipMemsetAsync(h->flags.p, 0, sizeof(int) * 128, s));
  if (an.na)
    stg::k_st_scatter<<<nblk(an.na), 256, 0, s>>>(an.na, d.a_dst.p, h->vals.p + an.nq, d.F.p, d.misc.p);
  const int nchk = (int)P.chk_idx.size();
  if (nchk) stg::k_st_check<<<nblk(nchk), 256, 0, s>>>(nchk, d.chk_idx.p, d.chk_kind.p, h->vals.p, h->flags.p);
  int *hs = (int *)h->hpin;
  HIPCHK(hipMemcpyAsync(hs, h->flags.p, sizeof(int) * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  if (hs[0]) return HQPKKT_E_FORMAT;  // not the -1.0 staircase (hqp/Hqp_IpLQDOCP.C:214-215) / a zero that fixes x_0
  h->have_values = true;
  h->factored = false;
  return 0;
}

// the tiles g.tile_map[0 .. ntiles) of a product (128 x 128 tiles; the blocks of G_xx one rank owns): their k ranges
// cut when they do not fill the chip (k_dgemm_tn_sk), one plain round otherwise
static int st_gemm_tiles(hqpkkt_t *h, stg::GemmArgs g, int ntiles, int cls) {
  if (ntiles <= 0 || g.M <= 0 || g.N <= 0) return 0;
  StagedDev *d = h->sd;
  const bool al16 = ((((uintptr_t)g.A | (uintptr_t)g.B) & 15) == 0) && ((g.lda & 1) == 0);
  if (d->zeros.p && al16) g.zeros = d->zeros.p;
  const int variant = g.zeros ? d->gemm_variant : stg::GEMM_REG4;
  const long long nslab = (g.K + stg::GEMM_BK - 1) / stg::GEMM_BK;
  if (d->sk_grid > 0 && ntiles % d->sk_grid != 0 && ntiles < 16LL * d->sk_grid && nslab >= 32 && ntiles <= d->sk_tiles) {
    stg::SplitPlan sk = stg::gemm_split_plan(ntiles, nslab, d->sk_grid);
    const StagedDev::SkTab *tab = d->sk_tab(ntiles, nslab, !h->capturing);
    if (tab) sk.table = tab->units->p, sk.stride = tab->stride;
    if (tab || stg::gemm_split_plan_pieces(sk) * 128LL * 128 <= d->sk_ws_elems) {
      sk.ws = d->sk_ws.p, sk.cnt = d->sk_cnt.p;
      KLAUNCH(h, cls, stg::gemm_launch_split(variant, d->sk_grid, h->stream, g, sk));
      return 0;
    }
  }
  KLAUNCH(h, cls, stg::gemm_launch_plain(variant, (unsigned)ntiles, h->stream, g, d->cus));
  return 0;
}

// One stage of the backward recursion when ONE system is sharded over several ranks (DESIGN.md section 7,
// staged_plan.hpp).  Rank p owns the state columns [c0, c1) of the stage: its memory holds those columns of F_k (and the
// control columns), its products are the strip W_p = V+ F_p and the blocks of G_xx the plan gives it - as W_p' F_q, with
// the other rank's F_q out of the GATHERED local blocks, which are static data and were requested a stage ago; W is not
// exchanged.  Everything control-sized is computed by every rank on identical data by launches of identical shape.
//   sA (the handle's): request the gather of stage k - 1's F  ->  W_p = V+ F_p  ->  its blocks of G_xx = W_p' F_q in ONE
//       launch (B from the gathered strips, the tiles of the plan's list), + H_xx  ->  pack (lower orientation)  ->
//       the GATHER OF THE BLOCKS: the one exchange on the critical path
//   sB: W_u = V+ f_u, the control rows of G = W_u' [F_x | F_u] and the carried rows B+ F (thin, deep products over the
//       gathered strips, cut in k), rank decision, K^-1 (k_st_small), Y (k_st_wide), Rm = K^-1 Y - beside the large products
// and, joined: V_k = G_xx - Y' Rm over the WHOLE lower triangle, mirrored, with G_xx read straight from the blocks in
// the exchange buffer (GemmArgs::rects) into the transient full block; the rank keeps its row strip for the solve.
static int exchange(hqpkkt_t *h, int op, double *buf, long long slot, int nslots, hipStream_t on);
// the gather of the ranks' local blocks of stage k into buffer k & 1: stream-ordered transport: on the exchanges' own
// stream behind `after` (the first stream's position when the buffer's last readers are done); otherwise here and now
static int staged_gather_f(hqpkkt_t *h, int k) {
  StagedDev &d = *h->sd;
  kktdev::StagedPlan &P = d.plan;
  const int NR = P.shard_count, RK = P.shard_rank, i = k & 1;
  const int wd = P.xcut[(size_t)k * (NR + 1) + RK + 1] - P.xcut[(size_t)k * (NR + 1) + RK], nloc = wd + P.mk[k], np = P.nk[k + 1];
  double *fg = d.misc.p + P.oFg[i], *mine = fg + (long long)RK * P.fgslot[k];
  hipStream_t sA = h->stream, sX = (h->xchg_sfn && d.stream_x) ? d.stream_x : nullptr;
  hipStream_t on = sX ? sX : sA;
  if (sX) {
    HIPCHK(hipEventRecord(d.ev_w[i], sA));
    HIPCHK(hipStreamWaitEvent(sX, d.ev_w[i], 0));
  }
  if (nloc > 0 && np > 0)
    KLAUNCH(h, KC_ST_VEC, stg::k_st_copy2d<<<std::min(np, 2048), 256, 0, on>>>(stage_ptr(d, k).F, P.ldfl[k], mine, P.ldfl[k], np, nloc));
  if (P.fgslot[k] > 0) {
    const int e = exchange(h, HQPKKT_XCHG_ALLGATHER, fg, P.fgslot[k], NR, sX);
    if (e) return e;
  }
  if (sX) HIPCHK(hipEventRecord(d.ev_x[i], sX));
  return 0;
}
static int staged_stage_sharded(hqpkkt_t *h, int k) {
  StagedDev &d = *h->sd;
  kktdev::StagedPlan &P = d.plan;
  const int NR = P.shard_count, RK = P.shard_rank;
  StagePtr sp = stage_ptr(d, k), sn = stage_ptr(d, k + 1);
  const int nn = P.nk[k], mm = P.mk[k], np = P.nk[k + 1];
  const int ek = P.eq_ptr[k + 1] - P.eq_ptr[k], q = P.qmax[k], cx = P.cap[k + 1];
  const int *cut = &P.xcut[(size_t)k * (NR + 1)];
  const int c0 = cut[RK], c1 = cut[RK + 1], wd = c1 - c0;
  const long long ldfl = P.ldfl[k], ldg = P.ldg[k], ldvn = P.ldv[k + 1], ldy = P.ldy[k], ldv = P.ldv[k], ldwl = P.ldwl[k], ldwu = P.ldwu;
  double *G = d.misc.p + P.oG, *Wl = d.misc.p + P.oWl, *Wu = d.misc.p + P.oWu, *xb = d.misc.p + P.oX;
  double *fg = d.misc.p + P.oFg[k & 1];  // the gathered local blocks of THIS stage (requested a stage ago)
  hipStream_t sA = h->stream, sB = d.stream2 ? d.stream2 : h->stream;
  hipStream_t sX = (h->xchg_sfn && d.stream_x) ? d.stream_x : nullptr;  // the exchanges' own stream (stream-ordered transport)
  struct StreamGuard {  // launches go to h->stream: back to the first stream on every way out
    hqpkkt_t *h;
    hipStream_t s;
    ~StreamGuard() { h->stream = s; }
  } guard{h, sA};
  const bool two = sB != sA;
  int e;
  struct JoinGuard {  // an error between fork and join must not leave the second stream forked
    bool armed = false;
    hipStream_t a, b;
    hipEvent_t ev;
    ~JoinGuard() {
      if (armed && hipEventRecord(ev, b) == hipSuccess) (void)hipStreamWaitEvent(a, ev, 0);
    }
  } join{false, sA, sB, d.ev_join};
  const int ne_x = P.h_mid[k] - P.h_ptr[k], ne_u = P.h_ptr[k + 1] - P.h_mid[k];
  auto add_h = [&](int first, int count) {
    if (count)
      KLAUNCH(h, KC_ASSEMBLE, stg::k_st_add_h<<<nblk(count), 256, 0, h->stream>>>(count, d.h_dst.p + first, d.h_tptr.p + first, d.h_terms.p,
                                                                                 h->vals.p, h->wt.p, G, 1));
  };
  // the next stage's F blocks travel while this stage is computed (its buffer's last readers, the stage before this one,
  // are behind us in the first stream)
  if (k > 0 && (e = staged_gather_f(h, k - 1))) return e;
  if (two) {
    HIPCHK(hipEventRecord(d.ev_fork, sA));
    HIPCHK(hipStreamWaitEvent(sB, d.ev_fork, 0));
    join.armed = true;
  }
  // ---- the control-sized chain, from the gathered F (identical on all ranks: the control columns out of rank 0's slot).
  // Its thin products are launches of hundreds of small workgroups: beside a product that fills every workgroup slot of
  // the chip (the cut form: strips of >= 1024 columns, up to four ranks at the headline width) each of them waits for
  // slots - measured: 1.6 ms for a 20 us kernel, the chain became the critical path - so they go FIRST on the first
  // stream there (0.2 ms), and beside the strip's product only where that one leaves CUs idle (one round of <= 256 tiles).
  const bool thin_first = wd >= 1024;
  h->stream = thin_first ? sA : sB;
  if (sX) HIPCHK(hipStreamWaitEvent(h->stream, d.ev_x[k & 1], 0));
  {
    const stg::StripTab *tab = d.wtabs.p + k;
    const long long ld0 = (cut[1] - cut[0] + mm + 7) / 8 * 8;  // rank 0's local block: [F_0 | F_u]
    const double *Fu = fg + (cut[1] - cut[0]);
    auto thin = [&](const double *A, long long lda, int M, double *C, long long ldc, int cls) -> int {  // C = A' [F_x | F_u]
      int e2;
      stg::GemmArgs gx{A, lda, fg, 0, nullptr, 0, C, ldc, M, nn, np, 1.0, 0.0, 0, 0};
      gx.bstrips = tab;
      if (nn > 0 && (e2 = st_gemm(h, gx, cls, !two || thin_first))) return e2;
      return mm > 0 ? st_gemm(h, stg::GemmArgs{A, lda, Fu, ld0, nullptr, 0, C + nn, ldc, M, mm, np, 1.0, 0.0, 0, 0}, cls, !two || thin_first) : 0;
    };
    if (mm > 0) {
      if ((e = st_gemm(h, stg::GemmArgs{sn.V, ldvn, Fu, ld0, nullptr, 0, Wu, ldwu, np, mm, np, 1.0, 0.0, 0, 0}, KC_ST_GEMM, !two || thin_first))) return e;
      if ((e = thin(Wu, ldwu, mm, G + (long long)nn * ldg, ldg, KC_ST_GEMM))) return e;
    }
    if (cx > 0 && (e = thin(sn.BT, P.ldb[k + 1], cx, sp.N + (size_t)ek * P.ldn[k], P.ldn[k], KC_ST_GEMM_UPD))) return e;
  }
  if (thin_first && two) {  // the rest of the chain beside the large products
    HIPCHK(hipEventRecord(d.ev_x1, sA));
    HIPCHK(hipStreamWaitEvent(sB, d.ev_x1, 0));
  }
  h->stream = sB;
  add_h(P.h_mid[k], ne_u);
  {
    stg::SmallArgs sa{G, ldg, nn, mm, sp.N, P.ldn[k], ek, P.cap[k + 1] > 0 ? sn.dyn + 1 : nullptr,
                      P.capn[k], P.cap[k], q, h->ge_tol, sp.Kinv, P.ldq[k], sp.Kmat, sp.T, P.ldt[k], sp.dyn, h->flags.p,
                      P.big[k] ? d.misc.p + P.oScr : nullptr};
    if (P.big[k]) {
      if ((e = st_small_big(h, d, sa, !two))) return e;
    } else if (P.qmax[k] > 64) {
      static const bool spd_test_fail = getenv("HQPKKT_SPD_TEST_FAIL") != nullptr;
      if (spd_test_fail) sa.mode = 100;
      KLAUNCH(h, KC_ST_SMALL, (stg::k_st_small<1024, false><<<1, 1024, d.lds_small, h->stream>>>(sa)));
    }
    else
      KLAUNCH(h, KC_ST_SMALL, stg::k_st_small<256><<<1, 256, d.lds_small, h->stream>>>(sa));
    stg::WideArgs wa{G, ldg, nn, mm, sp.N, P.ldn[k], P.capn[k], P.cap[k], q, sp.T, P.ldt[k], sp.dyn, sp.Y, ldy, sp.BT, P.ldb[k]};
    if ((e = st_rm(h, d, sp, k, !two, wa))) return e;
  }
  if (two) HIPCHK(hipEventRecord(d.ev_join, sB));
  // ---- sA: the strip of W, the rank's blocks of G_xx (rows = its strip) in one launch, H_xx, pack, the gather of the blocks
  h->stream = sA;
  if (wd > 0 && (e = st_gemm(h, stg::GemmArgs{sn.V, ldvn, sp.F, ldfl, nullptr, 0, Wl, ldwl, np, wd, np, 1.0, 0.0, 0, 0}))) return e;
  if (sX) HIPCHK(hipStreamWaitEvent(sA, d.ev_x[k & 1], 0));
  const int ntile = P.gtile_ptr[k + 1] - P.gtile_ptr[k];
  if (wd > 0 && ntile > 0) {
    stg::GemmArgs gg{Wl, ldwl, fg, 0, nullptr, 0, G + (long long)c0 * ldg, ldg, wd, nn, np, 1.0, 0.0, 0, 0};
    gg.tile_map = d.gtile.p + P.gtile_ptr[k], gg.bstrips = d.wtabs.p + k;
    if ((e = st_gemm_tiles(h, gg, ntile, KC_ST_GEMM))) return e;
  }
  if (ne_x)  // H_xx into the rank's own tiles only (everything else in G is left over from earlier stages and read by nobody)
    KLAUNCH(h, KC_ASSEMBLE, stg::k_st_add_h_owned<<<nblk(ne_x), 256, 0, h->stream>>>(ne_x, d.h_dst.p + P.h_ptr[k], d.h_tptr.p + P.h_ptr[k], d.h_terms.p,
                                                                                  h->vals.p, h->wt.p, G, ldg, d.gowned.p + d.gowned_off[k], (nn + 127) / 128));
  const int npk = d.prect_ptr[k + 1] - d.prect_ptr[k];
  if (npk > 0)
    KLAUNCH(h, KC_ST_VEC, stg::k_st_pack_rects<<<dim3(512, npk), 256, 0, sA>>>(d.prects.p + d.prect_ptr[k], G, ldg, xb + (long long)RK * P.xslot[k]));
  if (P.xslot[k] > 0) {
    if (sX) {
      HIPCHK(hipEventRecord(d.ev_w[2], sA));
      HIPCHK(hipStreamWaitEvent(sX, d.ev_w[2], 0));
    }
    if ((e = exchange(h, HQPKKT_XCHG_ALLGATHER, xb, P.xslot[k], NR, sX))) return e;
    if (sX) {
      HIPCHK(hipEventRecord(d.ev_x[2], sX));
      HIPCHK(hipStreamWaitEvent(sA, d.ev_x[2], 0));
    }
  }
  if (two) {
    HIPCHK(hipStreamWaitEvent(sA, d.ev_join, 0));
    join.armed = false;
  }
  // V_k = G_xx - Y' Rm: lower tiles, mirrored; G_xx from the blocks
  stg::GemmArgs gu{sp.Y, ldy, sp.Rm, ldy, xb, 0, sp.V, ldv, nn, nn, q, -1.0, 1.0, 1, 1};
  gu.rects = d.rtabs.p + k;
  if ((e = st_gemm(h, gu, KC_ST_GEMM_UPD))) return e;  // (q = 0, a stage without controls: V_k = G_xx, the k loop is empty)
  if (wd > 0)
    KLAUNCH(h, KC_ST_VEC, stg::k_st_copy2d<<<std::min(wd, 2048), 256, 0, sA>>>(sp.V + (long long)c0 * ldv, ldv, sp.Vs, ldv, wd, nn));
  return 0;
}

// Hqp_IpLQDOCP::factor (hqp/Hqp_IpLQDOCP.C:796-862): W^-1 Z, C'(W^-1 Z)C, then the backward
// recursion over the stages (ExRiccatiFactorSc, :1794-1999)
static int staged_run_factor(hqpkkt_t *h, const double *z, const double *w) {
  Analysis &an = h->an;
  StagedDev &d = *h->sd;
  kktdev::StagedPlan &P = d.plan;
  hipStream_t s = h->stream;
  const int m = an.m, K = P.K;
  int e;
  {  // the status words and V_K cleared by one kernel (no memset nodes in the captured sequence: kernels.hip.h, k_clear)
    const long long vk = (long long)P.nk[K] * P.ldv[K];
    kktdev::k_clear<<<(int)std::max<long long>(1, std::min<long long>(4096, (vk / 2 + 1023) / 1024)), 256, 0, s>>>(stage_ptr(d, K).V, vk, h->flags.p);
  }
  if (!h->capturing) HIPCHK(hipEventRecord(h->ev0, s));
  if (m > 0) KLAUNCH(h, KC_ASSEMBLE, k_weights<<<nblk(m), 256, 0, s>>>(1, m, an.n + an.me, z, w, h->wt.p, nullptr, h->flags.p));
  if (!h->capturing) HIPCHK(hipEventRecord(h->ev1, s));
  double *G = d.misc.p + P.oG, *W = d.misc.p + P.oW;
  {  // last stage: V_K = H_K, all its equality rows are carried
    StagePtr sp = stage_ptr(d, K);
    const int nK = P.nk[K], eK = P.eq_ptr[K + 1] - P.eq_ptr[K];
    const int ne = P.h_ptr[K + 1] - P.h_ptr[K];
    if (ne)
      KLAUNCH(h, KC_ASSEMBLE, stg::k_st_add_h<<<nblk(ne), 256, 0, s>>>(ne, d.h_dst.p + P.h_ptr[K], d.h_tptr.p + P.h_ptr[K], d.h_terms.p,
                                                                       h->vals.p, h->wt.p, sp.V, 0));
    KLAUNCH(h, KC_ST_SMALL, stg::k_st_last<<<nblk(std::max(nK, 1)), 256, 0, s>>>(nK, eK, P.cap[K], sp.N, P.ldn[K], sp.BT, P.ldb[K], sp.dyn));
    if (P.sharded) {  // this rank's rows of V_K for the solve
      const int c0 = P.xcut[(size_t)K * (P.shard_count + 1) + P.shard_rank], wd = P.xcut[(size_t)K * (P.shard_count + 1) + P.shard_rank + 1] - c0;
      if (wd > 0)
        KLAUNCH(h, KC_ST_VEC, stg::k_st_copy2d<<<std::min(wd, 2048), 256, 0, s>>>(sp.V + (long long)c0 * P.ldv[K], P.ldv[K], sp.Vs, P.ldv[K], wd, nK));
    }
  }
  if (P.sharded && K > 0 && (e = staged_gather_f(h, K - 1))) return e;  // (stage k requests stage k - 1's)
  for (int k = K - 1; k >= 0; k--) {
    if (P.sharded) {
      if ((e = staged_stage_sharded(h, k))) return e;
      continue;
    }
    StagePtr sp = stage_ptr(d, k), sn = stage_ptr(d, k + 1);
    const int nn = P.nk[k], mm = P.mk[k], np = P.nk[k + 1], nz = nn + mm;
    const int ek = P.eq_ptr[k + 1] - P.eq_ptr[k];
    const long long ldf = P.ldf[k], ldg = P.ldg[k], ldvn = P.ldv[k + 1];
    // The control-sized chain of the stage on the second stream, beside the large product G_xx (needs the
    // control columns to start at an even column: 16-byte loads of W + n)
    const bool ovl = d.overlap && mm > 0 && (nn % 2 == 0) && (d.overlap_mode == 1 || (nn >= 1280 && nn <= 4096));
    hipStream_t sA = h->stream, sB = ovl ? d.stream2 : h->stream;
    struct StreamGuard {  // launches go to h->stream: back to the first stream on every way out
      hqpkkt_t *h;
      hipStream_t s;
      ~StreamGuard() { h->stream = s; }
    } guard{h, sA};
    auto on_b = [&]() { h->stream = sB; };
    auto on_a = [&]() { h->stream = sA; };
    const int ne_x = P.h_mid[k] - P.h_ptr[k], ne_u = P.h_ptr[k + 1] - P.h_mid[k];
    auto add_h = [&](int first, int count) {
      if (count)
        KLAUNCH(h, KC_ASSEMBLE, stg::k_st_add_h<<<nblk(count), 256, 0, h->stream>>>(count, d.h_dst.p + first, d.h_tptr.p + first, d.h_terms.p,
                                                                                   h->vals.p, h->wt.p, G, 1));
    };
    // ---- W
    if ((e = st_gemm(h, stg::GemmArgs{sn.V, ldvn, sp.F, ldf, nullptr, 0, W, ldf, np, nz, np, 1.0, 0.0, 0, 0}))) return e;
    if (ovl) {
      HIPCHK(hipEventRecord(d.ev_fork, sA));
      HIPCHK(hipStreamWaitEvent(sB, d.ev_fork, 0));
    }
    // ---- G: the state part (large) on the first stream ...
    if (!ovl) {
      // G = F'W (lower tiles of the whole (n+m) x (n+m) block)
      if ((e = st_gemm(h, stg::GemmArgs{sp.F, ldf, W, ldf, nullptr, 0, G, ldg, nz, nz, np, 1.0, 0.0, 1, 0}))) return e;
      add_h(P.h_ptr[k], ne_x + ne_u);
    } else {
      if ((e = st_gemm(h, stg::GemmArgs{sp.F, ldf, W, ldf, nullptr, 0, G, ldg, nn, nn, np, 1.0, 0.0, 1, 0}))) return e;
      add_h(P.h_ptr[k], ne_x);
      // ... the control rows of G (Gux, Guu) = W_u' F and H's control part on the second
      on_b();
      if (mm > 0 && (e = st_gemm(h, stg::GemmArgs{W + nn, ldf, sp.F, ldf, nullptr, 0, G + nn * ldg, ldg, mm, nz, np, 1.0, 0.0, 0, 0}, KC_ST_GEMM, !ovl)))
        return e;
      add_h(P.h_mid[k], ne_u);
    }
    on_b();
    // carried rows: N_k[e..] = B+ F
    if (P.cap[k + 1] > 0 &&
        (e = st_gemm(h, stg::GemmArgs{sn.BT, P.ldb[k + 1], sp.F, P.ldf[k], nullptr, 0, sp.N + (size_t)ek * P.ldn[k], P.ldn[k],
                                      P.cap[k + 1], nz, np, 1.0, 0.0, 0, 0}, KC_ST_GEMM_UPD, !ovl)))
      return e;
    stg::SmallArgs sa{G, P.ldg[k], nn, mm, sp.N, P.ldn[k], ek, P.cap[k + 1] > 0 ? sn.dyn + 1 : nullptr,
                      P.capn[k], P.cap[k], P.qmax[k], h->ge_tol, sp.Kinv, P.ldq[k], sp.Kmat, sp.T, P.ldt[k], sp.dyn, h->flags.p,
                      P.big[k] ? d.misc.p + P.oScr : nullptr};
    if (P.big[k]) {
      if ((e = st_small_big(h, d, sa, !ovl))) return e;
    } else if (P.qmax[k] > 64) {
      static const bool spd_test_fail = getenv("HQPKKT_SPD_TEST_FAIL") != nullptr;
      if (spd_test_fail) sa.mode = 100;
      KLAUNCH(h, KC_ST_SMALL, (stg::k_st_small<1024, false><<<1, 1024, d.lds_small, h->stream>>>(sa)));
    }
    else
      KLAUNCH(h, KC_ST_SMALL, stg::k_st_small<256><<<1, 256, d.lds_small, h->stream>>>(sa));
    stg::WideArgs wa{G, P.ldg[k], nn, mm, sp.N, P.ldn[k], P.capn[k], P.cap[k], P.qmax[k], sp.T, P.ldt[k], sp.dyn,
                     sp.Y, P.ldy[k], sp.BT, P.ldb[k]};
    if ((e = st_rm(h, d, sp, k, !ovl, wa))) return e;
    on_a();
    if (ovl) {
      HIPCHK(hipEventRecord(d.ev_join, sB));
      HIPCHK(hipStreamWaitEvent(sA, d.ev_join, 0));
    }
    // V = Gxx - Y'Rm (lower tiles, mirrored)
    if ((e = st_gemm(h, stg::GemmArgs{sp.Y, P.ldy[k], sp.Rm, P.ldy[k], G, P.ldg[k], sp.V, P.ldv[k], nn, nn, P.qmax[k], -1.0, 1.0, 1, 1},
                     KC_ST_GEMM_UPD)))
      return e;
  }
  {
    StagePtr s0 = stage_ptr(d, 0);
    if (P.fixed_x0)
      KLAUNCH(h, KC_ST_SMALL, stg::k_st_check_fixed<<<1, 64, 0, s>>>(s0.dyn, h->flags.p));
    else if (P.big0) {
      // the inverse by the blocked sweep on the whole chip, checked against K0; the LU factorisation by one workgroup
      // behind it runs only where the sweep gave up (decided on the device)
      const bool legacy0 = false;
      static const double tol0 = getenv("HQPKKT_BLOCK_GJ_TOL") ? atof(getenv("HQPKKT_BLOCK_GJ_TOL")) : 1e-6;
      double *scr = d.misc.p + P.oScr;
      const int q = P.q0max;
      if (!legacy0) {
        const stg::X0Args xa{P.nk[0], q, s0.V, P.ldv[0], s0.BT, P.ldb[0], s0.dyn, d.misc.p + P.oK0, d.misc.p + P.oK0m, d.misc.p + P.oK0s,
                             P.ldq0, scr, h->flags.p};
        KLAUNCH(h, KC_ST_SMALL, stg::k_x0_prepare<<<nblk((long long)q * q), 256, 0, s>>>(xa));
        if ((e = st_blk_sweep(h, scr, q, true))) return e;
        KLAUNCH(h, KC_ST_SMALL, stg::k_x0_final<<<nblk((long long)q * q), 256, 0, s>>>(xa));
        const stg::BigScratch bs = stg::big_scratch(scr, q);
        if ((e = st_gemm(h, stg::GemmArgs{xa.K0mat, P.ldq0, xa.K0inv, P.ldq0, nullptr, 0, bs.Ks, bs.ldk, q, q, q, 1.0, 0.0, 0, 0}, KC_ST_GEMM_UPD, true)))
          return e;
        KLAUNCH(h, KC_ST_SMALL, stg::k_x0_check<<<1, 1024, 0, s>>>(xa, tol0));
      }
      KLAUNCH(h, KC_ST_SMALL, stg::k_st_init_factor<1024><<<1, 1024, d.lds_init, s>>>(P.nk[0], P.cap[0], s0.V, P.ldv[0], s0.BT, P.ldb[0], s0.dyn,
                                                                                    d.misc.p + P.oK0, d.misc.p + P.oK0m, d.misc.p + P.oK0s, P.ldq0, P.q0max, h->flags.p,
                                                                                    scr, legacy0 ? nullptr : stg::big_scratch(scr, q).flags));
    } else
      KLAUNCH(h, KC_ST_SMALL, stg::k_st_init_factor<256><<<1, 256, d.lds_init, s>>>(P.nk[0], P.cap[0], s0.V, P.ldv[0], s0.BT, P.ldb[0], s0.dyn,
                                                                                  d.misc.p + P.oK0, d.misc.p + P.oK0m, d.misc.p + P.oK0s, P.ldq0, P.q0max, h->flags.p, nullptr, nullptr));
  }
  if (!h->capturing) HIPCHK(hipEventRecord(h->evs1, s));
  HIPCHK(hipGetLastError());
  return 0;
}

// The same sweeps when ONE system is sharded over several ranks (staged_plan.hpp): the products with V_k run on the
// rank's ROW strip, those with F_k on its COLUMN strip (and the control columns); everything control-sized is computed
// by every rank.  Per stage one gather of a state-sized vector in the backward sweep (tt = v+ + V+ f, by rows) and one
// of the ranks' partial sums in the forward sweep (x+ = F s + f, by columns); the multipliers of the dynamics rows
// (V+ x+ + v+ + B+' eta+, by rows) are summed over the ranks once, at the end.  Not captured: the exchanges are calls.
static int staged_run_step_sharded(hqpkkt_t *h, const Vecs &v) {
  Analysis &an = h->an;
  StagedDev &d = *h->sd;
  kktdev::StagedPlan &P = d.plan;
  hipStream_t s = h->stream;
  const int n = an.n, m = an.m, K = P.K, NR = P.shard_count, RK = P.shard_rank;
  double *M = d.misc.p;
  double *S = M + P.oS, *qv = M + P.oQv, *gam = M + P.oGam, *tt = M + P.oTT, *xv = M + P.oXV, *xp = M + P.oXP, *dyx = M + P.oDyx;
  auto cut0 = [&](int k) { return P.xcut[(size_t)k * (NR + 1) + RK]; };
  auto width = [&](int k) { return P.xcut[(size_t)k * (NR + 1) + RK + 1] - P.xcut[(size_t)k * (NR + 1) + RK]; };
  int e;
  const long long ndx = (long long)P.ndyn + (P.fixed_x0 ? P.nk[0] : 0);
  KLAUNCH(h, KC_ST_VEC, stg::k_st_zero<<<nblk(std::max<long long>(ndx, 1)), 256, 0, s>>>(ndx, dyx));
  if (m > 0) KLAUNCH(h, KC_VECTOR, k_red_t<<<nblk(m), 256, 0, s>>>(m, v.w, h->wt.p, v.r3, v.r4, h->tz.p));
  KLAUNCH(h, KC_VECTOR, stg::k_st_q<<<nblk(n), 256, 0, s>>>(n, h->CT.ptr.p, h->CT.col.p, h->CT.src.p, h->vals.p, h->tz.p, v.r1, qv));
  {  // last stage
    StagePtr sp = stage_ptr(d, K);
    const int nK = P.nk[K], eK = P.eq_ptr[K + 1] - P.eq_ptr[K];
    KLAUNCH(h, KC_ST_VEC, stg::k_st_copy<<<nblk(nK), 256, 0, s>>>(nK, qv + P.nmk[K], sp.v));
    if (eK) KLAUNCH(h, KC_ST_VEC, stg::k_st_gather<<<nblk(eK), 256, 0, s>>>(eK, d.eq_rows.p + P.eq_ptr[K], v.r2, sp.beta));
  }
  for (int k = K - 1; k >= 0; k--) {
    StagePtr sp = stage_ptr(d, k), sn = stage_ptr(d, k + 1);
    const int nn = P.nk[k], mm = P.mk[k], np = P.nk[k + 1];
    const int c0 = cut0(k), wd = width(k), c0n = cut0(k + 1), wdn = width(k + 1);
    const double *f = v.r2 + P.nks[k];
    // tt = v+ + V+ f by rows: the ranks' strips side by side (strip p starts at p xw), gathered
    if (wdn > 0 && (e = st_gemv_rows(h, stg::GemvRows{sn.Vs, P.ldv[k + 1], wdn, np, f, sn.v + c0n, nullptr, 0, nullptr, nullptr,
                                                        xv + (long long)RK * P.xw[k + 1], 1.0})))
      return e;
    if ((e = exchange(h, HQPKKT_XCHG_ALLGATHER, xv, P.xw[k + 1], NR, nullptr))) return e;
    // gam = q_k + F' tt: the own state columns and the control columns
    if (wd > 0 && (e = st_gemv_cols(h, d, sp.F, P.ldfl[k], np, wd, xv, qv + P.nmk[k] + c0, 1.0, gam + c0))) return e;
    if (mm > 0 && (e = st_gemv_cols(h, d, sp.F + wd, P.ldfl[k], np, mm, xv, qv + P.nmk[k] + nn, 1.0, gam + nn))) return e;
    stg::BwdSmall ba{nn, mm, np, P.eq_ptr[k + 1] - P.eq_ptr[k], P.capn[k], P.cap[k], P.qmax[k], d.eq_rows.p + P.eq_ptr[k], v.r2,
                     P.cap[k + 1] > 0 ? sn.dyn + 1 : nullptr, sn.beta, sn.BT, P.ldb[k + 1], f, gam, sp.Kinv, sp.Kmat, P.ldq[k], sp.T, P.ldt[k],
                     sp.dyn, sp.rho, sp.beta};
    KLAUNCH(h, KC_ST_SMALL, stg::k_st_bwd_small<<<1, 256, sizeof(double) * (P.capn[k] + 3 * P.qmax[k] + 4 + 256), s>>>(ba));
    // v_k = gam_x - Y' rho: the own entries (all a later product needs)
    if (wd > 0 && (e = st_gemv_cols(h, d, sp.Y + c0, P.ldy[k], P.qmax[k], wd, sp.rho, gam + c0, -1.0, sp.v + c0))) return e;
  }
  {
    StagePtr s0 = stage_ptr(d, 0);
    const int n0 = P.nk[0];
    if (P.fixed_x0)
      KLAUNCH(h, KC_ST_VEC, stg::k_st_x0_fixed<<<nblk(std::max(n0, P.cap[0])), 256, 0, s>>>(n0, d.fix_rows.p, d.fix_src.p, h->vals.p, v.r2, S,
                                                                                         s0.eta, P.cap[0]));
    else {
      // the free initial state needs v_0 in full: gathered
      const int c0 = cut0(0), wd = width(0);
      if (wd > 0) KLAUNCH(h, KC_ST_VEC, stg::k_st_copy<<<nblk(wd), 256, 0, s>>>(wd, s0.v + c0, xv + (long long)RK * P.xw[0]));
      if ((e = exchange(h, HQPKKT_XCHG_ALLGATHER, xv, P.xw[0], NR, nullptr))) return e;
      KLAUNCH(h, KC_ST_VEC, stg::k_st_copy<<<nblk(n0), 256, 0, s>>>(n0, xv, s0.v));
      if (P.big0) {
        const int q = P.q0max, l8 = (q + 7) / 8 * 8;
        double *vec = M + P.oK0s + 3 * (long long)q + 8, *nb = vec, *pb = vec + l8, *y = vec + 2 * l8, *r = vec + 3 * l8;
        const stg::X0Vec xv0{n0, P.cap[0], q, s0.dyn, M + P.oK0s, s0.v, s0.beta, nb, pb, pb, S, s0.eta};
        KLAUNCH(h, KC_ST_VEC, stg::k_x0_rhs<<<nblk(q), 256, 0, s>>>(xv0));
        if ((e = st_gemv_rows(h, stg::GemvRows{M + P.oK0, P.ldq0, q, q, nb, nullptr, nullptr, 0, nullptr, nullptr, y, 1.0})) ||
            (e = st_gemv_rows(h, stg::GemvRows{M + P.oK0m, P.ldq0, q, q, y, pb, nullptr, 0, nullptr, nullptr, r, -1.0})) ||
            (e = st_gemv_rows(h, stg::GemvRows{M + P.oK0, P.ldq0, q, q, r, y, nullptr, 0, nullptr, nullptr, pb, 1.0})))
          return e;
        KLAUNCH(h, KC_ST_VEC, stg::k_x0_out<<<nblk(n0 + P.cap[0]), 256, 0, s>>>(xv0));
      }
      KLAUNCH(h, KC_ST_SMALL, stg::k_st_x0_free<<<1, 256, d.lds_x0, s>>>(n0, P.cap[0], P.q0max, M + P.oK0, M + P.oK0m, M + P.oK0s, P.ldq0, s0.dyn, s0.v,
                                                                s0.beta, S, s0.eta));
    }
  }
  for (int k = 0; k < K; k++) {
    StagePtr sp = stage_ptr(d, k), sn = stage_ptr(d, k + 1);
    const int nn = P.nk[k], mm = P.mk[k], np = P.nk[k + 1];
    const int c0 = cut0(k), wd = width(k), c0n = cut0(k + 1), wdn = width(k + 1);
    double *xk = S + P.nmk[k];
    // [u ; yhat] = -(Rm x + rho)
    double *uy = M + P.oUy;
    if (P.qmax[k] > 0) {
      stg::GemvRows gr{sp.Rm, P.ldy[k], P.qmax[k], nn, xk, sp.rho, nullptr, 0, nullptr, nullptr, uy, -1.0};
      KLAUNCH(h, KC_ST_VEC, stg::k_st_gemv_wide<<<P.qmax[k], 256, 0, s>>>(gr));
    }
    stg::FwdSmall fa{nn, mm, P.eq_ptr[k + 1] - P.eq_ptr[k], P.capn[k], P.cap[k], P.qmax[k], uy, sp.T, P.ldt[k],
                     sp.dyn, sp.eta, d.eq_rows.p + P.eq_ptr[k], xk + nn, v.dy, sn.eta, P.cap[k + 1]};
    KLAUNCH(h, KC_ST_SMALL, stg::k_st_fwd_small<<<1, 256, sizeof(double) * (P.capn[k] + 4), s>>>(fa));
    // x+ = F s + f: the own columns' share of every row, gathered, and added in the order of the ranks to f_u u + f
    if ((e = st_gemv_rows(h, stg::GemvRows{sp.F, P.ldfl[k], np, wd, xk + c0, nullptr, nullptr, 0, nullptr, nullptr, xp + (long long)RK * P.xpslot, 1.0})))
      return e;
    if ((e = exchange(h, HQPKKT_XCHG_ALLGATHER, xp, P.xpslot, NR, nullptr))) return e;
    if ((e = st_gemv_rows(h, stg::GemvRows{sp.F + wd, P.ldfl[k], np, mm, xk + nn, v.r2 + P.nks[k], nullptr, 0, nullptr, nullptr, tt, 1.0}))) return e;
    KLAUNCH(h, KC_ST_VEC, stg::k_st_sum_slots<<<nblk(np), 256, 0, s>>>(np, NR, xp, P.xpslot, tt, S + P.nmk[k + 1]));
    // p = V+ x+ + v+ + B+' eta+: the own rows
    if (wdn > 0 &&
        (e = st_gemv_rows(h, stg::GemvRows{sn.Vs, P.ldv[k + 1], wdn, np, S + P.nmk[k + 1], sn.v + c0n,
                                           P.cap[k + 1] > 0 ? sn.BT + (long long)c0n * P.ldb[k + 1] : nullptr, P.ldb[k + 1], sn.dyn + 1, sn.eta,
                                           dyx + P.nks[k] + c0n, 1.0})))
      return e;
  }
  {
    StagePtr sK = stage_ptr(d, K), s0 = stage_ptr(d, 0);
    const int eK = P.eq_ptr[K + 1] - P.eq_ptr[K], n0 = P.nk[0];
    if (eK) KLAUNCH(h, KC_ST_VEC, stg::k_st_y_last<<<nblk(eK), 256, 0, s>>>(eK, d.eq_rows.p + P.eq_ptr[K], sK.eta, v.dy));
    if (P.fixed_x0 && width(0) > 0 &&
        (e = st_gemv_rows(h, stg::GemvRows{s0.Vs, P.ldv[0], width(0), n0, S, s0.v + cut0(0), nullptr, 0, nullptr, nullptr, dyx + P.ndyn + cut0(0), 1.0})))
      return e;
    if (ndx > 0 && (e = exchange(h, HQPKKT_XCHG_ALLREDUCE_SUM, dyx, ndx, 1, nullptr))) return e;
    if (P.ndyn > 0) KLAUNCH(h, KC_ST_VEC, stg::k_st_copy<<<nblk(P.ndyn), 256, 0, s>>>(P.ndyn, dyx, v.dy));
    if (P.fixed_x0) KLAUNCH(h, KC_ST_VEC, stg::k_st_y_fixed<<<nblk(n0), 256, 0, s>>>(n0, d.fix_rows.p, d.fix_src.p, h->vals.p, dyx + P.ndyn, v.dy));
  }
  KLAUNCH(h, KC_VECTOR, stg::k_st_negate<<<nblk(n), 256, 0, s>>>(n, S, v.dx));
  if (m > 0)
    KLAUNCH(h, KC_VECTOR, k_red_dzdw<<<nblk(m), 256, 0, s>>>(m, h->C.ptr.p, h->C.col.p, h->C.src.p, h->vals.p, v.dx, h->wt.p, h->tz.p,
                                                             v.r3, v.dz, v.dw));
  HIPCHK(hipGetLastError());
  return 0;
}

// Hqp_IpLQDOCP::step (hqp/Hqp_IpLQDOCP.C:869-976) with ExRiccatiSolveSc (:2007-2182)
static int staged_run_step(hqpkkt_t *h, const Vecs &v) {
  if (h->sd->plan.sharded) return staged_run_step_sharded(h, v);
  Analysis &an = h->an;
  StagedDev &d = *h->sd;
  kktdev::StagedPlan &P = d.plan;
  hipStream_t s = h->stream;
  const int n = an.n, m = an.m, K = P.K;
  double *M = d.misc.p;
  double *S = M + P.oS, *qv = M + P.oQv, *gam = M + P.oGam, *tt = M + P.oTT, *tmp = M + P.oTmp;
  int e;
  if (m > 0) KLAUNCH(h, KC_VECTOR, k_red_t<<<nblk(m), 256, 0, s>>>(m, v.w, h->wt.p, v.r3, v.r4, h->tz.p));
  KLAUNCH(h, KC_VECTOR, stg::k_st_q<<<nblk(n), 256, 0, s>>>(n, h->CT.ptr.p, h->CT.col.p, h->CT.src.p, h->vals.p, h->tz.p, v.r1, qv));
  // The products with V are not part of the sweeps' chains: V+ f (f: the dynamics' right-hand side) is known before the
  // backward sweep starts, the dynamics rows' multipliers are wanted by nobody before the forward sweep is over - both
  // for many stages per launch (staged_symv_group), which leaves the F products and the control-sized kernels in the
  // chains.
  double *gv = M + P.oGv;
  // (on a stream of their own beside the chains - lowest priority, or a few workgroups that take the tiles in a stride -
  // the launches gained nothing: 35.8 - 37.4 ms per solve against 35.5; what the chains leave idle of HBM they lose again
  // when they share it)
  for (int gi = (int)d.symv_groups[0].size() - 1; gi >= 0; gi--)
    if ((e = staged_symv_group(h, d, 0, gi, v.r2, nullptr))) return e;
  if ((e = staged_symv_rows(h, d, 0, v.r2, nullptr))) return e;
  {  // last stage: v_K, and tt = v_K + V_K f_{K-1} for the stage before
    StagePtr sp = stage_ptr(d, K);
    const int nK = P.nk[K], eK = P.eq_ptr[K + 1] - P.eq_ptr[K];
    if (K > 0)
      KLAUNCH(h, KC_ST_VEC, stg::k_st_copy_add<<<nblk(nK), 256, 0, s>>>(nK, qv + P.nmk[K], sp.v, gv + P.nks[K - 1], tt));
    else
      KLAUNCH(h, KC_ST_VEC, stg::k_st_copy<<<nblk(nK), 256, 0, s>>>(nK, qv + P.nmk[K], sp.v));
    if (eK) KLAUNCH(h, KC_ST_VEC, stg::k_st_gather<<<nblk(eK), 256, 0, s>>>(eK, d.eq_rows.p + P.eq_ptr[K], v.r2, sp.beta));
  }
  for (int k = K - 1; k >= 0; k--) {
    StagePtr sp = stage_ptr(d, k), sn = stage_ptr(d, k + 1);
    const int nn = P.nk[k], mm = P.mk[k], np = P.nk[k + 1], nz = nn + mm;
    const double *f = v.r2 + P.nks[k];
    // gam = q_k + F' tt with tt = v+ + V+ f (from the stage behind)
    if ((e = st_gemv_cols(h, d, sp.F, P.ldf[k], np, nz, tt, qv + P.nmk[k], 1.0, gam))) return e;
    stg::BwdSmall ba{nn, mm, np, P.eq_ptr[k + 1] - P.eq_ptr[k], P.capn[k], P.cap[k], P.qmax[k], d.eq_rows.p + P.eq_ptr[k], v.r2,
                     P.cap[k + 1] > 0 ? sn.dyn + 1 : nullptr, sn.beta, sn.BT, P.ldb[k + 1], f, gam, sp.Kinv, sp.Kmat, P.ldq[k], sp.T, P.ldt[k],
                     sp.dyn, sp.rho, sp.beta};
    KLAUNCH(h, KC_ST_SMALL, stg::k_st_bwd_small<<<1, 256, sizeof(double) * (P.capn[k] + 3 * P.qmax[k] + 4 + 256), s>>>(ba));
    // v_k = gam_x - Y' rho (and tt = v_k + V_k f_{k-1} for the next stage of the sweep)
    if ((e = st_gemv_cols(h, d, sp.Y, P.ldy[k], P.qmax[k], nn, sp.rho, gam, -1.0, sp.v, k > 0 ? gv + P.nks[k - 1] : nullptr, k > 0 ? tt : nullptr)))
      return e;
  }
  {
    StagePtr s0 = stage_ptr(d, 0);
    const int n0 = P.nk[0];
    if (P.fixed_x0)
      KLAUNCH(h, KC_ST_VEC, stg::k_st_x0_fixed<<<nblk(std::max(n0, P.cap[0])), 256, 0, s>>>(n0, d.fix_rows.p, d.fix_src.p, h->vals.p, v.r2, S,
                                                                                         s0.eta, P.cap[0]));
    else {
      if (P.big0) {
        // with the inverse of the blocked sweep (K0s[3 q]: which form the area holds; decided on the device): three
        // products over the whole chip; k_st_x0_free behind them works only where the factors are in use
        const int q = P.q0max, l8 = (q + 7) / 8 * 8;
        double *vec = M + P.oK0s + 3 * (long long)q + 8, *nb = vec, *pb = vec + l8, *y = vec + 2 * l8, *r = vec + 3 * l8;
        const stg::X0Vec xv{n0, P.cap[0], q, s0.dyn, M + P.oK0s, s0.v, s0.beta, nb, pb, pb, S, s0.eta};
        KLAUNCH(h, KC_ST_VEC, stg::k_x0_rhs<<<nblk(q), 256, 0, s>>>(xv));
        if ((e = st_gemv_rows(h, stg::GemvRows{M + P.oK0, P.ldq0, q, q, nb, nullptr, nullptr, 0, nullptr, nullptr, y, 1.0})) ||
            (e = st_gemv_rows(h, stg::GemvRows{M + P.oK0m, P.ldq0, q, q, y, pb, nullptr, 0, nullptr, nullptr, r, -1.0})) ||
            (e = st_gemv_rows(h, stg::GemvRows{M + P.oK0, P.ldq0, q, q, r, y, nullptr, 0, nullptr, nullptr, pb, 1.0})))
          return e;
        KLAUNCH(h, KC_ST_VEC, stg::k_x0_out<<<nblk(n0 + P.cap[0]), 256, 0, s>>>(xv));
      }
      KLAUNCH(h, KC_ST_SMALL, stg::k_st_x0_free<<<1, 256, d.lds_x0, s>>>(n0, P.cap[0], P.q0max, M + P.oK0, M + P.oK0m, M + P.oK0s, P.ldq0, s0.dyn, s0.v,
                                                                s0.beta, S, s0.eta));
    }
  }
  for (int k = 0; k < K; k++) {
    StagePtr sp = stage_ptr(d, k), sn = stage_ptr(d, k + 1);
    const int nn = P.nk[k], mm = P.mk[k], np = P.nk[k + 1], nz = nn + mm;
    double *xk = S + P.nmk[k];
    // [u ; yhat] = -(Rm x + rho)
    double *uy = M + P.oUy;
    if (P.qmax[k] > 0) {
      stg::GemvRows gr{sp.Rm, P.ldy[k], P.qmax[k], nn, xk, sp.rho, nullptr, 0, nullptr, nullptr, uy, -1.0};
      KLAUNCH(h, KC_ST_VEC, stg::k_st_gemv_wide<<<P.qmax[k], 256, 0, s>>>(gr));
    }
    stg::FwdSmall fa{nn, mm, P.eq_ptr[k + 1] - P.eq_ptr[k], P.capn[k], P.cap[k], P.qmax[k], uy, sp.T, P.ldt[k],
                     sp.dyn, sp.eta, d.eq_rows.p + P.eq_ptr[k], xk + nn, v.dy, sn.eta, P.cap[k + 1]};
    KLAUNCH(h, KC_ST_SMALL, stg::k_st_fwd_small<<<1, 256, sizeof(double) * (P.capn[k] + 4), s>>>(fa));
    // x+ = F s + f (the multipliers p = V+ x+ + v+ + B+' eta+ behind the sweep)
    if ((e = st_gemv_rows(h, stg::GemvRows{sp.F, P.ldf[k], np, nz, xk, v.r2 + P.nks[k], nullptr, 0, nullptr, nullptr, S + P.nmk[k + 1], 1.0})))
      return e;
  }
  for (int gi = 0; gi < (int)d.symv_groups[1].size(); gi++)
    if ((e = staged_symv_group(h, d, 1, gi, nullptr, v.dy))) return e;
  if ((e = staged_symv_rows(h, d, 1, nullptr, v.dy))) return e;
  {
    StagePtr sK = stage_ptr(d, K), s0 = stage_ptr(d, 0);
    const int eK = P.eq_ptr[K + 1] - P.eq_ptr[K];
    if (eK) KLAUNCH(h, KC_ST_VEC, stg::k_st_y_last<<<nblk(eK), 256, 0, s>>>(eK, d.eq_rows.p + P.eq_ptr[K], sK.eta, v.dy));
    if (P.fixed_x0) {
      const int n0 = P.nk[0];
      if ((e = st_symv(h, d, stg::GemvRows{s0.V, P.ldv[0], n0, n0, S, s0.v, nullptr, 0, nullptr, nullptr, tmp, 1.0}))) return e;
      KLAUNCH(h, KC_ST_VEC, stg::k_st_y_fixed<<<nblk(n0), 256, 0, s>>>(n0, d.fix_rows.p, d.fix_src.p, h->vals.p, tmp, v.dy));
    }
  }
  KLAUNCH(h, KC_VECTOR, stg::k_st_negate<<<nblk(n), 256, 0, s>>>(n, S, v.dx));
  if (m > 0)
    KLAUNCH(h, KC_VECTOR, k_red_dzdw<<<nblk(m), 256, 0, s>>>(m, h->C.ptr.p, h->C.col.p, h->C.src.p, h->vals.p, v.dx, h->wt.p, h->tz.p,
                                                             v.r3, v.dz, v.dw));
  HIPCHK(hipGetLastError());
  return 0;
}
