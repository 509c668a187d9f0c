// Symbolic phase of the STAGED engine (see staged_plan.hpp).
#include "staged_plan.hpp"

#include <algorithm>
#include <cstdint>
#include <numeric>

namespace kktdev {

namespace {
const int CARRY_MAX = 256;   // carried constraint rows a stage may hand to the one before it
const int CONTROLS_MAX = 512;  // controls per stage
inline int up8(long long x) { return (int)((x + 7) / 8 * 8); }
inline long long up16(long long x) { return (x + 15) / 16 * 16; }
}  // namespace

// Stage sizes of a DOCP from the staircase its dynamics rows form in A: what Hqp_IpLQDOCP::Get_Dim reads off
// the same rows (hqp/Hqp_IpLQDOCP.C:201-287).  Per row only its length, the column of its last entry (the -1.0
// that multiplies a component of x_{k+1}) and of the entry before it are looked at, so a caller that keeps A
// as row lists (the reference-side binding: shim/) hands over three ints per row and never a CSR copy of the
// dynamics.  The walk: every dynamics row ends one column further right than the row before it while it
// belongs to the same stage; a stage ends where that column jumps (the controls of the next stage lie
// between) or where a row's other entries no longer reach back behind the states this stage has produced
// so far (stages without controls).  The walk is over when a row ends in the last column of A.
// Returns 0 and states[K+1], controls[K], first_col[K+1], dyn_rows, or 6 (HQPKKT_E_FORMAT).
int stages_from_staircase(int n, int rows, const int *row_len, const int *last_col, const int *prev_col,
                          std::vector<int> &states, std::vector<int> &controls, std::vector<int> &first_col, int &dyn_rows) {
  struct Stage {
    int first_target;  // column of the first state this stage produces (= first column of the NEXT stage's block)
    int produced;      // dynamics rows of this stage = states of the next one
  };
  std::vector<Stage> found;
  int target = -1;  // column the previous row ended in
  dyn_rows = -1;
  for (int i = 0; i < rows && dyn_rows < 0; i++) {
    if (row_len[i] < 2) return 6;
    const int end = last_col[i], reach = prev_col[i];
    if (end <= target || end >= n) return 6;
    const bool contiguous = end == target + 1;
    const bool same_stage = !found.empty() && contiguous && end - reach >= found.back().produced;
    if (same_stage)
      found.back().produced++;
    else
      found.push_back(Stage{end, 1});
    target = end;
    if (end == n - 1) dyn_rows = i + 1;
  }
  if (dyn_rows < 0 || found.empty()) return 6;
  const int K = (int)found.size();
  states.assign(K + 1, 0), controls.assign(K, 0), first_col.assign(K + 1, 0);
  for (int k = 0; k < K; k++) states[k + 1] = found[k].produced, first_col[k + 1] = found[k].first_target;
  // x_0: as many states as the first stage produces, unless its block is narrower (hqp/Hqp_IpLQDOCP.C:265)
  states[0] = std::min(states[1], first_col[1]);
  for (int k = 0; k < K; k++) {
    controls[k] = first_col[k + 1] - first_col[k] - states[k];
    if (controls[k] < 0) return 6;
  }
  return 0;
}

int StagedPlan::run(int n_, int me_, int m_, const int *Qp, const int *Qi, const int *Ap, const int *Ai,
                    const int *Cp, const int *Ci) {
  n = n_, m = m_;
  nq = n ? Qp[n] : 0, nc = m ? Cp[m] : 0;
  const int arows = me_;  // rows of the A that was handed over
  na = arows ? Ap[arows] : 0;
  nk.clear(), mk.clear(), nmk.clear(), nks.clear();

  // ------------------------------------------------------------------ stage sizes
  if (!given_nx.empty()) {
    K = (int)given_nx.size() - 1;
    if (K < 1 || (int)given_nu.size() != K) return 6;
    nk = given_nx;
    mk = given_nu;
    nmk.assign(K + 1, 0);
    for (int k = 0; k < K; k++) nmk[k + 1] = nmk[k] + nk[k] + mk[k];
    if (nmk[K] + nk[K] != n) return 6;
    nks.assign(K + 1, 0);
    for (int k = 0; k < K; k++) nks[k + 1] = nks[k] + nk[k + 1];
    ndyn = nks[K];
    if (arows < ndyn) return 6;
    if (!dense_dyn) {
      // the rows must be the staircase these sizes describe
      for (int k = 0; k < K; k++)
        for (int i = nks[k]; i < nks[k + 1]; i++) {
          if (Ap[i + 1] - Ap[i] < 1 || Ai[Ap[i + 1] - 1] != nmk[k + 1] + (i - nks[k])) return 6;
        }
    } else if (Ap[ndyn] != 0)
      return 6;  // dense dynamics: their rows of A are empty
  } else {
    // the -1.0 staircase (values are checked when they arrive: chk_idx)
    if (dense_dyn || arows == 0) return 6;
    std::vector<int> len(arows), tail(arows), before(arows);
    for (int i = 0; i < arows; i++) {
      len[i] = Ap[i + 1] - Ap[i];
      tail[i] = len[i] > 0 ? Ai[Ap[i + 1] - 1] : -1;
      before[i] = len[i] > 1 ? Ai[Ap[i + 1] - 2] : -1;
    }
    if (stages_from_staircase(n, arows, len.data(), tail.data(), before.data(), nk, mk, nmk, ndyn)) return 6;
    K = (int)nk.size() - 1;
    nks.assign(K + 1, 0);
    for (int k = 0; k < K; k++) nks[k + 1] = nks[k] + nk[k + 1];
    if (nks[K] != ndyn) return 6;
  }
  me = arows;
  std::vector<int> stage_of(n);
  for (int k = 0; k <= K; k++) {
    const int c1 = k < K ? nmk[k + 1] : n;
    for (int c = nmk[k]; c < c1; c++) stage_of[c] = k;
  }
  auto lcol = [&](int c) { return c - nmk[stage_of[c]]; };

  // ------------------------------------------------------------------ rows per stage
  // dynamics rows stay inside their stage (Check_Structure, hqp/Hqp_IpLQDOCP.C:304-313)
  if (!dense_dyn)
    for (int k = 0; k < K; k++)
      for (int i = nks[k]; i < nks[k + 1]; i++) {
        const int p0 = Ap[i], p1 = Ap[i + 1];
        if (p1 - p0 < 2 || stage_of[Ai[p0]] != k || stage_of[Ai[p1 - 2]] != k || stage_of[Ai[p1 - 1]] != k + 1)
          return 6;
      }
  const int row0 = ndyn;  // first non-dynamics row
  std::vector<std::vector<int>> eq(K + 1);
  for (int i = row0; i < arows; i++) {
    const int p0 = Ap[i], p1 = Ap[i + 1];
    if (p1 == p0) return 6;
    const int k = stage_of[Ai[p0]];
    if (stage_of[Ai[p1 - 1]] != k) return 6;
    eq[k].push_back(i);
  }
  for (int i = 0; i < m; i++) {
    const int p0 = Cp[i], p1 = Cp[i + 1];
    if (p1 == p0) return 6;
    if (stage_of[Ci[p0]] != stage_of[Ci[p1 - 1]]) return 6;
  }
  for (int i = 0; i < n; i++) {
    const int p0 = Qp[i], p1 = Qp[i + 1];
    if (p1 > p0 && (stage_of[Qi[p0]] != stage_of[i] || stage_of[Qi[p1 - 1]] != stage_of[i])) return 6;
  }
  // fixed initial state: a singleton row for every component of x_0 (hqp/Hqp_IpLQDOCP.C:343-351)
  {
    std::vector<int> frow(nk[0], -1), fsrc(nk[0], -1);
    int found = 0;
    for (int i : eq[0])
      if (Ap[i + 1] - Ap[i] == 1 && Ai[Ap[i]] < nk[0] && frow[Ai[Ap[i]]] < 0)
        frow[Ai[Ap[i]]] = i, fsrc[Ai[Ap[i]]] = Ap[i], found++;
    fixed_x0 = nk[0] > 0 && found == nk[0];
    fix_rows.clear(), fix_src.clear();
    if (fixed_x0) {
      std::vector<char> isfix(arows, 0);
      for (int j = 0; j < nk[0]; j++) isfix[frow[j]] = 1;
      std::vector<int> rest;
      for (int i : eq[0])
        if (!isfix[i]) rest.push_back(i);
      eq[0].swap(rest);
      fix_rows = frow;
      fix_src.resize(nk[0]);
      for (int j = 0; j < nk[0]; j++) fix_src[j] = nq + fsrc[j];
    }
  }
  eq_ptr.assign(K + 2, 0);
  eq_rows.clear();
  for (int k = 0; k <= K; k++) {
    for (int i : eq[k]) eq_rows.push_back(i);
    eq_ptr[k + 1] = (int)eq_rows.size();
  }

  // ------------------------------------------------------------------ capacities
  cap.assign(K + 1, 0), capn.assign(K + 1, 0), qmax.assign(K + 1, 0);
  cap[K] = capn[K] = (int)eq[K].size();
  if (cap[K] > CARRY_MAX) return 1;
  for (int k = K - 1; k >= 0; k--) {
    capn[k] = (int)eq[k].size() + cap[k + 1];
    // structural bound (nothing consumed), cut at what the kernels are built for: a stage that
    // would carry more than CARRY_MAX rows reports HQPKKT_E_SIZES at run time
    cap[k] = std::min(capn[k], CARRY_MAX);
    qmax[k] = mk[k] + std::min(mk[k], capn[k]);
    // at most mk[k] rows can be consumed by this stage's controls: more than CARRY_MAX are left for sure
    if (capn[k] - mk[k] > CARRY_MAX || mk[k] > CONTROLS_MAX) return 1;
  }
  q0max = fixed_x0 ? 0 : nk[0] + cap[0];
  // Stages whose control-sized work fits the LDS of one CU (~150 KB) run it there; the others ("big": some
  // hundreds of controls or carried rows) run the same elimination with their matrices in a scratch area of
  // the misc arena (k_st_small with SmallArgs::scratch, 1024 threads)
  big.assign(K + 1, 0);
  scratch_elems = 0;
  for (int k = 0; k < K; k++) {
    const long long q = qmax[k];
    const long long a = (long long)capn[k] * (mk[k] + capn[k]) * 8 + (capn[k] + mk[k] + 2) * 4LL + capn[k] * 8LL;
    const long long b = gj_lds_bytes(q);
    if (std::max(a, b) + 256 > 150 * 1024) {
      big[k] = 1;
      // the scaled K (rows of up8(q) doubles), its scaling, and for the blocked elimination (k_blk_*) two panels of 64
      // rows, one 64 x 64 block and a few flags; the constraint rows of phase (A) use the same area before K is built
      const long long ldk = (q + 7) / 8 * 8;
      scratch_elems = std::max(scratch_elems, std::max((long long)capn[k] * (mk[k] + capn[k]), q * ldk + ldk + 2 * 64 * ldk + 64 * 64 + 16) + 64);
    }
  }
  big0 = 0;
  if (!fixed_x0) {
    const long long q = q0max;
    if (gj_lds_bytes(q) + 256 > 150 * 1024) {
      // a free initial state of many components: [V_0 B_0'; B_0 0] of order n_0 + carried rows is inverted by the
      // blocked sweep on the whole chip, with the LU factorisation by ONE workgroup out of global memory behind it
      // where the sweep gives up - up to order 4096 (the LDS vectors of that workgroup); beyond that HQPKKT_E_SIZES
      if (q > 4096) return 1;
      big0 = 1;
      // (the blocked inverse, k_x0_*: the scaled matrix in rows of up8(q), scaling, two panels, one block, flags;
      // the one-workgroup LU behind it: q rows of q | 1)
      const long long ldk = (q + 7) / 8 * 8;
      scratch_elems = std::max(scratch_elems, std::max(q * (q | 1), q * ldk + ldk + 2 * 64 * ldk + 64 * 64 + 16) + 64);
    }
  }

  // ------------------------------------------------------------------ storage
  ldf.assign(K + 1, 8), ldv.assign(K + 1, 8), ldy.assign(K + 1, 8), ldn.assign(K + 1, 8);
  ldb.assign(K + 1, 8), ldq.assign(K + 1, 8), ldt.assign(K + 1, 8), ldg.assign(K + 1, 8);
  oF.assign(K + 1, 0), oV.assign(K + 1, 0);
  oY.assign(K + 1, 0), oR.assign(K + 1, 0), oK.assign(K + 1, 0), oKm.assign(K + 1, 0), oN.assign(K + 1, 0);
  oBT.assign(K + 1, 0), oT.assign(K + 1, 0), oVec.assign(K + 1, 0);
  long long fo = 0, vo = 0, mo = 0;
  long long wmax = 0, gmax = 0, resmax = 0;
  int nzmax = 0, nmax = 0;
  for (int k = 0; k <= K; k++) {
    const int nz = k < K ? nk[k] + mk[k] : nk[k];
    nzmax = std::max(nzmax, nz), nmax = std::max(nmax, nk[k]);
    ldv[k] = up8(nk[k]);
    oV[k] = vo, vo += up16((long long)nk[k] * ldv[k]);
    ldb[k] = up8(std::max(cap[k], 1));
    oBT[k] = mo, mo += up16((long long)nk[k] * ldb[k]);
    ldn[k] = up8(std::max(nz, 1));
    oN[k] = mo, mo += up16((long long)std::max(capn[k], 1) * ldn[k]);
    oVec[k] = mo, mo += up16((long long)nk[k] + 2LL * std::max(cap[k], 1) + std::max(qmax[k], 1) + 8);
    if (k < K) {
      ldf[k] = up8(nz), ldg[k] = up8(nz);
      oF[k] = fo, fo += up16((long long)nk[k + 1] * ldf[k]);
      ldy[k] = up8(std::max(nk[k], 1)), ldq[k] = up8(std::max(qmax[k], 1)), ldt[k] = up8(std::max(mk[k], 1));
      oY[k] = mo, mo += up16((long long)std::max(qmax[k], 1) * ldy[k]);
      oR[k] = mo, mo += up16((long long)std::max(qmax[k], 1) * ldy[k]);
      oK[k] = mo, mo += up16((long long)std::max(qmax[k], 1) * ldq[k]);
      oKm[k] = mo, mo += up16((long long)std::max(qmax[k], 1) * ldq[k]);
      resmax = std::max(resmax, (long long)std::max(qmax[k], 1) * ldy[k]);
      oT[k] = mo, mo += up16((long long)std::max(cap[k], 1) * ldt[k]);
      if (!sharded) wmax = std::max(wmax, (long long)nk[k + 1] * ldf[k]);  // (sharded: W_p goes straight into its exchange slot)
      gmax = std::max(gmax, (long long)nz * ldg[k]);
    }
  }
  oW = mo, mo += up16(wmax);
  oG = mo, mo += up16(gmax);
  ldq0 = up8(std::max(q0max, 1));
  oK0 = mo, mo += up16((long long)std::max(q0max, 1) * ldq0);
  oK0m = mo, mo += up16((long long)std::max(q0max, 1) * ldq0);
  // scaling and the two permutations | which form the area K0 holds | four vectors of the solve with the inverse
  oK0s = mo, mo += up16(3LL * std::max(q0max, 1) + 8 + 4LL * up8(std::max(q0max, 1)));
  oRes = mo, mo += up16(resmax);
  oGam = mo, mo += up16(nzmax + 8);
  {
    int qm = 1;
    for (int k = 0; k < K; k++) qm = std::max(qm, qmax[k]);
    oUy = mo, mo += up16(qm + 8);
  }
  oTT = mo, mo += up16(nmax + 8);
  oTmp = mo, mo += up16(nmax + 8);
  part_chunks = std::max(1, std::min(64, nmax / 64));
  oPart = mo, mo += up16((long long)part_chunks * (nzmax + 8));
  // partial sums of the symmetric products with V (k_st_symv_tiles: 64-row and 512-column tiles)
  oSym = mo, mo += up16(((long long)(nmax + 63) / 64 + (nmax + 511) / 512) * (nmax + 8));
  oGv = oSymB = 0, symb_elems = 0;
  if (!sharded) {
    long long all = 0;
    for (int k = 1; k <= K; k++) all += up16(symv_need(nk[k]));
    symb_elems = std::min(all, std::max(up16(symv_need(nmax)), 16LL << 20));  // (up to 128 MB: 35 stages of 5000 states per launch)
    oGv = mo, mo += up16(ndyn + 8);
    oSymB = mo, mo += up16(symb_elems);
  }
  oS = mo, mo += up16(n + 8);
  oQv = mo, mo += up16(n + 8);
  oScr = mo, mo += up16(scratch_elems);
  // ------------------------------------------------------------------ one system over several ranks (staged_plan.hpp)
  xcut.clear(), xw.clear(), ldfl.clear(), oFl.clear(), oVs.clear(), fgslot.clear(), ldwl.clear(), xslot.clear();
  xrects.clear(), xrect_ptr.clear(), gtile.clear(), gtile_ptr.clear();
  oX = oXV = oXP = oDyx = oWl = oWu = 0, xvslot = xpslot = 0, oVf[0] = oVf[1] = 0, oFg[0] = oFg[1] = 0;
  if (sharded) {
    const int P = shard_count, me_ = shard_rank;
    if (P < 1 || P > 16 || me_ < 0 || me_ >= P) return 1;
    xcut.assign((size_t)(K + 1) * (P + 1), 0), xw.assign(K + 1, 0);
    ldfl.assign(K + 1, 8), oFl.assign(K + 1, 0), oVs.assign(K + 1, 0), fgslot.assign(K + 1, 0), xslot.assign(K + 1, 0);
    ldwl.assign(K + 1, 8);
    long long flo = 0, vso = 0, vfmax = 0, fgmax = 0, xmax = 0, wlmax = 0, wumax = 0;
    int mmax = 1;
    for (int k = 0; k <= K; k++) {
      const long long nn = nk[k];
      if (k < K && (nn & 1)) return 1;  // the control columns of Floc start behind the strip: 16-byte loads need it even
      const long long T = (nn + 127) / 128;
      xw[k] = (int)(128 * ((T + P - 1) / P));
      int *cut = &xcut[(size_t)k * (P + 1)];
      for (int p = 0; p <= P; p++) cut[p] = (int)std::min<long long>(nn, (long long)p * xw[k]);
      const long long wd = cut[me_ + 1] - cut[me_];
      oVs[k] = vso, vso += up16(wd * ldv[k]);
      vfmax = std::max(vfmax, up16(nn * ldv[k]));
      xvslot = std::max<long long>(xvslot, up16(xw[k]));
      if (k < K) {
        ldfl[k] = up8(wd + mk[k]);
        oFl[k] = flo, flo += up16((long long)nk[k + 1] * ldfl[k]);
        const long long np = nk[k + 1];
        fgslot[k] = up16(np * up8(xw[k] + mk[k]));  // (every rank's local block: its strip and the control columns)
        fgmax = std::max(fgmax, fgslot[k]);
        ldwl[k] = up8(std::max<long long>(wd, 1));
        wlmax = std::max(wlmax, np * ldwl[k]);
        mmax = std::max(mmax, mk[k]);
        wumax = std::max(wumax, np * up8(std::max(mk[k], 1)));
      }
    }
    // the blocks of G_xx: who computes which
    xrect_ptr.assign(K + 1, 0), gtile_ptr.assign(K + 1, 0);
    const int D = (P - 1) / 2;
    for (int k = 0; k < K; k++) {
      const int *cut = &xcut[(size_t)k * (P + 1)];
      xrect_ptr[k] = (int)xrects.size(), gtile_ptr[k] = (int)gtile.size();
      std::vector<long long> fill(P, 0);
      auto add = [&](int a, int b, int r0, int r1, int owner) {
        if (r1 <= r0 || cut[b + 1] <= cut[b]) return;
        XRect r{a, b, r0, r1, cut[b], cut[b + 1], owner, owner == a, fill[owner]};
        fill[owner] += up16((long long)(r1 - r0) * (cut[b + 1] - cut[b]));
        xrects.push_back(r);
      };
      for (int a = 0; a < P; a++)
        for (int b = 0; b <= a; b++) {
          const int d = a - b;
          if (d == 0 || d <= D)
            add(a, b, cut[a], cut[a + 1], d == 0 ? a : b);
          else if (P - d <= D)
            add(a, b, cut[a], cut[a + 1], a);
          else {  // P even, the two blocks P / 2 apart: the upper rows to b, the rest to a
            const int h = cut[a + 1] - cut[a], rs = cut[a] + ((h / 2 + 127) / 128) * 128;
            add(a, b, cut[a], std::min(rs, cut[a + 1]), b);
            add(a, b, std::min(rs, cut[a + 1]), cut[a + 1], a);
          }
        }
      for (int p = 0; p < P; p++) xslot[k] = std::max(xslot[k], fill[p]);
      xslot[k] = up16(xslot[k]);
      xmax = std::max(xmax, xslot[k]);
      // this rank's tiles, in its row strip of the work block: (tile row in the strip) << 16 | global tile column
      for (int q = xrect_ptr[k]; q < (int)xrects.size(); q++) {
        const XRect &r = xrects[q];
        if (r.owner != me_) continue;
        // work rectangle: the block itself, or its transpose (rows = own columns)
        const int R0 = r.mine_rows ? r.r0 : r.c0, R1 = r.mine_rows ? r.r1 : r.c1, C0 = r.mine_rows ? r.c0 : r.r0,
                  C1 = r.mine_rows ? r.c1 : r.r1;
        for (int tn = C0 / 128; tn < (C1 + 127) / 128; tn++)
          for (int tm = R0 / 128; tm < (R1 + 127) / 128; tm++) {
            if (r.a == r.b && tn > tm) continue;  // diagonal block: lower tiles
            gtile.push_back(((tm - cut[me_] / 128) << 16) | tn);
          }
      }
    }
    xrect_ptr[K] = (int)xrects.size(), gtile_ptr[K] = (int)gtile.size();
    oFg[0] = mo, mo += up16(fgmax * P);
    oFg[1] = mo, mo += up16(fgmax * P);
    ldwu = up8(mmax);
    oWl = mo, mo += up16(wlmax + 8);
    oWu = mo, mo += up16(wumax + 8);
    oX = mo, mo += up16(xmax * P);
    oXV = mo, mo += up16(xvslot * P);
    xpslot = up16(nmax + 8);
    oXP = mo, mo += up16(xpslot * P);
    oDyx = mo, mo += up16((long long)ndyn + nk[0] + 8);
    oVf[0] = mo, mo += vfmax, oVf[1] = mo, mo += vfmax;  // (work blocks, like W and G)
    fo = flo, vo = vso;
  }
  f_elems = fo, v_elems = vo, misc_elems = mo;
  dyn_off.assign(K + 1, 0);
  dyn_ints = 0;
  for (int k = 0; k <= K; k++) dyn_off[k] = dyn_ints, dyn_ints += 2 + 2 * std::max(capn[k], 1);

  // ------------------------------------------------------------------ scatter of A
  a_dst.assign(na, -1);
  chk_idx.clear(), chk_kind.clear();
  if (!dense_dyn)
    for (int k = 0; k < K; k++)
      for (int i = nks[k]; i < nks[k + 1]; i++) {
        const int li = i - nks[k];
        for (int p = Ap[i]; p < Ap[i + 1] - 1; p++) {
          const int lc = Ai[p] - nmk[k];
          if (!sharded)
            a_dst[p] = oF[k] + (long long)li * ldf[k] + lc;
          else {  // the local block: own state columns, then the control columns
            const int c0 = xcut[(size_t)k * (shard_count + 1) + shard_rank], c1 = xcut[(size_t)k * (shard_count + 1) + shard_rank + 1];
            if (lc >= nk[k])
              a_dst[p] = oFl[k] + (long long)li * ldfl[k] + (c1 - c0) + (lc - nk[k]);
            else if (lc >= c0 && lc < c1)
              a_dst[p] = oFl[k] + (long long)li * ldfl[k] + (lc - c0);
          }
        }
        chk_idx.push_back(nq + Ap[i + 1] - 1), chk_kind.push_back(0);
      }
  for (int k = 0; k <= K; k++)
    for (int li = 0; li < (int)eq[k].size(); li++) {
      const int i = eq[k][li];
      for (int p = Ap[i]; p < Ap[i + 1]; p++) a_dst[p] = -(oN[k] + (long long)li * ldn[k] + lcol(Ai[p]) + 2);
    }
  for (int s : fix_src) chk_idx.push_back(s), chk_kind.push_back(1);

  // ------------------------------------------------------------------ H term lists
  {
    struct Raw {
      long long key;  // stage << 45 | (touches a control row / column) << 44 | li * ld + lj
      Term t;
    };
    std::vector<Raw> raw;
    raw.reserve((size_t)nq * 2 + (size_t)nc * 2);
    const int ONE = nq + na + nc, WONE = m;
    auto ldof = [&](int k) { return k < K ? ldg[k] : ldv[K]; };
    auto push = [&](int k, int li, int lj, Term t) {
      const long long ctl = (li >= nk[k] || lj >= nk[k]) ? 1 : 0;
      raw.push_back({((long long)k << 45) | (ctl << 44) | ((long long)li * ldof(k) + lj), t});
    };
    for (int i = 0; i < n; i++)
      for (int p = Qp[i]; p < Qp[i + 1]; p++) {
        const int j = Qi[p];
        if (j < i) continue;  // only col >= row is read (meschach/addon2_hqp.c:1078-1086)
        const int k = stage_of[i];
        push(k, lcol(i), lcol(j), Term{p, ONE, WONE});
        if (j != i) push(k, lcol(j), lcol(i), Term{p, ONE, WONE});
      }
    for (int r = 0; r < m; r++) {
      const int k = stage_of[Ci[Cp[r]]];
      for (int pa = Cp[r]; pa < Cp[r + 1]; pa++)
        for (int pb = Cp[r]; pb < Cp[r + 1]; pb++)
          push(k, lcol(Ci[pa]), lcol(Ci[pb]), Term{nq + na + pa, nq + na + pb, r});
    }
    std::stable_sort(raw.begin(), raw.end(), [](const Raw &a, const Raw &b) { return a.key < b.key; });
    h_ptr.assign(K + 2, 0), h_tptr.assign(1, 0), h_dst.clear(), h_terms.clear();
    std::vector<int> nstate(K + 1, 0);
    h_terms.reserve(raw.size());
    long long prev = -1;
    for (const Raw &e : raw) {
      if (e.key != prev) {
        h_dst.push_back(e.key & ((1LL << 44) - 1));
        h_tptr.push_back(h_tptr.back());
        h_ptr[(int)(e.key >> 45) + 1]++;
        if (!((e.key >> 44) & 1)) nstate[(int)(e.key >> 45)]++;
        prev = e.key;
      }
      h_terms.push_back(e.t);
      h_tptr.back()++;
    }
    for (int k = 0; k <= K; k++) h_ptr[k + 1] += h_ptr[k];
    h_mid.assign(K + 1, 0);
    for (int k = 0; k <= K; k++) h_mid[k] = h_ptr[k] + nstate[k];
  }

  // ------------------------------------------------------------------ work counts
  flops_factor = 0, bytes_step = 0;
  for (int k = 0; k < K; k++) {
    const long long nn = nk[k], mm = mk[k], np = nk[k + 1], nz = nn + mm, q = qmax[k];
    flops_factor += 2 * np * np * nz;       // W = V+ F
    flops_factor += np * nz * nz;           // G = F' W, lower half
    flops_factor += 2 * (long long)cap[k + 1] * np * nz;  // carried rows
    flops_factor += 2 * q * q * nn + q * nn * nn;         // Rm, V update (lower half)
    bytes_step += 8 * (2 * np * np + 2 * np * nz + 2 * q * nn);
  }
  return 0;
}

}  // namespace kktdev
