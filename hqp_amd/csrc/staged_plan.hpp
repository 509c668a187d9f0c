// Host-side symbolic phase of the STAGED engine: stage dimensions from the staircase of
// A (semantics of Hqp_IpLQDOCP::Get_Dim / Get_Constr_Dim / Check_Structure,
// hqp/Hqp_IpLQDOCP.C:201-287, 368-407, 298-354), static capacities of the carried
// constraint rows, storage plan of the dense stage blocks, term lists of the reduced
// Hessian H = Q + C'(Z/W)C and the scatter maps of the CSR values.  Integer work only.
#pragma once
#include <vector>

namespace kktdev {

// LDS bytes of the one-workgroup inverse of an order-q matrix (gj_inverse_any in staged.hip.h):
// the matrix, its scaling, the pivot column (>= 64) and row (>= 128: the register form broadcasts
// the augmented row), three int work arrays
inline long long gj_lds_bytes(long long q) {
  const long long cv = q > 64 ? q : 64, rv = q > 128 ? q : 128;
  return q * (q | 1) * 8 + (q + cv + rv) * 8 + 3 * q * 4 + 64;
}

int stages_from_staircase(int n, int rows, const int *row_len, const int *last_col, const int *prev_col,
                          std::vector<int> &states, std::vector<int> &controls, std::vector<int> &first_col, int &dyn_rows);

struct StagedPlan {
  int n = 0, me = 0, m = 0, K = 0;
  int nq = 0, na = 0, nc = 0;
  std::vector<int> nk, mk, nmk;  // states (K+1), controls (K), first column of a stage (K+1)
  std::vector<int> nks;          // first dynamics row of stage k (K+1 entries, nks[K] = ndyn)
  int ndyn = 0;
  std::vector<int> eq_ptr, eq_rows;  // own equality rows per stage (QP row indices), K+2 / total
  bool fixed_x0 = false;
  std::vector<int> fix_rows, fix_src;  // per x_0 component: its row of A and the index of its value in vals
  bool dense_dyn = false;  // dynamics handed over as dense blocks: their rows of A are empty

  // static bounds: cap[k] carried rows leaving stage k, capn[k] rows of N_k, qmax[k] order of K_k
  std::vector<int> cap, capn, qmax;
  int q0max = 0, ldq0 = 8;  // free initial state: order of [V_0 B_0'; B_0 0]
  std::vector<char> big;    // per stage: its control-sized matrices do not fit LDS (scratch in the misc arena instead)
  int big0 = 0;             // the same for the free initial state
  long long scratch_elems = 0, oScr = 0;

  // dense storage (element offsets; leading dimensions are multiples of 8)
  std::vector<int> ldf, ldv, ldy, ldn, ldb, ldq, ldt, ldg;
  std::vector<long long> oF, oV;                      // F arena, V arena
  std::vector<long long> oY, oR, oK, oKm, oN, oBT, oT;  // misc arena (oKm: K itself, next to its inverse oK)
  std::vector<long long> oVec;                        // per stage: v(n) beta(cap) rho(qmax) eta(cap)
  long long oW = 0, oG = 0, oK0 = 0, oK0m = 0, oK0s = 0, oRes = 0, oGam = 0, oTT = 0, oPart = 0, oS = 0, oQv = 0, oTmp = 0, oUy = 0, oSym = 0;
  // the solve's products with V that stand outside its two chains, many stages per launch (not sharded): g_k = V_{k+1} f_k
  // for all stages (ndyn doubles) and the partial sums of one launch's stages (symb_elems; symv_need(N) per stage)
  long long oGv = 0, oSymB = 0, symb_elems = 0;
  static long long symv_need(long long N) { return ((N + 63) / 64 + (N + 511) / 512) * (N + 8); }
  long long f_elems = 0, v_elems = 0, misc_elems = 0;
  int part_chunks = 1;
  std::vector<int> dyn_off;  // int arena: per stage [r, nl, R(capn), L(capn)]
  int dyn_ints = 0;

  // H term lists: entries of stage k are h_ptr[k] .. h_ptr[k+1]; dst = li * ld + lj inside the
  // stage's dense block (ld = ldg[k], last stage: ldv[K])
  struct Term {
    int s1, s2, wi;
  };
  std::vector<int> h_ptr, h_tptr;
  std::vector<int> h_mid;  // per stage: first entry that touches a control row / column (the state part comes first)
  std::vector<long long> h_dst;
  std::vector<Term> h_terms;

  // scatter of the A values: >= 0 offset into the F arena, <= -2: -(offset into misc + 2), -1: none
  std::vector<long long> a_dst;
  std::vector<int> chk_idx, chk_kind;  // values to check: kind 0 must be -1.0, kind 1 must be non-zero

  long long flops_factor = 0;  // as implemented (dense products of the recursion)
  long long bytes_step = 0;    // dense bytes one step streams

  // One system over several ranks (hqpkkt_set_shard), DESIGN.md section 7: the STATE COLUMNS of every stage are cut into one
  // contiguous range per rank - the same width for every rank but the last, a multiple of 128 - and the memory goes
  // with them: rank p keeps the columns [cut[p], cut[p+1]) of F_k next to the control columns (Floc_k = [F_p | F_u],
  // n+ x ldfl) and, for the solve, the ROWS [cut[p], cut[p+1]) of V_k.  Per stage of the factorisation:
  //   W_p = V+ F_p                                   local (V+ in full: the transient result of the stage before); W is
  //                                                  NOT exchanged
  //   gather of the ranks' Floc_k                    STATIC data: requested a stage ahead (stage k - 1's while stage k is
  //                                                  computed), into one of two buffers - off the critical path
  //   G_xx block (a, b), a >= b                      by one of the two ranks, as W_p' F_q = (V+ F_p)' F_q in its row strip
  //                                                  of the work block (the owner of the block's columns computes the
  //                                                  transpose): pair {a, b} belongs to b if a - b <= (P - 1) / 2 else to
  //                                                  a; with P even the pairs P / 2 apart are cut in two by rows;
  //                                                  diagonal blocks: lower tiles
  //   gather of the blocks (lower orientation)       n^2 / 2 doubles in all: the one exchange on the critical path
  //   V_k = G_xx - Y' Rm in full (every rank), its row strip kept
  // The control-sized chain (W_u = V+ F_u, the control rows of G = W_u' F, the carried rows B+ F, K^-1, Y, Rm) is
  // computed by every rank from the gathered F - by launches of the same shape everywhere, so that all ranks see
  // identical bits.  The solve's products run on the strips with one gather of a state-sized vector per stage and direction.
  int shard_rank = 0, shard_count = 1;
  bool sharded = false;          // several ranks, or one rank with a transport set (tests the exchange path)
  std::vector<int> xcut;         // (K+1) x (shard_count+1): first state column of rank p in stage k (k = K: rows of V_K)
  std::vector<int> xw;           // per stage (K+1): the common strip width (all ranks but the last)
  std::vector<int> ldfl;         // per stage: leading dimension of Floc_k (own strip + control columns)
  std::vector<long long> oFl, oVs;  // local F blocks (F arena), own row strips of V_k (V arena, ld = ldv[k])
  long long oVf[2] = {0, 0};     // the two full-size transient V blocks (misc arena): V_k lives in oVf[k & 1]
  long long oFg[2] = {0, 0};     // the gathered local F blocks: stage k's in oFg[k & 1], shard_count slots of fgslot[k] doubles (misc)
  std::vector<long long> fgslot;
  long long oWl = 0, oWu = 0;    // W_p = V+ F_p (n+ x ldwl[k]) and W_u = V+ F_u (n+ x ldwu) of the stage in work (misc)
  std::vector<int> ldwl;
  int ldwu = 8;
  long long oX = 0;              // the gather of the blocks of G_xx: shard_count slots of xslot[k] doubles (misc)
  std::vector<long long> xslot;
  long long oXV = 0;             // gathers of the solve: the ranks' strips of a state-sized vector, side by side (misc)
  long long xvslot = 0;
  long long oXP = 0;             // ... and shard_count whole state-sized vectors (the partial sums of x+ = F s + f)
  long long xpslot = 0;
  long long oDyx = 0;            // the dynamics rows' multipliers (+ V_0 x_0 + v_0 with a fixed x_0): summed over the ranks
  // a block (part) of G_xx in the exchange buffer 2, lower orientation: rows [r0, r1) x columns [c0, c1) of G_xx,
  // row-major with leading dimension c1 - c0 at slot `owner`, offset `off`; how the owner computes it: `mine_rows`
  // (A = its own F strip: the block is the rows of its strip) or not (A = W of the row block's owner, B = its own F strip:
  // computed transposed in its row strip of the work block, packed with a transposition)
  struct XRect {
    int a, b;            // block row / column (ranks)
    int r0, r1, c0, c1;
    int owner;
    bool mine_rows;
    long long off;       // inside the owner's slot
  };
  std::vector<XRect> xrects;      // all stages, all ranks: stage k holds xrect_ptr[k] .. xrect_ptr[k+1]
  std::vector<int> xrect_ptr;
  std::vector<int> gtile;         // this rank's tiles of its blocks' products: (tile row in the strip) << 16 | global tile column
  std::vector<int> gtile_ptr;     // per stage

  // explicit stage sizes (hqpkkt_set_stages): K, nx[K+1], nu[K]; empty: detect from A
  std::vector<int> given_nx, given_nu;

  // returns 0, or a HQPKKT_E_* code: 6 the pattern is not a staircase / rows leave their stage,
  // 1 a stage exceeds what the one-workgroup kernels hold
  int run(int n, int me, int m, const int *Qp, const int *Qi, const int *Ap, const int *Ai, const int *Cp,
          const int *Ci);
};

}  // namespace kktdev
