// Host-side symbolic phase of the KKT path: KKT pattern from the Q/A/C blocks,
// RCM ordering (semantics of hqp/sprcm.C:62-420), nested dissection of the RCM
// band into an assembly tree of supernodes, symbolic fronts, and the index maps
// the device kernels consume.  Integer work only; runs once per init().
#pragma once
#include <cstdint>
#include <vector>

namespace kktdev {

// One stored entry of the (permuted, symmetric) matrix that gets factored.
// value = sum over its terms of  sgn * vals[s1] * vals[s2] * wt[wi]
// (vals = [Qx | Ax | Cx | 1.0], wt = [per-inequality weight | 1.0]).
struct Term {
  int s1, s2, wi;
  double sgn;
};

struct Analysis {
  int mode = 0, n = 0, me = 0, m = 0, dim = 0, sbw = -1;
  int nq = 0, na = 0, nc = 0;  // nnz of the Q, A, C blocks as passed in

  // --- entries of the matrix in QP numbering (a <= b not required) ----------
  std::vector<int> ent_a, ent_b;       // QP indices of the two ends
  std::vector<int> term_ptr;           // CSR over terms, size nent+1
  std::vector<Term> terms;
  std::vector<int> diag_ent;           // REDUCED: entry id of the (i,i) entry, i < n

  // --- orderings --------------------------------------------------------------
  std::vector<int> qp2j;  // RCM: QP index -> band position  (reference's _QP2J)
  std::vector<int> q2e;   // QP index -> elimination index
  std::vector<int> e2q;

  // --- assembly tree (node ids are a postorder: children before parents) -----
  int nnodes = 0, nlevels = 0, max_front = 0, max_npiv = 0, max_nbor = 0;
  std::vector<int> piv_start, npiv, nbor, parent, level, child_slot;
  std::vector<long long> bptr;  // border_ptr, size nnodes+1
  std::vector<int> bidx;        // border rows (elimination indices), sorted per node
  std::vector<int> rel;         // same indexing as bidx: local index in the parent front
  std::vector<int> pinv;        // per child: parent front index -> index in the child's border, -1 if none
  std::vector<long long> pinv_off;
  std::vector<long long> panel_off, upd_off, x_off;  // element offsets into the arenas
  long long panel_elems = 0, upd_elems = 0, x_elems = 0, cb_elems = 0;
  std::vector<long long> cb_off;  // contribution-vector offsets (solve)
  std::vector<int> child_ptr, child_idx;  // children lists
  // Per-level work lists of a set of supernodes.
  struct Sched {
    int nnodes = 0;
    long long flops = 0;
    std::vector<int> level_ptr, level_nodes;  // nodes grouped by level
    std::vector<int> level_fsmall;            // per level: leading nodes that are small fronts
    std::vector<int> level_small;             // per level: the following nodes with npiv <= SMALL_PIVOTS
    std::vector<int> level_fs_p, level_fs_b;  // per level: largest npiv / nbor of its small fronts
    std::vector<int> level_sm_p;              // per level: largest npiv of the other small supernodes
    // tiles of the Schur update and slabs of the panel solve, grouped by level
    std::vector<int> upd_tile_ptr, upd_tiles;  // triples (node, ti, tj)
    std::vector<int> upd_big_ptr;              // per level: where its 128 x 128 tiles start (fronts with borders >= UPD_BIG_BORDER)
    std::vector<int> slab_ptr, slabs;          // pairs (node, slab): 32 border rows
    std::vector<int> gslab_ptr, gslabs;        // pairs (node, slab): 64 border rows (solve)
    std::vector<int> cblk_ptr, cblks;          // pairs (node, block of 16 pivot columns)
  };
  Sched sched[2];  // [0] this rank's subtrees (everything when not sharded), [1] replicated top

  // --- one system sharded over several ranks ----------------------
  int shard_rank = 0, shard_count = 1;
  long long upd_pingpong_bytes = 16LL << 30;  // update blocks beyond this: two alternating half-arenas
  bool upd_pingpong = false;
  std::vector<long long> upd_level_off, upd_level_len;  // ping-pong: the range a level's blocks occupy
  int long_chain_pivots = 0;  // > 0: pivots per supernode in the chains of separators of >= LONG_CHAIN_VERTS vertices
  bool small_fronts = true;  // fused one-wavefront kernels for fronts with few pivots and few border rows
  bool amalgamation = false;  // separators absorb their child separators while they stay small fronts
  int ordering = 0;  // 0: nested dissection of the RCM band, 1: nested dissection of the graph itself (irregular sparsity), 2: the same without the reference's RCM pass
  int slack_policy = 2;  // FULL mode, slack rows inside a node: 0 band order, 1 behind all x, 2 behind their own x
  std::vector<int> node_owner;  // owning rank per supernode, -1 = replicated top of the tree
  std::vector<int> xroots;      // subtree roots whose update / contribution blocks are exchanged
  long long upd_x_off = 0, upd_x_slot = 0;  // exchange region of the update arena: shard_count slots
  long long cb_x_off = 0, cb_x_slot = 0;    // same for the solve's contribution vectors
  std::vector<long long> zero_panel;  // (offset, length) pairs this rank clears per factor
  std::vector<signed char> keep_e;  // per elimination index: this rank contributes it to the all-reduce
  std::vector<long long> linv_off;             // explicit inverses of the L11 blocks, p x p each
  long long linv_elems = 0;

  // --- numeric assembly map ----------------------------------------------------
  std::vector<long long> ent_dst;  // offset into the panel arena per entry
  std::vector<int> ent_er, ent_ec; // elimination indices (row >= col) per entry

  // --- SpMV blocks for step()/residuum(): CSR with value indirection ----------
  struct Csr {
    int rows = 0;
    std::vector<int> ptr, col, src;
  };
  Csr Qfull, A, AT, C, CT;

  long long nnz_factor = 0, flops_factor = 0;

  // dimensions and the SpMV blocks only (what step()/residuum() of every mode need)
  int setup_blocks(int mode, int n, int me, int m, const int *Qp, const int *Qi, const int *Ap,
                   const int *Ai, const int *Cp, const int *Ci);
  int run(int mode, int n, int me, int m, const int *Qp, const int *Qi, const int *Ap,
          const int *Ai, const int *Cp, const int *Ci, int leaf_size, int max_pivots,
          int zd_policy = 0);
};

static const int UPD_TILE = 64;   // Schur-update tile edge (rows/cols per workgroup)
static const int LONG_CHAIN_VERTS = 768;
static const int UPD_BIG_BORDER = 2048;  // fronts with at least this many border rows: tiles of twice the edge (k_schur_update_big)
static const int SLAB_ROWS = 32;  // border rows per panel-solve workgroup
static const int SMALL_PIVOTS = 32;  // supernodes up to this size use k_factor_diag_small
static const int SMALL_BORDER = 16;  // ... and with at most this many border rows are "small fronts"

}  // namespace kktdev
