// Host side of the cut form of the STAGED engine's fp64 product (k_dgemm_tn_sk, staged.hip.h): the units of work of
// every workgroup as a table.  Plain C++ (no device code): used by staged.hip.h and, through hqpkkt_debug_sk_table, by
// the CPU tests.
#pragma once
#include <algorithm>
#include <vector>

namespace stg {
// one unit of work of a workgroup: the k-slabs [s0, s1) of tile `tile`; piece j of `pieces` of that tile (pieces == 1:
// the whole tile); the tile's pieces park their partial sums in the slots slot0 ... slot0 + pieces - 1 (order of k)
struct SkUnit {
  int tile;  // < 0: end of the workgroup's list
  unsigned short s0, s1;
  int slot0;
  unsigned short pieces, j;
};
static_assert(sizeof(SkUnit) == 16, "SkUnit is read as one 16-byte word");

// UNEQUAL shares for the two workgroups of a CU.  What the stamps of the headline's products say
// (profiles/r03_dgemm_stamps.txt, r06_sk_stamps.txt; tiles of 313 k-slabs): of the two workgroups a CU holds, the one
// dispatched first (class A: blockIdx.x < grid / 2) finishes a tile in ~1000 us and the second (class B) in ~1530 us
// while both run - the older wavefronts win the arbitration for the matrix pipe -, and a workgroup alone on its CU
// takes ~700 us.  With equal shares (gemm_split_plan: W = 3 whole tiles + an eighth for everybody) class A is done at
// 3090 us and the launch ends when class B is, at 3870 us.  And cut tiles are dear: a plan with 2560 parked pieces
// instead of 256 takes 4.4 instead of 3.76 ms (tools/sk_sweep.py, profiles/r06_sk_sweep.txt) - the pieces' pipeline
// fills, their parked sums and workgroups that no longer walk the same k.  So: WHOLE tiles as far as they go, more of
// them for class A (nA per workgroup) than for class B (nB), and only the remainder of less than grid / 2 tiles cut -
// by class B, by class A or by both, in at most two groups.  The candidates are compared by a model of the pace
// (sk_model_makespan) and the best one is listed per workgroup; within a round the workgroups of a class take
// neighbouring tiles in the order of their position in the XCD (the swizzled index), as the equal-share plan does.
struct SplitTable {
  std::vector<SkUnit> units;  // grid x stride
  int stride = 0;
  long long pieces = 0;   // parking slots
  double makespan = 0.0;  // the model's, in tile times of class A
  int nA = 0, nB = 0;     // whole tiles per workgroup of the two classes
};
static inline int xcd_swizzle_host(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return x * q + (x < r ? x : r) + (bid >> 3);
}
// the pace model: work wa / wb (in tiles) of a workgroup of class A / B on one CU; while both run a tile takes A 1.0
// and B 1.55, alone 0.71 (measured: 987 / 1532 / 700 us)
static inline double sk_model_makespan(double wa, double wb) {
  const double rA = 1.0, rB = 1.55, alone = 0.71;
  const double ta = wa * rA, tb = wb * rB;
  return tb <= ta ? tb + (wa - tb / rA) * alone : ta + (wb - ta / rB) * alone;
}
static inline bool gemm_split_table(long long tiles, long long nslab, int grid, SplitTable &best) {
  struct Group {
    long long begin, count;
    int nsplit, cls;  // cls: 3 both classes, 1 A, 2 B
  };
  if (grid < 2 || grid % 2 || nslab >= 65536 || tiles <= 0) return false;
  const long long H = grid / 2, n = tiles / H, R = tiles - n * H;
  const int smax = (int)std::max<long long>(1, std::min<long long>(16, nslab / 16));
  const double piece_cost = 0.04;  // (a cut piece: its pipeline fill and the parked sums, in tile times)
  auto members = [&](int cls) { return (cls == 3 ? 2 : 1) * H; };
  bool have = false;
  double best_t = 0.0;
  std::vector<Group> best_groups;
  for (long long nA = (n + 1) / 2; nA <= n; nA++) {
    const long long nB = n - nA;
    // remainder: R1 tiles cut s1 ways over c1, the other R - R1 tiles cut s2 ways over c2 (s2 as large as its class allows)
    for (int c1 : {2, 1, 3})
      for (int s1 = 1; s1 <= smax; s1 = s1 < 4 ? s1 + 1 : s1 * 2) {
        const long long R1 = std::min(R, members(c1) / s1);
        const long long R2 = R - R1;
        for (int c2 : {0, 1, 2, 3}) {
          if ((R2 == 0) != (c2 == 0)) continue;
          if (c2 == c1) continue;  // (the same class twice: one group of it)
          int s2 = 0;
          if (c2) {
            s2 = (int)std::min<long long>(smax, members(c2) / R2);
            if (s2 < 1) continue;
          }
          double ea = 0.0, eb = 0.0;  // extra work of the busiest workgroup of each class
          if (R1 > 0) {
            const double w = 1.0 / s1 + (s1 > 1 ? piece_cost : 0.0);
            if (c1 & 1) ea += w;
            if (c1 & 2) eb += w;
          }
          if (c2) {
            const double w = 1.0 / s2 + (s2 > 1 ? piece_cost : 0.0);
            if (c2 & 1) ea += w;
            if (c2 & 2) eb += w;
          }
          const double t = sk_model_makespan(nA + ea, nB + eb);
          if (!have || t < best_t - 1e-9) {
            have = true, best_t = t;
            best_groups.clear();
            long long next = 0;
            for (long long r = 0; r < nB; r++) best_groups.push_back({next, 2 * H, 1, 3}), next += 2 * H;
            for (long long r = nB; r < nA; r++) best_groups.push_back({next, H, 1, 1}), next += H;
            if (R1 > 0) best_groups.push_back({next, R1, s1, c1}), next += R1;
            if (c2) best_groups.push_back({next, R2, s2, c2}), next += R2;
            best.nA = (int)nA, best.nB = (int)nB;
          }
        }
      }
  }
  if (!have) return false;
  // per workgroup its units in the order of the groups; rank within the class by the swizzled index
  std::vector<int> rankA(grid, -1), rankB(grid, -1), rankAll(grid, -1), bid_of_v(grid);
  for (int b = 0; b < grid; b++) bid_of_v[xcd_swizzle_host(b, grid)] = b;
  {
    int ra = 0, rb = 0;
    for (int v = 0; v < grid; v++) {
      const int b = bid_of_v[v];
      rankAll[b] = v;
      if (b < H) rankA[b] = ra++; else rankB[b] = rb++;
    }
  }
  std::vector<long long> slot0(tiles, -1);
  long long slots = 0;
  for (const Group &p : best_groups)
    if (p.nsplit > 1)
      for (long long t = p.begin; t < p.begin + p.count; t++) slot0[t] = slots, slots += p.nsplit;
  std::vector<std::vector<SkUnit>> per(grid);
  for (const Group &p : best_groups)
    for (int b = 0; b < grid; b++) {
      const int r = p.cls == 3 ? rankAll[b] : p.cls == 1 ? rankA[b] : rankB[b];
      if (r < 0 || r >= p.count * p.nsplit) continue;
      const long long ti = r % p.count, j = r / p.count, t = p.begin + ti;
      const long long L = (nslab + p.nsplit - 1) / p.nsplit, s0 = std::min(nslab, j * L), s1 = std::min(nslab, s0 + L);
      per[b].push_back(SkUnit{(int)t, (unsigned short)s0, (unsigned short)s1, (int)(p.nsplit > 1 ? slot0[t] : 0), (unsigned short)p.nsplit, (unsigned short)j});
    }
  size_t stride = 1;
  for (int b = 0; b < grid; b++) stride = std::max(stride, per[b].size() + 1);
  best.stride = (int)stride, best.pieces = slots, best.makespan = best_t;
  best.units.assign((size_t)grid * stride, SkUnit{-1, 0, 0, 0, 0, 0});
  for (int b = 0; b < grid; b++)
    for (size_t i = 0; i < per[b].size(); i++) best.units[(size_t)b * stride + i] = per[b][i];
  return true;
}
}  // namespace stg
