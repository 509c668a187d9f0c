// CPU sanitizer target ("ASan/UBSan on the CPU backend"): the host-side C++ of the product -
// analysis.cpp (RCM, nested dissection, symbolic fronts, shard plans) and staged_plan.cpp (stage detection, storage
// plan, column cuts) - compiled with -fsanitize=address,undefined and run over generated structures: banded KKT
// systems (both plugins, three orderings, sharded over 1 / 3 / 8 ranks), the Prg_DID staircase, multistage DOCPs with
// final / path constraints and a free initial state, dense hand-over, sharded STAGED plans.  tests/test_c_host.py
// builds and runs it; any report of the sanitizers ends the process with a non-zero status.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../hqp_amd/csrc/analysis.hpp"
#include "../../hqp_amd/csrc/staged_plan.hpp"

struct Csr {
  std::vector<int> p, i;
};
static unsigned long long rng_state = 88172645463325252ULL;
static unsigned rnd() {
  rng_state ^= rng_state << 13, rng_state ^= rng_state >> 7, rng_state ^= rng_state << 17;
  return (unsigned)(rng_state >> 11);
}
// banded QP: Q upper band b, A rows over [2i, 2i+b), C = identity on every second variable
static void banded(int n, int b, Csr &Q, Csr &A, Csr &C, int &me, int &m) {
  Q.p.assign(1, 0), A.p.assign(1, 0), C.p.assign(1, 0);
  Q.i.clear(), A.i.clear(), C.i.clear();
  for (int r = 0; r < n; r++) {
    for (int c = r; c < n && c <= r + b; c++)
      if (c == r || rnd() % 3) Q.i.push_back(c);
    Q.p.push_back((int)Q.i.size());
  }
  me = n / 2;
  for (int r = 0; r < me; r++) {
    for (int c = 2 * r; c < n && c < 2 * r + b; c++) A.i.push_back(c);
    A.p.push_back((int)A.i.size());
  }
  m = 0;
  for (int r = 0; r < n; r += 2) C.i.push_back(r), C.p.push_back((int)C.i.size()), m++;
}
// DOCP staircase: K stages, nx states, nu controls; own equality rows per stage (path), final-state rows, x_0 fixed or free
static void docp(int K, int nx, int nu, int path, int fin, bool fixed0, Csr &Q, Csr &A, Csr &C, int &n, int &me, int &m) {
  const int nz = nx + nu;
  n = K * nz + nx;
  Q.p.assign(1, 0), A.p.assign(1, 0), C.p.assign(1, 0);
  Q.i.clear(), A.i.clear(), C.i.clear();
  for (int r = 0; r < n; r++) Q.i.push_back(r), Q.p.push_back((int)Q.i.size());
  for (int k = 0; k < K; k++)
    for (int li = 0; li < nx; li++) {
      for (int c = 0; c < nz; c++) A.i.push_back(k * nz + c);
      A.i.push_back((k + 1) * nz + li);
      A.p.push_back((int)A.i.size());
    }
  if (fixed0)
    for (int j = 0; j < nx; j++) A.i.push_back(j), A.p.push_back((int)A.i.size());
  for (int k = 0; k < K; k++)
    for (int e = 0; e < path; e++) {
      for (int c = 0; c < nz; c++)
        if ((c + e) % 2 == 0) A.i.push_back(k * nz + c);
      A.p.push_back((int)A.i.size());
    }
  for (int e = 0; e < fin; e++) {
    for (int c = 0; c < nx; c++)
      if ((c + e) % 3 != 1) A.i.push_back(K * nz + c);
    A.p.push_back((int)A.i.size());
  }
  me = (int)A.p.size() - 1;
  m = 0;
  for (int k = 0; k < K; k++)
    for (int j = 0; j < nu; j++) C.i.push_back(k * nz + nx + j), C.p.push_back((int)C.i.size()), m++;
}

int main() {
  int checks = 0;
  Csr Q, A, C;
  int n, me, m;
  for (int mode = 0; mode < 2; mode++)
    for (int ordering = 0; ordering < 3; ordering++)
      for (int shards : {1, 3, 8}) {
        banded(700, 9, Q, A, C, me, m);
        for (int rank = 0; rank < shards; rank += shards > 1 ? shards - 1 : 1) {
          kktdev::Analysis an;
          an.ordering = ordering, an.shard_rank = rank, an.shard_count = shards;
          const int e = an.run(mode, 700, me, m, Q.p.data(), Q.i.data(), A.p.data(), A.i.data(), C.p.data(), C.i.data(), 0, 0, mode ? 0 : -1);
          if (e || an.nnodes < 1 || an.sbw < 1) return 10 + e;
          checks++;
        }
      }
  // narrow band (Prg_DID-like), small leaves / supernodes, amalgamation
  docp(60, 2, 1, 0, 2, true, Q, A, C, n, me, m);
  for (int amal = 0; amal < 2; amal++) {
    kktdev::Analysis an;
    an.amalgamation = amal != 0;
    if (an.run(1, n, me, m, Q.p.data(), Q.i.data(), A.p.data(), A.i.data(), C.p.data(), C.i.data(), 8, 16, 0)) return 30;
    checks++;
  }
  // STAGED plans
  struct Case {
    int K, nx, nu, path, fin;
    bool fixed0;
  };
  for (const Case &c : {Case{6, 5, 2, 0, 0, true}, Case{8, 6, 3, 1, 4, true}, Case{5, 4, 2, 0, 2, false}, Case{3, 40, 300, 30, 0, true},
                        Case{12, 130, 16, 0, 120, true}}) {
    docp(c.K, c.nx, c.nu, c.path, c.fin, c.fixed0, Q, A, C, n, me, m);
    for (int shards : {1, 2, 3}) {
      kktdev::StagedPlan P;
      P.shard_count = shards, P.shard_rank = shards - 1, P.sharded = shards > 1;
      const int e = P.run(n, me, m, Q.p.data(), Q.i.data(), A.p.data(), A.i.data(), C.p.data(), C.i.data());
      if (shards > 1 && (c.nx & 1)) {  // (a sharded system needs an even number of states per stage)
        if (e != 1) return 40;
        continue;
      }
      if (e || P.K != c.K || P.nk[1] != c.nx || P.mk[0] != c.nu || P.fixed_x0 != c.fixed0) return 41 + e;
      checks++;
    }
    // explicit stage sizes + dense hand-over: the dynamics rows are empty
    kktdev::StagedPlan D;
    D.given_nx.assign(c.K + 1, c.nx), D.given_nu.assign(c.K, c.nu), D.dense_dyn = true;
    std::vector<int> Ap2(me + 1, 0), Ai2;
    const int ndyn = c.K * c.nx;
    for (int r = ndyn; r < me; r++) {
      for (int p = A.p[r]; p < A.p[r + 1]; p++) Ai2.push_back(A.i[p]);
      Ap2[r + 1] = (int)Ai2.size();
    }
    if (D.run(n, me, m, Q.p.data(), Q.i.data(), Ap2.data(), Ai2.data(), C.p.data(), C.i.data())) return 50;
    checks++;
  }
  // not a staircase
  banded(200, 6, Q, A, C, me, m);
  {
    kktdev::StagedPlan P;
    if (P.run(200, me, m, Q.p.data(), Q.i.data(), A.p.data(), A.i.data(), C.p.data(), C.i.data()) != 6) return 60;
    checks++;
  }
  printf("sanitize_host ok: %d structures analysed\n", checks);
  return 0;
}
