"""One case of tools/fuzz.py under variations of the tree / pivot options, to localise a failure.
Usage: python tools/fuzz_case.py <case> [...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import fuzz
from hqp_amd import ipmatrix
from oracle import oracleapi

for case in map(int, sys.argv[1:]):
    prog, st, kind, kw, tag = fuzz.make_case(case)
    O = oracleapi.OracleIpMatrix(kind)
    O.init(prog)
    O.factor(st[0], st[1])
    print(tag, "oracle res", O.solve(*st)[1], flush=True)
    variants = [dict(kw)]
    for extra in (dict(zd_policy=0), dict(zd_policy=2), dict(small_fronts=False), dict(slack_policy=0),
                  dict(leaf_size=12), dict(leaf_size=16), dict(leaf_size=24), dict(max_pivots=128),
                  dict(amalgamation=False)):
        v = dict(kw)
        v.update(extra)
        if v not in variants:
            variants.append(v)
    for v in variants:
        try:
            M = fuzz.CLS[kind](**v)
            M.init(prog)
            M.factor(prog, st[0], st[1])
            d = [np.zeros(k) for k in (prog.n, prog.me, prog.m, prog.m)]
            M.step(prog, *st, *d)
            r1 = M.residuum(prog, *st, *d)
            res = M.solve(prog, *st, *d)
            s = M.stats()
            print("   ", v, "step res %.3g solve res %.3g" % (r1, res),
                  {k: s[k] for k in ("n_supernodes", "n_levels", "max_front", "n_2x2", "n_perturbed", "refine_rounds", "kmax")}, flush=True)
        except Exception as e:
            print("   ", v, "FAILED", repr(e)[:80], flush=True)
