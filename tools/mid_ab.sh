#!/bin/bash
# same-box A/B of HQPKKT_MID_SPLIT (1: 128 x 128 tiles cut over one workgroup per CU for 129 .. 383 tiles; 2: the
# triangular products of the split form on one workgroup per CU): factorisation time of K stages at several widths
cd $GRAFT_REPO_ROOT
run() { python tools/c4_bench.py $1 $2 50 3 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  nx', d['nx'], 'factor ms %.2f' % d['ms_factor_dev'], 'TF %.1f' % d['tflops_factor'], 'res %.1e' % d['res'])"; }
for rep in 1 2; do
  for mode in 0 1; do
    echo "== HQPKKT_MID_SPLIT=$mode"
    for nx in 1500 1800 2000 2300 2500 2700; do HQPKKT_MID_SPLIT=$mode run 60 $nx; done
  done
  for mode in 0 2; do
    echo "== HQPKKT_MID_SPLIT=$mode (C4 width, 30 stages)"
    HQPKKT_MID_SPLIT=$mode run 30 5000
    HQPKKT_MID_SPLIT=$mode run 60 3000
  done
done
