#!/bin/bash
# stamps inside the whole-tree launch of k_factor_diag_small on Prg_DID K = 2000: a leaf, a node of level 1, of level 5, the root
mkdir -p gpurun_out
out=gpurun_out/r06_fds_tree_stamps.txt
: > $out
for B in 0 700 1000 1022; do
  echo "== block $B of 1023" >> $out
  HQPKKT_LIB=$PWD/tools/_build/libstamps_fds_$B.so timeout 120 python3 tools/stamps_small.py 2000 2>&1 | grep -v amdgpu >> $out
done
cat $out
