#!/bin/bash
# 256 x 128 tiles (one workgroup of 16 wavefronts per CU, HQPKKT_DGEMM_TILE256 = LDS buffers) against the launch rules
# in use, full products: the W of a C4 stage, a square 8192, a multiple of the tile, ragged shapes (err: sampled entries
# against exact sums)
cd $GRAFT_REPO_ROOT
SH="5000x5050x5000x0 8192x8192x8192x0 5120x5120x5000x0 4096x5120x4096x0 3000x3050x3000x0 700x520x300x0 513x1100x65x0"
for rep in 1 2; do
echo "== default"; timeout 200 python3 tools/dgemm_shapes.py $SH 2>&1 | grep dgemm
echo "== 256 x 128, two buffers"; HQPKKT_DGEMM_TILE256=2 timeout 200 python3 tools/dgemm_shapes.py $SH 2>&1 | grep dgemm
echo "== 256 x 128, three buffers"; HQPKKT_DGEMM_TILE256=3 timeout 200 python3 tools/dgemm_shapes.py $SH 2>&1 | grep dgemm
done
