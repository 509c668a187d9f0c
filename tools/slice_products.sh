#!/bin/bash
# The single-GPU pieces of DESIGN.md section 7's model (one system over P ranks, C4 stage width): the strip products on
# their own by shape (hqpkkt_debug_dgemm: plain launch rules, and the cut form forced), then what ONE rank of P runs per
# stage (tools/shard_pieces.py: null transport, per-kernel durations from rocprofv3).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
echo "## W strip [W_p | W_u] = V+ [F_p | F_u]: 5000 x (strip + 50) x 5000 for 2 / 4 / 8 ranks; launch rules of st_gemm"
python3 tools/dgemm_shapes.py 5000x2610x5000x0 5000x1330x5000x0 5000x690x5000x0 5000x640x5000x0 2>&1 | grep dgemm
echo "## the same with the cut form forced (HQPKKT_DGEMM_FORCE_SPLIT)"
HQPKKT_DGEMM_FORCE_SPLIT=1 python3 tools/dgemm_shapes.py 5000x2610x5000x0 5000x1330x5000x0 5000x690x5000x0 5000x640x5000x0 2>&1 | grep dgemm
echo "## halves by rows (the two-part exchange): 2560 x (strip + 50) x 5000"
python3 tools/dgemm_shapes.py 2560x2610x5000x0 2560x1330x5000x0 2560x690x5000x0 2>&1 | grep dgemm
echo "## rank-q update V = G_xx - Y'Rm (whole lower triangle, mirrored, by every rank): 5000 x 5000 x 100"
python3 tools/dgemm_shapes.py 5000x5000x100x1x1 2>&1 | grep dgemm
echo "## one rank's stage, per kernel"
bash tools/shard_pieces.sh 1:0 2:0 4:0 8:0 8:7
