#!/bin/bash
# The single-GPU pieces of DESIGN.md section 7's model (one system over P ranks, C4 stage width): the strip products on
# their own by shape (hqpkkt_debug_dgemm: plain launch rules, and the cut form forced), then what ONE rank of P runs per
# stage (tools/shard_pieces.py: null transport, per-kernel durations from rocprofv3).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
echo "## W strip W_p = V+ F_p: 5000 x strip x 5000 for 2 / 4 / 8 ranks; launch rules of st_gemm"
python3 tools/dgemm_shapes.py 5000x2560x5000x0 5000x1280x5000x0 5000x640x5000x0 2>&1 | grep dgemm
echo "## the same with the cut form forced (HQPKKT_DGEMM_FORCE_SPLIT)"
HQPKKT_DGEMM_FORCE_SPLIT=1 python3 tools/dgemm_shapes.py 5000x2560x5000x0 5000x1280x5000x0 5000x640x5000x0 2>&1 | grep dgemm
echo "## rank-q update V = G_xx - Y'Rm (whole lower triangle, mirrored, by every rank): 5000 x 5000 x 100"
python3 tools/dgemm_shapes.py 5000x5000x100x1x1 2>&1 | grep dgemm
echo "## one rank's stage, per kernel"
bash tools/shard_pieces.sh 1:0 2:0 2:1 4:0 4:3 8:0 8:7
# (the model from these pieces: python3 tools/shard_model.py <this output>)
