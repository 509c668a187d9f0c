#!/bin/bash
# Round 6, first collection: where a 16-pivot panel of k_factor_blk spends its time (stamps), baselines on this box.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06a; rm -rf $O; mkdir -p $O
HQPKKT_LIB=$PWD/hqp_amd/libhqpkkt_stamps.so timeout 120 python3 tools/stamps_fb.py 160 qd > $O/stamps_fb_160.txt 2>&1
HQPKKT_LIB=$PWD/hqp_amd/libhqpkkt_stamps.so timeout 120 python3 tools/stamps_fb.py 128 qd > $O/stamps_fb_128.txt 2>&1
timeout 300 python3 tools/block_time.py > $O/block_time.txt 2>&1
timeout 300 python3 bench.py --workload c2 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $O/bench_c2.json
timeout 300 python3 tools/ip_profile.py 2000 > $O/ip_did.txt 2>&1
cat $O/stamps_fb_160.txt
tail -3 $O/ip_did.txt
