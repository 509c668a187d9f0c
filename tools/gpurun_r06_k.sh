#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06k; rm -rf $O; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_franke.py tests/test_gpu_fuzz.py -q -m gpu -x > $O/tests.txt 2>&1
tail -8 $O/tests.txt
timeout 300 python3 tools/ip_profile.py 2000 2>&1 | grep -v amdgpu | head -3
HQPKKT_IP_THREE_READS=1 timeout 300 python3 tools/ip_profile.py 2000 2>&1 | grep -v amdgpu | head -3
timeout 300 python3 tools/ip_profile.py 33333 2>&1 | grep -v amdgpu | head -3
