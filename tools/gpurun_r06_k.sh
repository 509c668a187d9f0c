#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06k; rm -rf $O; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_franke.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -q -m gpu -x > $O/tests.txt 2>&1
tail -4 $O/tests.txt
timeout 300 python3 tools/ip_profile.py 2000 2>&1 | grep -v amdgpu | head -3
timeout 300 python3 tools/ip_profile.py 33333 2>&1 | grep -v amdgpu | head -3
timeout 300 python3 bench.py --workload c2 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(' ms_per_step %.4f residual %.3e' % (d['ms_per_step'], d.get('residual', float('nan'))), {k: round(v['ms_per_step'],4) for k,v in d['kernels'].items() if v['ms_per_step']})"
