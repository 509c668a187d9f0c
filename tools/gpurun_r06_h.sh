#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06h; rm -rf $O; mkdir -p $O
for L in own_stamps skipupd_stamps; do echo "== $L"; HQPKKT_LIB=$PWD/hqp_amd/libhqpkkt_$L.so timeout 120 python3 tools/stamps_fb.py 160 qd 2>&1 | grep -v amdgpu | tail -16; done
for L in "" _own; do
  echo "lib$L"
  HQPKKT_LIB=$PWD/hqp_amd/libhqpkkt$L.so timeout 120 python3 tools/block_time.py 2>&1 | grep "qd.*p=\(128\|150\|160\)" | cut -c1-70
  HQPKKT_LIB=$PWD/hqp_amd/libhqpkkt$L.so timeout 300 python3 bench.py --workload c2 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(' ms_per_step %.4f residual %.3e' % (d['ms_per_step'], d.get('residual', float('nan'))), {k: round(v['ms_per_step'],4) for k,v in d['kernels'].items() if v['ms_per_step']})"
done
