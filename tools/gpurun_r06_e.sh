#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06e; rm -rf $O; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_block.py tests/test_gpu_parity.py -x -q -m gpu > $O/tests.txt 2>&1
tail -3 $O/tests.txt
HQPKKT_LIB=$PWD/hqp_amd/libhqpkkt_stamps.so timeout 120 python3 tools/stamps_fb.py 160 qd 2>&1 | grep -v amdgpu > $O/stamps_fb_160.txt; cat $O/stamps_fb_160.txt
for L in "" _s2 _s6 _s8; do
  echo "lib$L" >> $O/variants.txt
  HQPKKT_LIB=$PWD/hqp_amd/libhqpkkt$L.so timeout 120 python3 tools/block_time.py 2>&1 | grep "qd.*p=\(128\|160\)" | cut -c1-70 >> $O/variants.txt
  HQPKKT_LIB=$PWD/hqp_amd/libhqpkkt$L.so timeout 300 python3 bench.py --workload c2 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(' ms_per_step %.4f residual %.3e' % (d['ms_per_step'], d.get('residual', float('nan'))), {k: round(v['ms_per_step'],4) for k,v in d['kernels'].items() if v['ms_per_step']})" >> $O/variants.txt
done
cat $O/variants.txt
