#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c; rm -rf $O; mkdir -p $O
HQPKKT_LIB=$PWD/hqp_amd/libhqpkkt_stamps.so python3 tools/stamps_ps.py 2>&1 | grep -v amdgpu | tee $O/ps_stamps.txt
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/tests.txt 2>&1
tail -3 $O/tests.txt
for v in "256" "768"; do
  echo "SU1_MAX=$v" >> $O/c2_variants.txt
  HQPKKT_SU1_MAX=$v timeout 300 python3 bench.py --workload c2 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(' ms_per_step %.4f residual %.3e' % (d['ms_per_step'], d.get('residual', float('nan'))), {k: round(v['ms_per_step'],4) for k,v in d['kernels'].items() if v['ms_per_step']})" >> $O/c2_variants.txt
done
cat $O/c2_variants.txt
bash tools/c2_trace.sh 160 2>&1 | grep -v amdgpu | grep "panel_solve\|schur"
