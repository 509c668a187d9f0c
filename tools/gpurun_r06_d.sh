#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06d; rm -rf $O; mkdir -p $O
for v in "0 0" "-1 -1" "16 10" "16 0" "0 10" "8 5" "32 20"; do
  set -- $v
  echo "XCD_PS=$1 XCD_SU=$2" >> $O/c2_variants.txt
  HQPKKT_XCD_PS=$1 HQPKKT_XCD_SU=$2 HQPKKT_SU1_MAX=256 timeout 300 python3 bench.py --workload c2 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(' ms_per_step %.4f residual %.3e' % (d['ms_per_step'], d.get('residual', float('nan'))), {k: round(v['ms_per_step'],4) for k,v in d['kernels'].items() if v['ms_per_step']})" >> $O/c2_variants.txt
done
cat $O/c2_variants.txt
