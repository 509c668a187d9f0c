#!/bin/bash
# Round 6: the border kernels (k_panel_solve<16|32>, k_schur_update<1|2>) and the skewed LDS images of k_factor_blk
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06b; rm -rf $O; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_block.py tests/test_gpu_parity.py tests/test_gpu_solve_top.py -x -q -m gpu > $O/tests.txt 2>&1
tail -5 $O/tests.txt
for v in "768 768" "0 0" "100000 100000" "256 256" "2048 768" "768 2048"; do
  set -- $v
  echo "PS16_MAX=$1 SU1_MAX=$2" >> $O/c2_variants.txt
  HQPKKT_PS16_MAX=$1 HQPKKT_SU1_MAX=$2 timeout 300 python3 bench.py --workload c2 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(' ms_per_step %.4f residual %.3e' % (d['ms_per_step'], d.get('residual', float('nan'))), {k: round(v['ms_per_step'],4) for k,v in d['kernels'].items() if v['ms_per_step']})" >> $O/c2_variants.txt
done
cat $O/c2_variants.txt
timeout 120 python3 tools/block_time.py 2>&1 | grep -v amdgpu | head -8 > $O/block_time.txt; cat $O/block_time.txt
