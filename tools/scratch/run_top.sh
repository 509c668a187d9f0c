#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_solve_top.py -x -q 2>&1 | tail -15
for v in "" "HQPKKT_MAX_PIVOTS=160" "HQPKKT_NO_SOLVE_TOP=1"; do
  echo "== c2 $v"
  env $v timeout 300 python bench.py --workload c2 --steps 30 --warmup 5 --no-cpu-baseline --no-ip 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], 'residual', d.get('residual'), {k: round(v['ms_per_step'], 3) for k, v in d['kernels'].items()} if 'kernels' in d else d.keys())"
done
timeout 300 python tools/scratch/top_info.py 2>&1 | grep -v "amdgpu.ids\|  level" | tail -4
