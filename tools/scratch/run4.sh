cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_block.py -x -q 2>&1 | tail -8
for mp in 128 160 192; do
  echo "== HQPKKT_MAX_PIVOTS=$mp"
  HQPKKT_MAX_PIVOTS=$mp timeout 300 python bench.py --workload c2 --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('ms_per_step', d['ms_per_step'], 'levels', d['config'].get('tree_levels'), 'res', d['residual'], 'refine', d['refine_rounds'], 'slow', d.get('n_slow_pivots'))
print({k: round(v,3) for k,v in d['kernel_ms_per_step'].items() if v})
print({k: v for k,v in d['kernel_launches_per_step'].items() if v})
"
done
echo "== old kernel"
HQPKKT_OLD_FD=1 timeout 300 python bench.py --workload c2 --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('ms_per_step', d['ms_per_step'], 'levels', d['config'].get('tree_levels'))
print({k: round(v,3) for k,v in d['kernel_ms_per_step'].items() if v})
"
