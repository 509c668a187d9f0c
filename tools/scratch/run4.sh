cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_block.py -x -q 2>&1 | tail -3
for mp in 128 160 192; do
  echo "== HQPKKT_MAX_PIVOTS=$mp"
  HQPKKT_MAX_PIVOTS=$mp timeout 300 python bench.py --workload c2 --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('ms_per_step', round(d['ms_per_step'],3), 'levels', d['config'].get('tree_levels'), 'res', d['residual'], 'refine', d['refine_rounds'])
print({k: round(v,3) for k,v in d['kernel_ms_per_step'].items() if v})
"
done
