cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -12
HQPKKT_MAX_PIVOTS=160 timeout 300 python bench.py --workload c2 --steps 20 --warmup 3 --no-ip 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('cap 160: ms_per_step', round(d['ms_per_step'],3), 'levels', d['config'].get('tree_levels'), 'res', d['residual'])
print({k: round(v,3) for k,v in d['kernel_ms_per_step'].items() if v})
"
timeout 300 python bench.py --workload c2 --steps 20 --warmup 3 --no-ip 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('default: ms_per_step', round(d['ms_per_step'],3), 'levels', d['config'].get('tree_levels'), 'res', d['residual'])
print({k: round(v,3) for k,v in d['kernel_ms_per_step'].items() if v})
"
