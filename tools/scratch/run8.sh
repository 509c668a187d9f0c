cd $GRAFT_REPO_ROOT
echo skip
timeout 600 python bench.py --gpus 2 --backend gloo --share-gpu --stages 20 --steps 3 --warmup 1 --no-ip 2>gpurun_out/run8.err | grep '^{' | tail -1 > gpurun_out/bench_2rank.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_2rank.json'))
print({k:d[k] for k in ['value','n_gpus','scaling','replicas_value','ms_per_step']})
print(json.dumps(d['shard'], indent=1)[:1500])
print(d['replicas'])
PY
tail -5 gpurun_out/run8.err
