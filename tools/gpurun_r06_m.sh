#!/bin/bash
# the cut form by the table (unequal shares): the products alone, stamps, then the bench line with and without
mkdir -p gpurun_out
out=gpurun_out/r06_sk_table.txt
: > $out
for t in 0 1 0 1; do
  echo "== HQPKKT_SK_TABLE=$t" >> $out
  HQPKKT_SK_TABLE=$t python3 tools/dgemm_stamps.py 5000x5050x5000x0 5050x5050x5000x1 5000x5000x5000x1 3000x3050x3000x0 2000x2050x2000x0 >> $out 2>&1
done
echo "== stamps" >> $out
HQPKKT_DGEMM_STAMPS=1 python3 tools/dgemm_stamps.py 5000x5050x5000x0 5050x5050x5000x1 >> $out 2>&1
python3 -m pytest tests/test_gpu_staged.py -q -x 2>&1 | tail -3 >> $out
HQPKKT_SK_TABLE=0 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-ip > gpurun_out/r06_bench_sk0.json 2>> $out
python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-ip > gpurun_out/r06_bench_sk1.json 2>> $out
python3 - >> $out <<'PY'
import json
for f in ("gpurun_out/r06_bench_sk0.json", "gpurun_out/r06_bench_sk1.json"):
    d = json.loads(open(f).read().strip().split("\n")[-1])
    print(f, d["value"], d["ms_per_step"], d["ms_factor"], d["ms_solve"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["residual"])
PY
grep -v "^  wg" $out | tail -40
