#!/bin/bash
# FETCH_SIZE / TCC hit counters per kernel of a mid-size factorisation (rocprofv3 --pmc, kernel trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
nx=${1:-1000}
for ctr in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum"; do
  rm -rf /tmp/p
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/p -- python3 $R/tools/c4_bench.py 12 $nx 50 2 > /tmp/p.log 2>&1
  f=$(find /tmp/p -name '*counter_collection.csv' | head -1)
  echo "== $ctr"
  python3 - "$f" <<'PY'
import csv, sys, collections
if not sys.argv[1]:
    print("  (no counter file)"); sys.exit()
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "dgemm" in k:
        print("  ", k, {c: (len(x), round(sum(x) / len(x), 1)) for c, x in v.items()})
PY
done
