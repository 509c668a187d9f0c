#!/bin/bash
# per-kernel durations of one mid-size factorisation + solve (rocprofv3 kernel trace); optional env passes through
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for nx in ${SIZES:-1000 2000}; do
  rm -rf /tmp/p
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p -- python3 $R/tools/c4_bench.py 40 $nx 50 3 > /tmp/p.log 2>&1
  f=$(find /tmp/p -name '*kernel_stats.csv' | head -1)
  echo "== nx=$nx (40 stages, 4 factor+solve)"
  python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print("  %-70s calls %5s  avg %8.1f us  total %8.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
