#!/bin/bash
# durations of the individual launches of one stage (rocprofv3 kernel trace, dispatch order)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
nx=${1:-1000}
rm -rf /tmp/p
rocprofv3 --kernel-trace --output-format csv -d /tmp/p -- python3 $R/tools/c4_bench.py 12 $nx 50 2 > /tmp/p.log 2>&1
f=$(find /tmp/p -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last factorisation: find the last run of launches that starts with k_weights
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "k_weights" in n]
start = idx[-1]
t0 = int(rows[start]["Start_Timestamp"])
prev_end = t0
for r in rows[start:start + 40]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("  +%8.1f us  gap %5.1f  dur %7.1f us  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:80]))
    prev_end = e
PY
