#!/bin/bash
# per-kernel durations of what one rank of 2 / 4 / 8 runs per stage (C4 stage width), and the unsharded stage beside it
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pieces; rm -rf $O; mkdir -p $O
for cfg in ${@:-1:0 2:0 4:0 8:0 8:7}; do
  set -- ${cfg/:/ }
  echo "== ranks $1, rank $2 (K = 8 stages of 5000 states, 50 controls)"
  python3 tools/shard_pieces.py $1 $2 2>/dev/null | grep '^{'
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$1_$2 -- python3 tools/shard_pieces.py $1 $2 > /dev/null 2>&1
  python3 - $O/kt_$1_$2 <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"))[-1]
rows = [r for r in csv.DictReader(open(f)) if "at::native" not in r["Name"] and "rocclr" not in r["Name"]]
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:12]:
    print(f'   {r["Name"][:100]:100s} calls {int(r["Calls"]):5d}  avg {float(r["AverageNs"])/1e3:9.1f} us  total {float(r["TotalDurationNs"])/1e6:9.2f} ms')
PY
done
