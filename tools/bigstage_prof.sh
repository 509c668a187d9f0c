#!/bin/bash
# per-kernel durations of the big-stage cases (rocprofv3 kernel trace); output under gpurun_out/bigstage/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/bigstage; mkdir -p $O
for c in "200 100" "260 200" "520 512" "200 300 40"; do
  rm -rf /tmp/p
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p -- python3 $R/tools/bigstage_one.py $c > /tmp/p.log 2>&1
  f=$(find /tmp/p -name '*kernel_stats.csv' | head -1)
  echo "== $c" | tee -a $O/stats.txt
  if [ -n "$f" ]; then head -7 $f | cut -d, -f1-4 | cut -c1-150 | tee -a $O/stats.txt; else tail -5 /tmp/p.log; fi
done
