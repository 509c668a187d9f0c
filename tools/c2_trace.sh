#!/bin/bash
# Kernel timeline of one C2 factor + solve (rocprofv3 kernel trace of the eager profile pass of the bench).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/c2trace; rm -rf $O; mkdir -p $O
HQPKKT_MAX_PIVOTS=${1:-160} rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 bench.py --workload c2 --steps 2 --warmup 1 --no-cpu-baseline --no-ip > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob('gpurun_out/c2trace/kt/*/*kernel_trace.csv'))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last factorisation: from the last k_weights on
idx = [i for i, r in enumerate(rows) if 'k_weights' in r['Kernel_Name']]
i0 = idx[-1]
t0 = int(rows[i0]['Start_Timestamp'])
prev = t0
for r in rows[i0:i0 + 120]:
    name = r['Kernel_Name'].split('(')[0].replace('kktdev::', '')[:44]
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s - t0) / 1e3:9.1f} us  +{(s - prev) / 1e3:6.1f} gap  {(e - s) / 1e3:7.1f} us  grid {r.get('Grid_Size', r.get('Grid_Size_X', '?')):>8}  {name}")
    prev = e
PY
