#!/bin/bash
# wave-level counters of the dominant kernel (rocprofv3 --pmc, kernel trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for ctr in "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA" "SQC_ICACHE_REQ SQC_ICACHE_MISSES" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  rm -rf /tmp/p
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/p -- python3 $R/bench.py --stages 20 --steps 1 --warmup 1 --no-cpu-baseline --no-ip > /tmp/p.log 2>&1
  f=$(find /tmp/p -name '*counter_collection.csv' | head -1)
  echo "== $ctr"
  python3 - "$f" <<'PY'
import csv, sys, collections
if not sys.argv[1]:
    print("  (no counter file)"); sys.exit()
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "dgemm_tn_sk" in k:
        print("  ", {c: (len(x), "%.4g" % (sum(x) / len(x))) for c, x in v.items()})
PY
done
