#!/bin/bash
# LDS counters of the dominant kernel (rocprofv3 --pmc, kernel trace only): bank conflicts against active cycles
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for ctr in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES" ; do
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
        print("  ", k, {c: (len(x), "%.4g" % (sum(x) / len(x))) for c, x in v.items()})
PY
done
