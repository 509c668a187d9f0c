#!/bin/bash
# Wave-level and LDS counters of the tree engine's kernels on the C2 system (rocprofv3 --pmc in separate passes, kernel
# trace only): per kernel the mean per launch over the launches of three factor + solve steps.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/p
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/p -- python3 $R/bench.py --workload c2 --steps 2 --warmup 1 --no-cpu-baseline --no-ip > /tmp/p.log 2>&1
  f=$(find /tmp/p -name '*counter_collection.csv' | head -1)
  echo "== $ctr"
  python3 - "$f" <<'PY'
import csv, sys, collections
if not sys.argv[1]:
    print("  (no counter file)"); sys.exit()
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("kktdev::", "").replace("void ", "")
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if any(s in k for s in ("k_factor_blk", "k_panel_solve", "k_schur_update", "k_solve_top", "k_assemble_simple", "k_residual")):
        print("  %-44s" % k[:44], {c: (len(x), "%.4g" % (sum(x) / len(x))) for c, x in acc[k].items()})
PY
done
