"""Matrix-pipe utilisation per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
GRBM_GUI_ACTIVE pass: MFMA busy cycles / (4 SIMDs x busy cycles of the shader engines), per kernel.
usage: python tools/pmc_busy.py <dir of the pass>"""
import collections
import csv
import glob
import os
import sys

f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True))[-1]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    nm = r["Kernel_Name"].split("(")[0].replace("kktdev::", "")
    tot[nm][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[nm].add(r["Dispatch_Id"])
print("kernel,launches,SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CYCLES,GRBM_GUI_ACTIVE,mfma_busy_per_gui_active_per_4simd_per_cu")
for k, c in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)):
    mf, sq, gui = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("SQ_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 256 CUs x 4 SIMDs matrix pipes
    util = mf / (gui / 8.0 * 256 * 4) if gui else float("nan")
    print(f"{k},{len(cnt[k])},{mf:.0f},{sq:.0f},{gui:.0f},{util:.4f}")
