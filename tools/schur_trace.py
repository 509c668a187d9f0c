"""Per-launch rate of the Schur-update kernels from a rocprofv3 --kernel-trace csv (tools/kkt_classes.py under the profiler):
python tools/schur_trace.py <kernel_trace.csv> [pivots per supernode, default 160]"""
import csv, sys
p = int(sys.argv[2]) if len(sys.argv) > 2 else 160
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "schur_update" in r["Kernel_Name"]]
half = len(rows) // 2  # (two factorisations: the second one is warm)
tot = {}
print("kernel workgroups ms TFLOP/s(tiles as if full)")
for r in rows[half:]:
    big = "big" in r["Kernel_Name"]
    wg = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
    ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    edge = 128 if big else 64
    tf = wg * edge * edge * p * 2 / (ms * 1e-3) / 1e12
    k = "big" if big else "small"
    tot[k] = tot.get(k, 0.0) + ms
    if ms > 0.3:
        print(k, wg, round(ms, 3), round(tf, 1))
print({k: round(v, 2) for k, v in tot.items()})
