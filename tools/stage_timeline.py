"""One stage of a factorisation out of a rocprofv3 kernel trace: every launch with its start (us after the stage's
first), duration, queue.  python tools/stage_timeline.py <dir with *_kernel_trace.csv> [skip launches]"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[-1]
rows = [r for r in csv.DictReader(open(f)) if "at::native" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last factorisation: find the last k_clear and print from the second stage on
idx = [i for i, r in enumerate(rows) if "k_clear" in r["Kernel_Name"]]
i0 = idx[-1]
seg = rows[i0:]
t0 = int(seg[0]["Start_Timestamp"])
n = 0
for r in seg[:int(sys.argv[2]) if len(sys.argv) > 2 else 70]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f'{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  q{r.get("Queue_Id", "?"):>3}  {r["Kernel_Name"][:80]}')
