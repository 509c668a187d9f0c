"""Kernel timeline of one Mehrotra iteration out of a rocprofv3 results .db: for each
kernel between two consecutive k_ip_rhs launches its start offset, duration and the gap
to the previous kernel (us)."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "k_ip_rhs" in r[0]]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
a, b = idx[which], idx[which + 1]
t0, prev = rows[a][1], rows[a][1]
busy = 0
for name, s, e in rows[a:b]:
    nm = name.split("(")[0].replace("kktdev::", "").replace("void ", "")[:40]
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev) / 1e3:7.1f}  {nm}")
    prev = e
    busy += e - s
print("iteration", (rows[b][1] - t0) / 1e3, "us, kernels busy", busy / 1e3, "us, launches", b - a)
