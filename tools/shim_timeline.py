"""Kernels and copies of one iteration of the reference's Mehrotra loop on our plugin (rocprofv3 results .db):
between two k_weights launches (one factorisation each)."""
import sqlite3
import sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')").fetchall()]
rows = [(n, s, e) for n, s, e in c.execute("select name, start, end from kernels order by start").fetchall()]
if "memory_copies" in tabs:
    try:
        rows += [("COPY " + str(n), s, e) for n, s, e in c.execute("select name, start, end from memory_copies order by start").fetchall()]
    except Exception as ex:  # noqa: BLE001
        print("copies:", ex)
rows.sort(key=lambda r: r[1])
idx = [i for i, r in enumerate(rows) if "k_weights" in r[0]]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
a, b = idx[which], idx[which + 1]
t0, prev = rows[a][1], rows[a][1]
busy = 0
for name, s, e in rows[a:b]:
    nm = name.split("(")[0].replace("kktdev::", "").replace("void ", "")[:44]
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev) / 1e3:7.1f}  {nm}")
    prev = max(prev, e)
    busy += e - s
print("iteration", (rows[b][1] - t0) / 1e3, "us, device busy", busy / 1e3, "us, items", b - a)
