"""Per-kernel / per-grid durations out of a rocprofv3 results .db (rocpd sqlite)."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else None
rows = c.execute("select name, count(*), sum(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 14]:
    print(f"{r[0][:64]:64s} n={r[1]:6d} tot={r[2]/1e6:8.2f} ms avg={r[2]/r[1]/1e3:7.2f} us {100*r[2]/tot:5.1f}%")
if pat:
    for r in c.execute("select grid_x, workgroup_x, count(*), avg(end-start), min(end-start) from kernels where name like ? "
                       "group by grid_x order by grid_x desc", ("%" + pat + "%",)):
        print("  grid", r[0], "wg", r[1], "n", r[2], "avg us", round(r[3] / 1e3, 2), "min", round(r[4] / 1e3, 2))
