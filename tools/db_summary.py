"""Per-kernel average duration from a rocprofv3 results .db: python tools/db_summary.py FILE.db [N]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [x for x in t if "kernel_dispatch" in x][0]
ks = [x for x in t if "kernel_symbol" in x][0]
q = f"select s.kernel_name, count(*), avg(d.end-d.start), sum(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by 1 order by 4 desc"
for r in list(c.execute(q))[: int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    print(f"{r[0][:90]:90s} n={r[1]:5d} avg {r[2] / 1e3:8.1f} us")
