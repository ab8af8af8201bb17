"""Per-kernel totals of a rocprofv3 results database: python tools/prof_summary.py DB [launch divisor] [rows]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute(f"select s.kernel_name, count(*), sum(d.end-d.start)/1e6 from {kd} d join {ks} s on d.kernel_id=s.id group by 1 order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print("total ms %.2f (/%g = %.2f)" % (tot, div, tot / div))
for r in rows[:top]:
    print("%9.2f ms %7.1f launches %9.1f us/launch  %s" % (r[2] / div, r[1] / div, 1e3 * r[2] / r[1], r[0][:120]))
