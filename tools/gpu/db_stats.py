"""Per-kernel table from a rocprofv3 results database (rocpd sqlite): calls, average and total duration."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
q = ("select s.kernel_name, count(*), avg(d.end-d.start)/1e3, sum(d.end-d.start)/1e3 from %s d join %s s on d.kernel_id=s.id "
     "group by s.kernel_name order by 4 desc limit %d" % (kd, ks, int(sys.argv[2]) if len(sys.argv) > 2 else 12))
for r in cur.execute(q):
    print("%-100s %6d %10.1f us %12.1f us" % (r[0][:100], r[1], r[2], r[3]))
