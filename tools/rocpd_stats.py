"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) into the per-kernel statistics table
(name, calls, total / average / min / max duration, share) -- same content as rocprofv3's kernel_stats.csv.
Usage: python tools/rocpd_stats.py results.db [out.csv]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % disp)]
scol = [r[1] for r in cur.execute("pragma table_info(%s)" % sym)]
namecol = "display_name" if "display_name" in scol else ("kernel_name" if "kernel_name" in scol else "name")
q = ("select s.%s, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
     "from %s d join %s s on d.kernel_id = s.id group by s.%s order by 3 desc" % (namecol, disp, sym, namecol))
rows = list(cur.execute(q))
tot = sum(r[2] for r in rows)
lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage"]
for r in rows:
    lines.append('"%s",%d,%d,%.1f,%d,%d,%.2f' % (r[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot))
out = "\n".join(lines)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
print(out[:6000])
