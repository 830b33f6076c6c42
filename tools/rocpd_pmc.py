"""Per-kernel PMC counter averages from a rocprofv3 rocpd database (developer tool).
Usage: python tools/rocpd_pmc.py results.db"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
T = lambda p: [t for t in tabs if t.startswith(p)][0]
disp, sym, pe, ip = T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol"), T("rocpd_pmc_event"), T("rocpd_info_pmc")
scol = [r[1] for r in cur.execute("pragma table_info(%s)" % sym)]
namecol = "display_name" if "display_name" in scol else "kernel_name"
dcols = [r[1] for r in cur.execute("pragma table_info(%s)" % disp)]
q = ("select d.id, s.%s, d.end-d.start, p.name, e.value from %s e join %s p on e.pmc_id=p.id "
     "join %s d on e.event_id=d.event_id join %s s on d.kernel_id=s.id" % (namecol, pe, ip, disp, sym))
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(set)
dur = defaultdict(float)
seen = set()
for did, name, d, pname, val in cur.execute(q):
    acc[name][pname] += val
    cnt[name].add(did)
    if did not in seen:
        seen.add(did)
        dur[name] += d
names = sorted(acc, key=lambda n: -dur[n])
for n in names[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    k = len(cnt[n])
    print("%s  calls=%d avg_us=%.1f" % (n[:90], k, dur[n] / k / 1e3))
    for pname in sorted(acc[n]):
        print("    %-28s %14.0f" % (pname, acc[n][pname] / k))
