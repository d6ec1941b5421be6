#!/usr/bin/env python
"""Per-kernel average / minimum duration from a rocprofv3 results database (the sqlite file rocprofv3 writes)."""
import glob, sqlite3, sys
db = sys.argv[1] if len(sys.argv) > 1 else None
if db is None or not db.endswith(".db"):
    db = glob.glob((db or ".") + "/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
q = (f"select s.kernel_name, count(*), avg(d.end-d.start)/1000.0, min(d.end-d.start)/1000.0 from {kd} d join {ks} s "
     f"on d.kernel_id=s.id group by s.kernel_name order by 3 desc")
for r in c.execute(q):
    print(f"{r[0][:90]:90s} n={r[1]:5d} avg={r[2]:9.1f}us min={r[3]:9.1f}us")
