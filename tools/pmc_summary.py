#!/usr/bin/env python3
"""Per-kernel average of a PMC counter from rocprofv3 sqlite outputs (one db per counter pass)."""
import sqlite3
import sys

for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    view = "counters_collection" if "counters_collection" in tabs else None
    if not view:
        print(path, "no counters_collection view; tables:", tabs[:10])
        continue
    cols = [d[0] for d in cur.execute(f"select * from {view} limit 1").description]
    name_col = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else cols[0])
    rows = cur.execute(f"select {name_col}, counter_name, count(*), avg(value), sum(value) from {view} "
                       f"group by {name_col}, counter_name order by 5 desc").fetchall()
    print(path)
    for r in rows[:12]:
        print(f"  {str(r[0])[:50]:50s} {r[1]:12s} n={r[2]:6d} avg={r[3]:12.1f}")
