#!/usr/bin/env python3
"""Per-launch start / duration / gap of the last kernels of a rocprofv3 --kernel-trace database (steady state of a bench run)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ks = db.execute("select start,end,name,grid_x from kernels order by start").fetchall()
ks = ks[len(ks) // 2: len(ks) // 2 + n]
prev = None
for s, e, nm, g in ks:
    print(f"{nm.split('(')[0][:28]:28s} grid {g:6d} dur {(e - s) / 1e3:7.2f} us  gap before {((s - prev) / 1e3 if prev else 0):6.2f} us")
    prev = e
