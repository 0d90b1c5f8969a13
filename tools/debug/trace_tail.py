#!/usr/bin/env python3
"""Per-launch time line of the LAST n kernels of a rocprofv3 --kernel-trace database (start offset, duration, gap to the
previous kernel, us): where do the first updates of a short timed region lose time?  Usage: trace_tail.py trace_results.db [n]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ks = db.cursor().execute("select start,end,name from kernels order by start").fetchall()
ks = [k for k in ks if k[2].startswith(("s8r", "s32", "k_gemm_lds", "k_dw64", "k_cycle_open"))][-n:]
t0 = ks[0][0]
prev = None
for s, e, nm in ks:
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{(s - t0) / 1e3:9.2f} us  +{(e - s) / 1e3:6.2f}  gap {gap:5.2f}  {nm.split('(')[0]}")
    prev = e
upd = [(ks[i][0], ks[i + 1][1]) for i in range(0, len(ks) - 1, 2)]
print("per update (chain start -> optimizer end):", " ".join(f"{(b - a) / 1e3:.1f}" for a, b in upd))
print(f"first start -> last end: {(ks[-1][1] - ks[0][0]) / 1e3:.1f} us for {len(ks)} kernels")
