#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace sqlite database: per-kernel count / avg / share, plus the
per-launch sequence of one update step (durations in us).  Usage: trace_summary.py trace_results.db"""
import sqlite3
import sys


def short_kernel(name):
    """'void s8r4::k_fb_split8<0>(unsigned long long, ...)' -> 'k_fb_split8<0>': no return type, namespace or argument list
    (what csrc's launch log -- hp_agent_update_kernels -- calls the kernel, and what bench.py looks a committed average up by;
    bench.py carries the same five lines)."""
    n = name.strip()
    if n.startswith("void "):
        n = n[5:]
    return n.split("(")[0].split("::")[-1]


db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
# the split launch's prologue (target chains of a sequence's first update only: a much smaller grid) is listed on its own row
rows = cur.execute("select case when name like '%k_fb_split8%' and 2 * grid_x < (select max(grid_x) from kernels k2 where k2.name = kernels.name) "
                   "then name || '[prologue: target chains only]' else name end as nm, count(*), avg(end-start), min(end-start), "
                   "max(end-start), sum(end-start) from kernels group by nm order by 6 desc").fetchall()
tot = sum(r[5] for r in rows)
cmd = sys.argv[2] if len(sys.argv) > 2 else "python bench.py --steps 2000 --warmup 400 --no-cpu-baseline --no-profile"
print(f"# rocprofv3 --kernel-trace --stats -d <dir> -o trace -- {cmd}   (kernel durations from trace_results.db, "
      "tools/trace_summary.py)")
try:   # fingerprint of the kernel sources this trace was taken on (bench.py: roofline.duration_stale)
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from pmc_summary import csrc_sha16
    print("# csrc_sha16", csrc_sha16())
except Exception:   # noqa: BLE001
    pass
print(f"{'kernel':44s} {'calls':>7s} {'avg_us':>8s} {'min_us':>8s} {'max_us':>8s} {'share':>6s}")
for r in rows:
    nm = short_kernel(r[0].split("[prologue")[0]) + ("[prologue]" if "[prologue" in r[0] else "")   # (first column: no spaces)
    print(f"{nm[:44]:44s} {r[1]:7d} {r[2]/1e3:8.2f} {r[3]/1e3:8.2f} {r[4]/1e3:8.2f} {100*r[5]/tot:5.1f}%")
ks = cur.execute("select start,end,name,grid_x from kernels order by start").fetchall()
g = [i for i, k in enumerate(ks) if short_kernel(k[2]).startswith(("k_fb_split8", "k_fb_slab"))]
if len(g) > 4:
    i0 = g[len(g) // 2]
    i1 = g[len(g) // 2 + 1]
    seq = ks[i0:i1]
    print("one update step:", " ".join(f"{short_kernel(k[2])}:{(k[1]-k[0])/1e3:.1f}" for k in seq),
          f"| total {(seq[-1][1]-seq[0][0])/1e3:.1f} us, {len(seq)} launches")
