#!/usr/bin/env python3
"""Per-kernel average of the FETCH_SIZE / WRITE_SIZE counters from two rocprofv3 sqlite outputs (one db per counter
pass, tools/gpu_pmc.sh) -> text table on stdout and, with --json PATH, the per-launch HBM traffic file bench.py reads.

  python tools/pmc_summary.py FETCH.db WRITE.db [--json profiles/rNN_pmc_traffic.json] [--note "..."]

MI355X_MICROARCH.md: both counters report KiB; on gfx950 FETCH_SIZE counts wide coalesced reads at half their
bytes (x2 correction applied in hbm_bytes_per_launch), WRITE_SIZE is uncalibrated."""
import glob
import hashlib
import json
import os
import sqlite3
import sys


_COMMENT_OR_STRING = None


def _code_only(text):
    """The source without comments and with runs of white space collapsed: editing a comment must not make committed counter
    files read as stale."""
    global _COMMENT_OR_STRING
    import re
    if _COMMENT_OR_STRING is None:
        _COMMENT_OR_STRING = re.compile(r'//[^\n]*|/\*.*?\*/|"(?:\\.|[^"\\])*"', re.S)
    text = _COMMENT_OR_STRING.sub(lambda m: m.group(0) if m.group(0).startswith('"') else " ", text)
    return " ".join(text.split())


def csrc_sha16():
    """Fingerprint of the kernel sources the counters were taken on, comments and white space aside (bench.py recomputes it and
    marks the traffic figure stale when the code has changed since)."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rl_arm_under_sparse_reward_amd", "csrc")
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.h")) + glob.glob(os.path.join(root, "*.inc"))):
        h.update(os.path.basename(path).encode())
        with open(path, "r", encoding="utf-8", errors="replace") as f:
            h.update(_code_only(f.read()).encode())
    return h.hexdigest()[:16]


def per_kernel(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    if "counters_collection" not in tabs:
        raise SystemExit(f"{path}: no counters_collection view; tables: {tabs[:10]}")
    cols = [d[0] for d in cur.execute("select * from counters_collection limit 1").description]
    name_col = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else cols[0])
    rows = cur.execute(f"select {name_col}, counter_name, count(*), avg(value), sum(value) from counters_collection "
                       f"group by {name_col}, counter_name order by 5 desc").fetchall()
    return rows


def main():
    args = sys.argv[1:]
    out_json, note = None, ""
    if "--json" in args:
        i = args.index("--json"); out_json = args[i + 1]; del args[i:i + 2]
    if "--note" in args:
        i = args.index("--note"); note = args[i + 1]; del args[i:i + 2]
    table = {}
    for path in args:
        print(path)
        for name, counter, n, avg, _ in per_kernel(path)[:14]:
            # FETCH_SIZE / WRITE_SIZE count KiB (the x2 fetch correction is applied further down); every other counter is a plain count
            unit = "KiB/launch" if counter in ("FETCH_SIZE", "WRITE_SIZE") else "per launch"
            print(f"  {str(name)[:50]:50s} {counter:12s} n={n:6d} avg={avg:12.1f} {unit}")
            short = str(name).strip()
            short = (short[5:] if short.startswith("void ") else short).split("(")[0].split("::")[-1]   # "void s8r4::k_fb_split8<0>(...)" -> "k_fb_split8<0>"
            table.setdefault(short, {})[counter] = avg
    if out_json:
        kernels = {}
        for k, v in table.items():
            if "FETCH_SIZE" in v and "WRITE_SIZE" in v and k.startswith("k_"):
                kernels[k] = {"FETCH_SIZE_KiB": round(v["FETCH_SIZE"], 1), "WRITE_SIZE_KiB": round(v["WRITE_SIZE"], 1),
                              "hbm_bytes_per_launch": int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)}
        with open(out_json, "w") as f:
            json.dump({"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes, tools/gpu_pmc.sh) "
                                 "-- python bench.py --steps 200 --warmup 40 --no-cpu-baseline --no-profile; " + note,
                       "units": "KiB per launch as reported; MI355X_MICROARCH.md: FETCH_SIZE counts wide coalesced reads at "
                                "half their bytes on gfx950 (x2 correction applied in hbm_bytes_per_launch), WRITE_SIZE "
                                "uncalibrated", "csrc_sha16": csrc_sha16(), "kernels": kernels}, f, indent=1)
        print("wrote", out_json)


if __name__ == "__main__":
    main()
