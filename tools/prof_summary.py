#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace: per-kernel calls / total / average /
min / max duration and share of GPU time -- the same table `rocprofv3 --stats` prints in CSV mode.

usage: python tools/prof_summary.py gpurun_out/prof1/r01_results.db [> profiles/r01_kernel_stats.txt]
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "")
    return name[:110]


def main(path):
    con = sqlite3.connect(path)
    tables = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tables if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tables if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in con.execute(f"pragma table_info({disp})")]
    scol = [r[1] for r in con.execute(f"pragma table_info({sym})")]
    namecol = "display_name" if "display_name" in scol else ("kernel_name" if "kernel_name" in scol else "name")
    q = (f"select s.{namecol}, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
         f"from {disp} d join {sym} s on d.kernel_id = s.id group by s.{namecol} order by 3 desc")
    rows = list(con.execute(q))
    total = sum(r[2] for r in rows)
    print(f"# source: {path}")
    print(f"# total kernel time {total/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    print(f"{'kernel':112s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
    for name, n, tot, mn, mx in rows:
        print(f"{short(name):112s} {n:7d} {tot/1e6:10.3f} {tot/n/1e3:9.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100.0*tot/total:6.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
