#!/usr/bin/env python
"""Timeline of the kernel chains of a few steady-state frames from a rocprofv3 kernel trace (rocpd sqlite): which chain is the
critical one of the three-chain schedule, and how much each launch is stretched there.

usage: python tools/chain_timeline.py <results.db> [window_us=2400] [> gpurun_out/timeline.txt]
Prints every dispatch that starts inside a window in the middle of the LAST long run of dispatches: start offset, duration, the gap
to the previous dispatch of the same queue, queue / stream id, kernel.  Then per queue: busy time and idle time inside the window.
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name).replace("void ", "")
    name = name.replace("conv_gemm_kernel", "gemm").replace("small_cell_gemm_kernel", "small_cell")
    return name[:60]


def main(path, window_us=2400.0):
    con = sqlite3.connect(path)
    tables = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tables if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tables if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in con.execute(f"pragma table_info({disp})")]
    scol = [r[1] for r in con.execute(f"pragma table_info({sym})")]
    namecol = "display_name" if "display_name" in scol else ("kernel_name" if "kernel_name" in scol else "name")
    qcols = [c for c in ("queue_id", "stream_id") if c in cols]
    print("# dispatch columns:", ",".join(cols))
    sel = ", ".join("d." + c for c in qcols) if qcols else "0"
    rows = list(con.execute(f"select d.start, d.end, s.{namecol}, {sel} from {disp} d join {sym} s on d.kernel_id = s.id order by d.start"))
    if not rows:
        print("no dispatches")
        return
    t_end = rows[-1][1]
    # the middle of the last 40 % of the trace (the timed region / long run of the bench)
    t0 = rows[0][0]
    mid = t0 + (t_end - t0) * 0.8
    w0, w1 = mid, mid + window_us * 1e3
    sel_rows = [r for r in rows if w0 <= r[0] < w1]
    print(f"# window {window_us:.0f} us at {(w0 - t0) / 1e6:.1f} ms of {(t_end - t0) / 1e6:.1f} ms; {len(sel_rows)} dispatches; queue key = {qcols}")
    last_end = {}
    busy = {}
    print(f"{'start_us':>9s} {'dur_us':>8s} {'gap_us':>8s}  {'queue':12s} kernel")
    for st, en, name, *q in sel_rows:
        key = "/".join(str(x) for x in q)
        gap = (st - last_end[key]) / 1e3 if key in last_end else float("nan")
        last_end[key] = en
        busy[key] = busy.get(key, 0.0) + (en - st) / 1e3
        print(f"{(st - w0) / 1e3:9.1f} {(en - st) / 1e3:8.1f} {gap:8.1f}  {key:12s} {short(name)}")
    print("# per queue: busy us inside the window")
    for k, v in sorted(busy.items()):
        print(f"#   {k:12s} {v:9.1f} us busy of {window_us:.0f}")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 2400.0)
