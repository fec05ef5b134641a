#!/usr/bin/env python
"""Timeline of one steady-state frame from a rocprofv3 kernel trace (rocpd sqlite): start/end of every dispatch relative to
the frame start, queue/stream, and the union/overlap of busy time.  usage: timeline.py results.db [frame_index]"""
import re, sqlite3, sys

def main(path, frame=40):
    con = sqlite3.connect(path)
    tables = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    T = lambda p: next(t for t in tables if t.startswith(p))
    disp, sym = T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol")
    scol = [r[1] for r in con.execute(f"pragma table_info({sym})")]
    namecol = "display_name" if "display_name" in scol else "kernel_name"
    dcol = [r[1] for r in con.execute(f"pragma table_info({disp})")]
    qcol = "queue_id" if "queue_id" in dcol else ("stream_id" if "stream_id" in dcol else "tid")
    rows = list(con.execute(f"select s.{namecol}, d.start, d.end, d.{qcol} from {disp} d join {sym} s on d.kernel_id = s.id order by d.start"))
    short = lambda n: re.sub(r"\(.*$", "", n).replace("void ", "")[:44]
    # frames are delimited by advance_kernel
    marks = [i for i, r in enumerate(rows) if "advance_kernel" in r[0]]
    if len(marks) < frame + 2:
        frame = len(marks) // 2
    # one iteration = from the end of one advance_kernel to the end of the next one that is at least 300 us later
    tA = rows[marks[frame]][2]
    nxt = next(i for i in marks if rows[i][2] > tA + 300e3)
    tB = rows[nxt][2]
    fr = [r for r in rows if r[1] >= tA and r[2] <= tB]
    t0 = tA
    ev = []
    for n, s, e, q in fr:
        print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{q}  {short(n)}")
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    busy = both = 0; depth = 0; last = ev[0][0]
    for t, d in ev:
        if depth >= 1: busy += t - last
        if depth >= 2: both += t - last
        depth += d; last = t
    span = max(r[2] for r in fr) - t0
    print(f"# frame span {span / 1e3:.1f} us, >=1 kernel running {busy / 1e3:.1f} us, >=2 running {both / 1e3:.1f} us, sum of durations {sum(r[2] - r[1] for r in fr) / 1e3:.1f} us")

if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
