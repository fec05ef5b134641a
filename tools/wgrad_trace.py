"""Development aid: durations of consecutive wgrad_kernel dispatches from a rocprofv3 kernel trace (rocpd sqlite).
usage: python tools/wgrad_trace.py results.db [first] [count]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
tables = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
T = lambda p: next(t for t in tables if t.startswith(p))
disp, sym = T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol")
scol = [r[1] for r in con.execute(f"pragma table_info({sym})")]
namecol = "display_name" if "display_name" in scol else "kernel_name"
dcol = [r[1] for r in con.execute(f"pragma table_info({disp})")]
gx = "grid_size_x" if "grid_size_x" in dcol else ("grid_x" if "grid_x" in dcol else None)
q = f"select d.start, d.end - d.start{', d.' + gx if gx else ''} from {disp} d join {sym} s on d.kernel_id = s.id where s.{namecol} like '%wgrad_kernel%' order by d.start"
rows = list(con.execute(q))
first = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2 // 22 * 22
count = int(sys.argv[3]) if len(sys.argv) > 3 else 44
for i, r in enumerate(rows[first:first + count]):
    print(i, f"{r[1] / 1e3:8.1f} us", r[2] if len(r) > 2 else "")
