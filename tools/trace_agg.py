#!/usr/bin/env python
"""Aggregate a rocprofv3 kernel-trace CSV: per kernel (and per grid shape with --grids) dispatch count / average / total."""
import csv, collections, sys
path = sys.argv[1]
grids = "--grids" in sys.argv
steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 1
rows = list(csv.DictReader(open(path)))
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"].split("(")[0]
    key = n
    if grids:
        wx, wy, wz = int(r["Workgroup_Size_X"]), int(r["Workgroup_Size_Y"]), int(r["Workgroup_Size_Z"])
        key = (n, int(r["Grid_Size_X"]) // wx, int(r["Grid_Size_Y"]) // wy, int(r["Grid_Size_Z"]) // wz)
    agg.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
tot = sum(sum(v) for v in agg.values())
print(f"total {tot / 1000:.2f} ms over {len(rows)} dispatches; per step ({steps}): {tot / 1000 / steps:.3f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{str(k)[:110]:110s} n={len(v):5d} avg {sum(v) / len(v):8.1f} us  per-step {sum(v) / 1000 / steps:7.3f} ms  {100 * sum(v) / tot:5.1f}%")
