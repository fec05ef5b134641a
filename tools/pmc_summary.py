#!/usr/bin/env python
"""Per-kernel averages of rocprofv3 --pmc counters from a rocpd sqlite database.
usage: python tools/pmc_summary.py results.db [kernel-substring]"""
import sqlite3
import sys
from collections import defaultdict


def main(path, filt="", frames=0, all_kernels=False):
    con = sqlite3.connect(path)
    tables = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    T = lambda p: next(t for t in tables if t.startswith(p))
    disp, sym, pmc, info = T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol"), T("rocpd_pmc_event"), T("rocpd_info_pmc")
    scol = [r[1] for r in con.execute(f"pragma table_info({sym})")]
    namecol = "display_name" if "display_name" in scol else "kernel_name"
    pcols = [r[1] for r in con.execute(f"pragma table_info({pmc})")]
    icols = [r[1] for r in con.execute(f"pragma table_info({info})")]
    q = (f"select s.{namecol}, i.name, d.id, p.value, (d.end - d.start) from {pmc} p join {disp} d on p.event_id = d.event_id "
         f"join {sym} s on d.kernel_id = s.id join {info} i on p.pmc_id = i.id")
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(set)
    dur = defaultdict(dict)
    for name, cname, did, val, dt in con.execute(q):
        name = name.split("(")[0].replace("void ", "")
        if filt and filt not in name:
            continue
        acc[name][cname] += val
        cnt[name].add(did)
        dur[name][did] = dt
    for name in sorted(acc, key=lambda n: -sum(dur[n].values())):
        n = len(cnt[name])
        print(f"{name[:100]}  dispatches={n}  avg_us(profiled)={sum(dur[name].values())/n/1e3:.1f}")
        for c, v in sorted(acc[name].items()):
            print(f"    {c:32s} {v/n:16.1f}")
    if frames < 0:   # auto: one advance_kernel per frame in the one-chain schedule
        frames = max((len(cnt[n]) for n in cnt if "advance_kernel" in n), default=0)
    if frames:
        # the rollout loop's kernels run a whole number of times per frame; everything else (torch's zero-fills of the 1.1 GB of
        # output frames and the states at engine construction, the weight packers, the once-per-event static part of stage 1) is
        # setup and is listed apart -- dividing it by the few frames of a counter pass would bill it to the frame
        # (--all-kernels: a training step, where torch's copies / fills and the packers ARE part of every window)
        is_setup = lambda n: (not all_kernels) and (n.startswith("at::native") or "pack_" in n or "stage1_static" in n or len(cnt[n]) % frames != 0)
        loop = [n for n in acc if not is_setup(n)]
        setup = [n for n in acc if n not in loop]
        tot, tot_setup = defaultdict(float), defaultdict(float)
        for name in acc:
            for c, v in acc[name].items():
                (tot if name in loop else tot_setup)[c] += v
        t_all = sum(sum(dur[n].values()) for n in loop)
        print(f"# totals over the rollout loop's kernels, per frame ({frames} frames): kernel time {t_all/frames/1e3:.1f} us")
        for c, v in sorted(tot.items()):
            print(f"#   {c:32s} {v/frames:18.1f}")
        print(f"# setup kernels (once per engine / event, NOT in the per-frame totals): {', '.join(sorted(n[:40] for n in setup)) or 'none'}")
        for c, v in sorted(tot_setup.items()):
            print(f"#   {c + ' (whole run)':32s} {v:18.1f}")


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    frames = next((int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--frames=")), 0)
    main(args[0], args[1] if len(args) > 1 else "", frames, "--all-kernels" in sys.argv)
