#!/usr/bin/env python
"""Phase timeline of ONE cooperative cell launch from s_memtime stamps (tuning build with -DURNN_TRACE):
    python tools/build_variants.py trace; URNN_LIB=u-rnn_amd/liburnn_hip_trace.so python tools/trace_coop.py enc3 [dec3 ...]
Stamps (urnn_small.hip COOP_STAMP): 0 entry | 1 panel loaded + split | 2 block barrier | 3 gate k-loop done | 4 statistics written, h requested
| 5 grid barrier 1 passed | 6 gate statistics folded | 7 gated, r.h in the panel | 8 block barrier | 9 candidate k-loop + statistics
| 10 grid barrier 2 passed | 11 candidate statistics folded | 12 blended, stores drained."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from urnn_amd import ops, _lib
from urnn_amd.rollout import RolloutEngine
import urnn_amd.weights as uw

NAMES_TILES = ["setup: tiles, bias, first DMA", "k-loop (phase A)", "gate stats + W2h request", "GRID BARRIER 1", "fold gates + sync", "r.h in place + sync",
               "cand h-groups | z sigmoid", "GRID BARRIER 2 (+ z hand-over)", "fold cand + sync", "blend + store drain"]
NAMES = ["panel load+split", "block barrier", "gate k-loop", "stats + h request", "GRID BARRIER 1", "fold gates", "gate + r.h -> panel", "block barrier",
         "cand k-loop + stats", "GRID BARRIER 2", "fold cand", "blend + store drain"]


def main():
    which = sys.argv[1:] or ["enc3", "dec3"]
    cfg_name = os.environ.get("TRACE_CONFIG", "location1")
    H, W, nums, T, rain_max, cum_max, spatial = bench.CONFIGS[cfg_name]
    dev = torch.device("cuda:0")
    net, sd, cfg = bench.build_net(H, W, 2 * nums + 3, dev)
    eng = RolloutEngine(net, H, W, nums, rain_max, cum_max, max_frames=8, net_cfg=cfg, use_graph=False, spatial_rain=spatial, device=dev)
    eng.load_event(uw.make_event(8, H, W, rain_max, seed=42, spatial_rain=spatial))
    eng.reset()
    eng.run(2)
    e1, e2, e3, d1, d2, d3 = eng.states
    cells = {"enc1": (net.encoder.rnn1, eng.a1, None, e1), "dec1": (net.decoder.rnn1, eng.u2, e1, d3), "enc2": (net.encoder.rnn2, eng.a2, None, e2),
             "dec2": (net.decoder.rnn2, eng.u3, e2, d2), "enc3": (net.encoder.rnn3, eng.a3, None, e3), "dec3": (net.decoder.rnn3, None, e3, d1)}
    L = _lib.lib()
    setters = [getattr(L, n, None) for n in ("urnn_debug_set_trace_urnn_small", "urnn_debug_set_trace_urnn_coop_tiles")]
    assert all(f is not None for f in setters), "build the trace library first: python tools/build_variants.py trace; URNN_LIB=u-rnn_amd/liburnn_hip_trace.so"
    for f in setters:
        f.argtypes = [ctypes.c_void_p]

    def setter(p):
        return max(f(p) for f in setters)
    for name in which:
        cell, x, e, h = cells[name]
        tmp = h.clone()
        ws = ops.workspace(ops.gru_cell_workspace_bytes(*h.shape), dev)
        flags = ops.PHASE_ALL | ops.PHASE_COOP
        for _ in range(3):
            cell.step(x, e, h, out=tmp, phases=flags, ws=ws)
        # launch duration without stamps
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            cell.step(x, e, h, out=tmp, phases=flags, ws=ws)
        b.record()
        b.synchronize()
        us = a.elapsed_time(b) * 1e3 / 20
        nblk = 1024
        buf = torch.zeros(nblk * 16 * 16, dtype=torch.int64, device=dev)
        assert setter(buf.data_ptr()) == 0
        torch.cuda.synchronize()
        cell.step(x, e, h, out=tmp, phases=flags, ws=ws)
        torch.cuda.synchronize()
        setter(0)
        t = buf.cpu().numpy().reshape(nblk, 16, 16).astype(np.float64)
        used = t[:, :, 0] > 0
        nb = int(used[:, 0].sum())
        nw = int(used[0].sum())
        # s_memtime is a 100 MHz counter shared by the chip: reference everything to the earliest entry stamp
        tiles_kernel = name in ("enc2", "dec2")      # (half resolution: urnn_coop_tiles.hip, 11 stamps)
        names, ns = (NAMES_TILES, 11) if tiles_kernel else (NAMES, 13)
        # s_memtime counts shader cycles, per XCD: reference every block to the earliest entry stamp of its own XCD (block % 8)
        tb = t.copy()
        for xc in range(8):
            sel = used.copy()
            sel[np.arange(nblk) % 8 != xc] = False
            if sel.any():
                tb[np.arange(nblk) % 8 == xc] -= t[sel][:, 0].min()
        tt = tb[used][:, :ns] / 100.0                # units of 100 cycles (~0.043 us at 2.3 GHz)
        print(f"\n{name}: {nb} blocks x {nw} waves; launch {us:.1f} us (events, 20 launches back to back); stamped span {tt[:, ns - 1].max():.1f} x 100 cycles")
        print(f"{'phase (units of 100 cycles)':34s} {'mean':>7s} {'p10':>7s} {'p50':>7s} {'p90':>7s} {'max':>7s}   ends at (mean / max)")
        for k, nm in enumerate(names):
            d = tt[:, k + 1] - tt[:, k]
            print(f"{nm:34s} {d.mean():7.2f} {np.percentile(d, 10):7.2f} {np.percentile(d, 50):7.2f} {np.percentile(d, 90):7.2f} {d.max():7.2f}   {tt[:, k + 1].mean():7.2f} / {tt[:, k + 1].max():7.2f}")
        print(f"entry stamps: first {tt[:, 0].min():.2f}, mean {tt[:, 0].mean():.2f}, last {tt[:, 0].max():.2f} us (block scheduling skew)")
        # barrier anatomy: when did the LAST wave arrive (stamp before) and when did waves leave
        for k, nm in (((3, "grid barrier 1"), (7, "grid barrier 2")) if tiles_kernel else ((4, "grid barrier 1"), (9, "grid barrier 2"))):
            print(f"{nm}: last arrival {tt[:, k].max():.2f} us, first release {tt[:, k + 1].min():.2f}, last release {tt[:, k + 1].max():.2f} us")


if __name__ == "__main__":
    main()
