#!/usr/bin/env python
"""Where does cand_fused_kernel spend its time?  (tuning build: -DURNN_TUNING [-DURNN_TRACE], selected with URNN_LIB)
  1. the launch timed alone under ablation masks URNN_TUNE_ABL (1 no phase 2, 4 no stores, 16 no phase-1 MFMAs, 32 no epilogue, 64 no k-loop)
  2. with a trace build: per-wave phase timeline from s_memtime stamps.
usage: URNN_LIB=.../liburnn_hip_tune.so python tools/cand_probe.py [enc1 dec1]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from urnn_amd import ops, _lib
from urnn_amd.rollout import RolloutEngine
import urnn_amd.weights as uw

cells_wanted = [a for a in sys.argv[1:] if not a.startswith("-")] or ["enc1", "dec1"]
H, W, nums, T, rain_max, cum_max, spatial = bench.CONFIGS["location1"]
dev = torch.device("cuda:0")
net, sd, cfg = bench.build_net(H, W, 63, dev)
eng = RolloutEngine(net, H, W, nums, rain_max, cum_max, max_frames=8, net_cfg=cfg, use_graph=False, device=dev)
eng.load_event(uw.make_event(8, H, W, rain_max, seed=42)); eng.reset(); eng.run(2)
e1, e2, e3, d1, d2, d3 = eng.states
cells = {"enc1": (net.encoder.rnn1, eng.a1, None, e1), "dec1": (net.decoder.rnn1, eng.u2, e1, d3),
         "enc2": (net.encoder.rnn2, eng.a2, None, e2), "dec2": (net.decoder.rnn2, eng.u3, e2, d2)}
L = _lib.lib()
flush = torch.empty(300 * 1024 * 1024 // 4, device=dev)     # > Infinity Cache: every timed launch starts cold


def timeit(fn, iters=12):
    ts = []
    for _ in range(iters):
        flush.add_(1.0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for which in cells_wanted:
    cell, x, e, h = cells[which]
    tmp = h.clone()
    ws = ops.workspace(ops.gru_cell_workspace_bytes(*h.shape), dev)
    flags = ops.PHASE_FUSED_R if hasattr(ops, "PHASE_FUSED_R") else 0
    cell.step(x, e, h, out=tmp, ws=ws, phases=ops.PHASE_ALL | flags)
    for phase, name in ((ops.PHASE_GATES, "gates"), (ops.PHASE_CAND, "cand")):
        for abl in ([0] if phase == ops.PHASE_GATES else [0, 1, 4, 32, 33, 16, 17, 49, 64, 65, 96]):
            os.environ["URNN_TUNE_ABL"] = str(abl)
            us = timeit(lambda: cell.step(x, e, h, out=tmp, phases=phase | flags, ws=ws))
            print(f"{which} {name:5s} abl={abl:3d}: {us:7.1f} us", flush=True)
    os.environ["URNN_TUNE_ABL"] = "0"
    setter = getattr(L, "urnn_debug_set_trace_urnn_cand_fused", None)
    if setter is None:
        continue
    nw = 256 * 8
    buf = torch.zeros(nw * 8 * 8, dtype=torch.int64, device=dev)
    setter.argtypes = [ctypes.c_void_p]
    for stag in (0,) + tuple(int(s) for s in os.environ.get("PROBE_STAGGERS", "").split(",") if s):
        os.environ["URNN_TUNE_CAND_STAGGER"] = str(stag)
        buf.zero_()
        flush.add_(1.0)
        assert setter(buf.data_ptr()) == 0
        torch.cuda.synchronize()
        cell.step(x, e, h, out=tmp, phases=ops.PHASE_CAND | flags, ws=ws)
        torch.cuda.synchronize()
        setter(0)
        t = buf.cpu().numpy().reshape(nw, 8, 8).astype(np.float64)
        used = t[:, :, 7] > 0
        used[:, 7] = False
        t0 = t[used][:, 0].min()
        print(f"---- {which} cand trace, stagger {stag}: waves {int(used[:,0].sum())} items {int(used.sum())}; kernel span {t[used][:,7].max() - t0:.0f} ticks")
        pro = t[:, 7, :3]
        okp = pro[:, 2] > 0
        print(f"  prologue: entry -> own work done {np.mean(pro[okp,1]-pro[okp,0]):7.0f} (waves 0-3 {np.mean((pro[:,1]-pro[:,0])[okp & (np.arange(nw)%8<4)]):7.0f}), -> past the barrier {np.mean(pro[okp,2]-pro[okp,0]):7.0f}; "
              f"entry -> end of the wave's last tile: mean {np.mean(np.max(t[:, :7, 7], axis=1)[okp] - pro[okp, 0]):8.0f} max {np.max(np.max(t[:, :7, 7], axis=1)[okp] - pro[okp, 0]):8.0f}")
        names = ["(k-loop: waiting for slots)", "k-loop - waits", "h loads", "gates rb0", "phase-2 rest", "epi sums", "epi stores", "(total)"]
        wave_id = np.arange(nw) % 8
        for cls, sel in (("A (waves 0-3)", wave_id < 4), ("B (waves 4-7)", wave_id >= 4)):
            for it in range(3):
                m = used[:, it] & sel
                if not m.any():
                    continue
                d = t[m][:, it]
                seg = [d[:, k + 1] - d[:, k] for k in range(7)] + [d[:, 7] - d[:, 0]]
                print(f"  {cls} tile#{it} n={int(m.sum()):4d} start {np.mean(d[:,0]) - t0:8.0f} | " +
                      " ".join(f"{nm} {np.mean(sg):7.0f}" for nm, sg in zip(names, seg)) + f" | end {np.mean(d[:,7]) - t0:8.0f} (max {d[:,7].max() - t0:8.0f})")
    os.environ["URNN_TUNE_CAND_STAGGER"] = "0"
