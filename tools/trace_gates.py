#!/usr/bin/env python
"""Per-wave phase timeline of the dec1 gate GEMM from s_memtime stamps (tuning build with -DURNN_TRACE)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from urnn_amd import ops, _lib
from urnn_amd.rollout import RolloutEngine
import urnn_amd.weights as uw

which = sys.argv[1] if len(sys.argv) > 1 else "dec1"
phase = {"gates": ops.PHASE_GATES, "cand": ops.PHASE_CAND}[sys.argv[2] if len(sys.argv) > 2 else "gates"]
H, W, nums, T, rain_max, cum_max, spatial = bench.CONFIGS["location1"]
dev = torch.device("cuda:0")
net, sd, cfg = bench.build_net(H, W, 63, dev)
eng = RolloutEngine(net, H, W, nums, rain_max, cum_max, max_frames=8, net_cfg=cfg, use_graph=False, device=dev)
eng.load_event(uw.make_event(8, H, W, rain_max, seed=42)); eng.reset(); eng.run(2)
e1, e2, e3, d1, d2, d3 = eng.states
cells = {"enc1": (net.encoder.rnn1, eng.a1, None, e1), "dec1": (net.decoder.rnn1, eng.u2, e1, d3),
         "enc2": (net.encoder.rnn2, eng.a2, None, e2), "dec2": (net.decoder.rnn2, eng.u3, e2, d2),
         "enc3": (net.encoder.rnn3, eng.a3, None, e3), "dec3": (net.decoder.rnn3, None, e3, d1)}
cell, x, e, h = cells[which]
tmp = h.clone()
ws = ops.workspace(ops.gru_cell_workspace_bytes(*h.shape), dev)
cell.step(x, e, h, out=tmp, ws=ws)          # the candidate phase reads the gates' raw values and partial statistics
for _ in range(3):
    cell.step(x, e, h, out=tmp, phases=phase, ws=ws)
nw = 512 * 8
buf = torch.zeros(nw * 8 * 4, dtype=torch.int64, device=dev)
L = _lib.lib()
setters = [getattr(L, "urnn_debug_set_trace_" + tu, None) for tu in ("urnn_gemm", "urnn_gemm_gates", "urnn_gemm_cand", "urnn_gemm_deconv")]
setters = [f for f in setters if f is not None]          # one trace pointer per translation unit of the GEMM template
assert setters, "build the trace library first: python tools/build_variants.py trace; URNN_LIB=u-rnn_amd/liburnn_hip_trace.so"
for f in setters:
    f.argtypes = [ctypes.c_void_p]
    assert f(buf.data_ptr()) == 0
torch.cuda.synchronize()
cell.step(x, e, h, out=tmp, phases=phase, ws=ws)
torch.cuda.synchronize()
for f in setters:
    f(0)
t = buf.cpu().numpy().reshape(nw, 8, 4).astype(np.float64)
used = t[:, :, 3] > 0
# s_memtime bases differ across the chip: reference every wave to the earliest stamp of its own block
tb = t.reshape(nw // 8, 8 * 8, 4)
ub = used.reshape(nw // 8, 8 * 8)
for i in range(nw // 8):
    if ub[i].any():
        t0b = tb[i][ub[i]][:, 0].min()
        tb[i][ub[i]] -= t0b
t = tb.reshape(nw, 8, 4)
t0 = 0.0
print(which, "waves with work:", int(used[:, 0].sum()), "items:", int(used.sum()))
tk = t[used]
clk = 100e6   # s_memtime ticks at 100 MHz on gfx9? report raw and assume
span = (tk[:, 3].max() - t0)
print("kernel span ticks:", span)
for name, a, b in (("prologue (start->first frag)", 0, 1), ("k-loop", 1, 2), ("epilogue (incl. store drain)", 2, 3)):
    d = tk[:, b] - tk[:, a]
    print(f"{name:32s} mean {d.mean():10.0f}  p10 {np.percentile(d,10):10.0f}  p50 {np.percentile(d,50):10.0f}  p90 {np.percentile(d,90):10.0f}  max {d.max():10.0f}")
# per wave class (A = first wave on its SIMD, B = second, held back half a tile), times relative to the kernel start
wave_id = np.arange(nw) % 8
for cls, sel in (("A", wave_id < 4), ("B", wave_id >= 4)):
    for it in range(4):
        m = used[:, it] & sel
        if m.any():
            d = t[m][:, it]
            print(f"{cls} item#{it}: n={m.sum():5d} start {d[:,0].mean()-t0:9.0f} frag {np.mean(d[:,1]-d[:,0]):7.0f} kloop {np.mean(d[:,2]-d[:,1]):9.0f} "
                  f"(p10 {np.percentile(d[:,2]-d[:,1],10):9.0f} p90 {np.percentile(d[:,2]-d[:,1],90):9.0f}) epi {np.mean(d[:,3]-d[:,2]):8.0f} end {d[:,3].mean()-t0:9.0f} (max {d[:,3].max()-t0:9.0f})")
# per-item index breakdown
for it in range(4):
    m = used[:, it]
    if m.any():
        d = t[m][:, it]
        print(f"item#{it}: n={m.sum()} start {d[:,0].mean()-t0:10.0f} kloop {np.mean(d[:,2]-d[:,1]):10.0f} epi {np.mean(d[:,3]-d[:,2]):10.0f} end {d[:,3].mean()-t0:10.0f}")
