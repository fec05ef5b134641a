#!/usr/bin/env python
"""Event timing of the GRU gate / candidate GEMMs at a given size (development tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from urnn_amd import ops
from urnn_amd.rollout import RolloutEngine
import urnn_amd.weights as uw

def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) * 1e3 / iters

H, W, nums, T, rain_max, cum_max, spatial = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "location1"]
dev = torch.device("cuda:0")
net, sd, cfg = bench.build_net(H, W, 2 * nums + 3, dev)
eng = RolloutEngine(net, H, W, nums, rain_max, cum_max, max_frames=8, spatial_rain=spatial, net_cfg=cfg, use_graph=False, device=dev)
eng.load_event(uw.make_event(8, H, W, rain_max, seed=42, spatial_rain=spatial)); eng.reset(); eng.run(2)
enc, dec = net.encoder, net.decoder
e1, e2, e3, d1, d2, d3 = eng.states
cells = [("enc1", enc.rnn1, eng.a1, None, e1), ("dec1", dec.rnn1, eng.u2, e1, d3), ("enc2", enc.rnn2, eng.a2, None, e2), ("dec2", dec.rnn2, eng.u3, e2, d2)]
out = []
for name, cell, x, e, h in cells:
    tmp = h.clone()
    ws = ops.workspace(ops.gru_cell_workspace_bytes(h.shape[0], h.shape[1], h.shape[2], h.shape[3]), h.device)   # one scratch: the candidate reads the gates' output
    g = timeit(lambda: cell.step(x, e, h, out=tmp, phases=ops.PHASE_GATES, ws=ws))
    c = timeit(lambda: cell.step(x, e, h, out=tmp, phases=ops.PHASE_CAND, ws=ws))
    out.append(f"{name}: gates {g:7.1f} us  cand {c:6.1f} us")
print(os.environ.get("URNN_LIB", "product"), " | ".join(out))
