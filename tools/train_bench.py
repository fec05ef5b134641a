#!/usr/bin/env python
"""Time SWP training windows on the HIP path at a real grid size (development tool; first-version kernels)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import urnn_amd.weights as uw
from urnn_amd.training import Trainer

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="location1")
ap.add_argument("--seq-num", type=int, default=4)
ap.add_argument("--windows", type=int, default=3)
a = ap.parse_args()
H, W, nums, T, rain_max, cum_max, spatial = bench.CONFIGS[a.config]
dev = torch.device("cuda:0")
net, sd, cfg = bench.build_net(H, W, 2 * nums + 3, dev)
tr = Trainer(net, H, W, nums, rain_max, cum_max, lr=1e-4, grad_clip=1.0)
frames = a.seq_num * (a.windows + 1)
ev = uw.make_event(frames, H, W, rain_max, seed=42, spatial_rain=spatial)
label = (torch.rand(1, frames, H, W, device=dev) ** 3)
label[label < 0.1] = 0
states = None
loss, states = tr.train_window(ev, label[:, :a.seq_num], 0, a.seq_num, states)      # warm-up window (allocations)
torch.cuda.synchronize()
t0 = time.perf_counter()
for w in range(1, a.windows + 1):
    loss, states = tr.train_window(ev, label[:, w * a.seq_num:(w + 1) * a.seq_num], w * a.seq_num, a.seq_num, states)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.windows
print(f"{a.config} {H}x{W}: window of {a.seq_num} steps (forward + backward + Adam) {dt * 1e3:.1f} ms = {dt / a.seq_num * 1e3:.1f} ms/step; "
      f"loss {float(loss[0]):.4f}, grad norm {float(tr.last['clip'][1]):.3f}, peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
