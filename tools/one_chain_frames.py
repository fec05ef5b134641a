#!/usr/bin/env python
"""N frames of the ONE-chain rollout engine, eager (every dispatch attributed), nothing else: the target of the counter passes that measure
the one-chain schedule's bytes per frame (tools/collect_profiles.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE -- python tools/one_chain_frames.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from urnn_amd.rollout import RolloutEngine
import urnn_amd.weights as uw

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 12
H, W, nums, T, rain_max, cum_max, spatial = bench.CONFIGS["location1"]
dev = torch.device("cuda:0")
net, sd, cfg = bench.build_net(H, W, 2 * nums + 3, dev)
eng = RolloutEngine(net, H, W, nums, rain_max, cum_max, max_frames=frames, net_cfg=cfg, use_graph=False, overlap=False, device=dev)
eng.load_event(uw.make_event(frames, H, W, rain_max, seed=42))
eng.reset()
eng.run(frames)
torch.cuda.synchronize()
print("frames", frames, "cooperative cells:", {k: bool(v) for k, v in eng._coop.items()})
