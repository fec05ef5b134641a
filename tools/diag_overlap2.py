"""Frame-by-frame: one-chain vs two-chain engine, all six states + cls + raw after every frame; first mismatch per tensor with its
spatial pattern.  DIAG_T frames, DIAG_STEP frames per run() call."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import urnn_amd.weights as uw  # noqa: E402
from urnn_amd.net_config import load_net_config  # noqa: E402
from urnn_amd.networks import ED, get_network_params  # noqa: E402
from urnn_amd.rollout import RolloutEngine  # noqa: E402

H = W = 500
nums, T = 30, int(os.environ.get("DIAG_T", "60"))
dev = torch.device("cuda:0")
sd = uw.make_state_dict(H, W, 63, seed=0)
ep, dp = get_network_params(False, H, W, 63, load_net_config())
net = ED(False, ep, dp, 0.5, False, H, W)
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net = net.to(dev).eval()
ev = uw.make_event(T, H, W, 6.0, seed=5)
use_graph = bool(int(os.environ.get("DIAG_GRAPH", "0")))
seq = RolloutEngine(net, H, W, nums, 6.0, 250.0, max_frames=T, keep_raw=True, overlap=False, use_graph=use_graph)
ovl = RolloutEngine(net, H, W, nums, 6.0, 250.0, max_frames=T, keep_raw=True, overlap=True, use_graph=use_graph)
for e in (seq, ovl):
    e.load_event(ev)
    e.reset()
names = ["e1", "e2", "e3", "d1", "d2", "d3"]
reported = set()
for t in range(T):
    seq.run(1)
    ovl.run(1)
    torch.cuda.synchronize()
    items = [(n, a, b) for n, a, b in zip(names, seq.final_states(), ovl.final_states())]
    items += [("cls", seq.out_cls[t], ovl.out_cls[t]), ("raw", seq.out_raw[t], ovl.out_raw[t])]
    for n, a, b in items:
        d = a != b
        k = int(d.sum())
        if k and n not in reported:
            reported.add(n)
            idx = torch.nonzero(d.reshape(-1), as_tuple=True)[0].cpu().numpy()
            Pp = a.shape[-1] * a.shape[-2]
            ch, px = idx // Pp, idx % Pp
            print(f"frame {t} {n} {tuple(a.shape)}: {k} values differ, max |d| {float((a - b).abs().max()):.3e}; channels {sorted(set(ch.tolist()))[:12]}..., "
                  f"pixel range {px.min()}..{px.max()}, distinct pixels {len(set(px.tolist()))}, pixel // 64 distinct {len(set((px // 64).tolist()))}, first pixels {sorted(set(px.tolist()))[:10]}")
    if len(reported) == 8:
        break
print("tensors that ever differed:", sorted(reported))
