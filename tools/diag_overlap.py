"""Where do the two-chain schedule's frames differ from the one-chain schedule's (they must be bit-identical)?  Prints, per
differing frame, how many pixels of cls / pre-mask regression differ, their magnitude and their spatial pattern (development aid)."""
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
nums, T = 30, int(os.environ.get("DIAG_T", "120"))
dev = torch.device("cuda:0")
sd = uw.make_state_dict(H, W, 63, seed=0)
ep, dp = get_network_params(False, H, W, 63, load_net_config())
net = ED(False, ep, dp, 0.5, False, H, W)
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net = net.to(dev).eval()
ev = uw.make_event(T, H, W, 6.0, seed=5)
seq = RolloutEngine(net, H, W, nums, 6.0, 250.0, max_frames=T, keep_raw=True, overlap=False, use_graph=True)
seq.rollout(ev)
torch.cuda.synchronize()
ref_cls, ref_raw = seq.out_cls[:T].clone(), seq.out_raw[:T].clone()
ref_states = [s.clone() for s in seq.final_states()]
seq.rollout(ev)
torch.cuda.synchronize()
print("one-chain run twice: cls equal", torch.equal(seq.out_cls[:T], ref_cls), "raw equal", torch.equal(seq.out_raw[:T], ref_raw))
for graph in (True, False):
    ovl = RolloutEngine(net, H, W, nums, 6.0, 250.0, max_frames=T, keep_raw=True, overlap=True, use_graph=graph)
    for rep in range(3):
        ovl.rollout(ev)
        torch.cuda.synchronize()
        nbad = 0
        for t in range(T):
            for name, got, want in (("cls", ovl.out_cls[t, 0], ref_cls[t, 0]), ("raw", ovl.out_raw[t, 0], ref_raw[t, 0])):
                d = (got != want)
                n = int(d.sum())
                if n:
                    nbad += 1
                    if nbad <= 6:
                        ys, xs = torch.nonzero(d, as_tuple=True)
                        flat = (ys * W + xs).cpu().numpy()
                        mag = float((got - want).abs().max())
                        print(f"  graph={graph} run {rep} frame {t} {name}: {n} pixels differ, max |d| {mag:.3e}, flat index range {flat.min()}..{flat.max()}, "
                              f"first few {flat[:8].tolist()}, runs of consecutive: {int((np.diff(flat) == 1).sum())}")
        st_eq = [bool(torch.equal(a, b)) for a, b in zip(ovl.final_states(), ref_states)]
        print(f"graph={graph} run {rep}: {nbad} (frame, tensor) pairs differ; final states equal: {st_eq}")
