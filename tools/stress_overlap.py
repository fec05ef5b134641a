"""Repeat whole-event rollouts on the benchmarked schedule and report every departure from the first one: frame, tensor, how many
pixels, magnitude and the spatial pattern of the differing pixels (development aid for run-to-run differences)."""
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
nums, T = 30, int(os.environ.get("DIAG_T", "360"))
N = int(os.environ.get("DIAG_N", "40"))
overlap = bool(int(os.environ.get("DIAG_OVERLAP", "1")))
dev = torch.device("cuda:0")
sd = uw.make_state_dict(H, W, 63, seed=0)
ep, dp = get_network_params(False, H, W, 63, load_net_config())
net = ED(False, ep, dp, 0.5, False, H, W)
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net = net.to(dev).eval()
ev = uw.make_event(T, H, W, 6.0, seed=5)
eng = RolloutEngine(net, H, W, nums, 6.0, 250.0, max_frames=T, keep_raw=True, overlap=overlap, use_graph=True)
eng.rollout(ev)
torch.cuda.synchronize()
ref_cls, ref_raw = eng.out_cls[:T].clone(), eng.out_raw[:T].clone()
ref_states = [s.clone() for s in eng.final_states()]
nbad_runs = 0
for rep in range(N):
    eng.rollout(ev)
    torch.cuda.synchronize()
    bad = []
    for name, got, want in (("cls", eng.out_cls[:T, 0], ref_cls[:, 0]), ("raw", eng.out_raw[:T, 0], ref_raw[:, 0])):
        neq = (got != want).reshape(T, -1)
        per_frame = neq.sum(1).cpu().numpy()
        for t in np.nonzero(per_frame)[0][:4]:
            px = torch.nonzero(neq[t], as_tuple=True)[0].cpu().numpy()
            mag = float((got[t] - want[t]).abs().max())
            bad.append(f"{name} frame {t}: {len(px)} px, max|d| {mag:.2e}, px range {px.min()}..{px.max()}, distinct px//512 {len(set((px // 512).tolist()))}, first {px[:6].tolist()}")
        if per_frame.any():
            bad.append(f"{name}: frames differing: {np.nonzero(per_frame)[0][:20].tolist()} ({int((per_frame > 0).sum())} frames)")
    st = [bool(torch.equal(a, b)) for a, b in zip(eng.final_states(), ref_states)]
    if bad or not all(st):
        nbad_runs += 1
        print(f"rollout {rep}: states equal {st}")
        for b in bad:
            print("   ", b)
print(f"overlap={overlap}: {nbad_runs} of {N} rollouts differ from the first")
