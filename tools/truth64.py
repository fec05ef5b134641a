"""Who is closer to the TRUE trajectory?  The CPU oracle stores its intermediates in float32 (fp64 accumulation only), at the same points
as the reference / plain-fp32 torch do -- so over a long rollout "error against the oracle" contains the oracle's own amplified storage
roundings, which a torch-fp32 rollout largely shares and the HIP engine (which stores fewer intermediates) does not.  Here the yardstick
is the same step evaluated entirely in float64 (tests/torch_ref.py with double parameters and states, on the GPU): configs[1], frames
0 .. N-1 from zero states; pre-mask regression of the HIP engine, of plain-fp32 torch and of the oracle against that trajectory
(conftest.rel_err metric).
usage (GPU box): python tools/truth64.py [--n 100]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import rel_err  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=100)
ap.add_argument("--event-seed", type=int, default=42)
ap.add_argument("--no-oracle", action="store_true")
a = ap.parse_args()

import torch_ref  # noqa: E402
import urnn_amd.weights as uw  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from urnn_amd.dataset import preprocess_inputs  # noqa: E402
from urnn_amd.net_config import load_net_config  # noqa: E402
from urnn_amd.networks import ED, get_network_params  # noqa: E402
from urnn_amd.rollout import RolloutEngine  # noqa: E402

H = W = 500
NUMS, RAIN_MAX, CUM_MAX, T_EVENT = 30, 6.0, 250.0, 360
dev = torch.device("cuda:0")
C = 2 * NUMS + 3
sd = uw.make_state_dict(H, W, C, seed=0)
ep, dp = get_network_params(False, H, W, C, load_net_config())
net = ED(False, ep, dp, 0.5, False, H, W)
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net = net.to(dev).eval()
ev = uw.make_event(T_EVENT, H, W, RAIN_MAX, seed=a.event_seed)
N = a.n
shapes = [(1, 64, H, W), (1, 96, H // 2, W // 2), (1, 96, H // 4, W // 4), (1, 96, H // 4, W // 4), (1, 96, H // 2, W // 2), (1, 64, H, W)]
t0 = time.time()


def torch_rollout(dtype):
    p = {k: torch.from_numpy(v).to(dev).to(dtype) for k, v in sd.items()}
    st = [torch.zeros(s, device=dev, dtype=dtype) for s in shapes]
    raws = []
    with torch.no_grad():
        for t in range(N):
            x = preprocess_inputs(t, ev, dev, nums=NUMS, rain_max=RAIN_MAX, cumsum_rain_max=CUM_MAX)[:, 0].to(dtype)   # the float32 inputs, exactly
            _, _, raw, st = torch_ref.step(p, x, st, H, W)
            raws.append(raw.double().cpu().numpy().reshape(1, H, W))
    return raws, [s.double().cpu().numpy() for s in st]


truth, truth_st = torch_rollout(torch.float64)
print(f"# float64 trajectory: {N} frames in {time.time() - t0:.0f} s", flush=True)
t32, t32_st = torch_rollout(torch.float32)
eng = RolloutEngine(net, H, W, NUMS, RAIN_MAX, CUM_MAX, max_frames=T_EVENT, keep_raw=True, overlap=True, use_graph=True)
eng.load_event(ev)
eng.reset()
eng.run(N)
torch.cuda.synchronize()
hip = [eng.out_raw[t].cpu().numpy().reshape(1, H, W) for t in range(N)]
hip_st = [s.cpu().numpy() for s in eng.final_states()]
rows = {"HIP engine": (hip, hip_st), "torch fp32": (t32, t32_st)}
if not a.no_oracle:
    onet = orc.OracleNet(sd)
    ost = [np.zeros(s, dtype=np.float32) for s in shapes]
    o_raw = []
    for t in range(N):
        _, ost, aux = onet.step(orc.preprocess_inputs(t, ev, NUMS, RAIN_MAX, CUM_MAX)[:, 0], ost, True)
        o_raw.append(aux["reg_raw"].reshape(1, H, W))
    rows["CPU oracle (fp32 storage)"] = (o_raw, ost)
    print(f"# oracle: {time.time() - t0:.0f} s", flush=True)
print("frame | " + " | ".join(rows))
errs = {k: np.array([rel_err(np.asarray(v[0][t], dtype=np.float64), truth[t]) for t in range(N)]) for k, v in rows.items()}
for t in list(range(0, N, 10)) + [N - 1]:
    print(f"{t:5d} | " + " | ".join(f"{errs[k][t]:.2e}" for k in rows))
lo = min(50, N - 1)
for k in rows:
    e = errs[k]
    se = [rel_err(np.asarray(s, dtype=np.float64), ts) for s, ts in zip(rows[k][1], truth_st)]
    print(f"{k:28s} vs float64: worst frame {e.max():.2e} (frame {int(e.argmax())}); mean over frames {lo}..{N - 1}: {e[lo:].mean():.2e}; frames 0..39 max {e[:40].max():.2e}; "
          f"final states {', '.join(f'{v:.1e}' for v in se)}")
if "CPU oracle (fp32 storage)" in rows:
    eo = {k: np.array([rel_err(np.asarray(rows[k][0][t], dtype=np.float64), np.asarray(rows['CPU oracle (fp32 storage)'][0][t], dtype=np.float64)) for t in range(N)]) for k in ("HIP engine", "torch fp32")}
    for k, e in eo.items():
        print(f"{k:28s} vs the oracle: worst frame {e.max():.2e}; mean over frames {lo}..{N - 1}: {e[lo:].mean():.2e}")
print(f"# {time.time() - t0:.0f} s")
