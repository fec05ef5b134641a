"""How much does the error of an fp32 rollout against the CPU oracle move when the computation is perturbed by ONE rounding error?
configs[1] (500x500, C = 63), frames 0 .. N-1 from zero states.  The plain-fp32 torch restatement (tests/torch_ref.py) and the HIP
engine are each run unperturbed and K times with their six states multiplied, once, after frame 0, by (1 + 6e-8 * N(0,1)) -- a
perturbation of the size of a single fp32 rounding per element.  If the spread of the perturbed runs' worst-frame error is as wide
as the gap between the two implementations, the gap is the rollout's sensitivity, not an arithmetic property of either.
usage (GPU box): python tools/noise_floor.py [--n 100] [--k 3]"""
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
ap.add_argument("--k", type=int, default=3)
ap.add_argument("--event-seed", type=int, default=42)
ap.add_argument("--skip-torch", action="store_true")
ap.add_argument("--matrix-mode", default="fp32")
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
from urnn_amd import ops as _ops  # noqa: E402
from urnn_amd._lib import check as _check, lib as _lib  # noqa: E402
_check(_lib().urnn_set_matrix_mode(_ops.MATRIX_MODES[a.matrix_mode]), "urnn_set_matrix_mode")
print(f"# matrix mode {a.matrix_mode}")
C = 2 * NUMS + 3
sd = uw.make_state_dict(H, W, C, seed=0)
ep, dp = get_network_params(False, H, W, C, load_net_config())
net = ED(False, ep, dp, 0.5, False, H, W)
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net = net.to(dev).eval()
ev = uw.make_event(T_EVENT, H, W, RAIN_MAX, seed=a.event_seed)
N = a.n

t0 = time.time()
CACHE = f"/tmp/noise_floor_oracle_{a.event_seed}_{N}.npy"
onet = orc.OracleNet(sd)
ost = [np.zeros(s, dtype=np.float32) for s in [(1, 64, H, W), (1, 96, H // 2, W // 2), (1, 96, H // 4, W // 4), (1, 96, H // 4, W // 4),
                                                (1, 96, H // 2, W // 2), (1, 64, H, W)]]
if os.path.isfile(CACHE):
    o_raw = list(np.load(CACHE))
    print(f"# oracle outputs from {CACHE}")
else:
    o_raw = []
    for t in range(N):
        _, ost, aux = onet.step(orc.preprocess_inputs(t, ev, NUMS, RAIN_MAX, CUM_MAX)[:, 0], ost, True)
        o_raw.append(aux["reg_raw"].reshape(1, H, W))
    np.save(CACHE, np.stack(o_raw))
    print(f"# oracle: {N} frames in {time.time() - t0:.0f} s")


def noise(shape, gen):
    return 1.0 + 6e-8 * torch.randn(shape, device=dev, generator=gen)


def torch_run(seed):
    pt = {k: torch.from_numpy(v).to(dev) for k, v in sd.items()}
    st = [torch.zeros(s.shape, device=dev) for s in ost]
    gen = torch.Generator(device=dev)
    errs = []
    with torch.no_grad():
        for t in range(N):
            x = preprocess_inputs(t, ev, dev, nums=NUMS, rain_max=RAIN_MAX, cumsum_rain_max=CUM_MAX)[:, 0]
            _, _, tr, st = torch_ref.step(pt, x, st, H, W)
            if t == 0 and seed is not None:
                gen.manual_seed(seed)
                st = [s * noise(s.shape, gen) for s in st]
            errs.append(rel_err(tr.cpu().numpy().reshape(1, H, W), o_raw[t]))
    return errs


def hip_run(seed):
    # one kernel chain: eng.states are then the live buffers of the next frame (the two-chain schedule ping-pongs the encoder states)
    eng = RolloutEngine(net, H, W, NUMS, RAIN_MAX, CUM_MAX, max_frames=T_EVENT, keep_raw=True, overlap=False, use_graph=True)
    eng.load_event(ev)
    eng.reset()
    eng.run(1)
    torch.cuda.synchronize()
    if seed is not None:
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed)
        for s in eng.states:
            s.mul_(noise(s.shape, gen))
    eng.run(N - 1)
    torch.cuda.synchronize()
    raw = eng.out_raw[:N].cpu().numpy()
    return [rel_err(raw[t].reshape(1, H, W), o_raw[t]) for t in range(N)]


def summary(name, errs):
    e = np.array(errs)
    lo = min(50, N - 1)
    print(f"{name:28s} worst frame {e.max():.2e} (frame {int(e.argmax())}); mean over frames {lo}..{N - 1}: {e[lo:].mean():.2e}; "
          f"frames 0..{min(40, N) - 1}: max {e[:40].max():.2e}")
    return e.max(), e[lo:].mean()


res = {"torch": [], "hip": []}
if not a.skip_torch:
    res["torch"].append(summary("torch-fp32", torch_run(None)))
    for k in range(a.k):
        res["torch"].append(summary(f"torch-fp32 + 1 ulp noise #{k}", torch_run(100 + k)))
res["hip"].append(summary("HIP", hip_run(None)))
for k in range(a.k):
    res["hip"].append(summary(f"HIP + 1 ulp noise #{k}", hip_run(100 + k)))
for name, v in res.items():
    if not v:
        continue
    w = np.array([x[0] for x in v]); m = np.array([x[1] for x in v])
    print(f"{name}: worst-frame error over {len(v)} runs: min {w.min():.2e} max {w.max():.2e}; heavy-rain mean: min {m.min():.2e} max {m.max():.2e}")
print(f"# {time.time() - t0:.0f} s")
