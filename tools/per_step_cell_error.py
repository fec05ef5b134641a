"""Per-frame error of ONE full-resolution cell step (new state vs the float64 cell on the same inputs) along the fp32 torch trajectory
of configs[1]: HIP cell vs torch-fp32 cell, enc1 and dec1, rms and max, every frame.  usage (GPU box): python tools/per_step_cell_error.py [N]"""
import os
import sys

import torch
import torch.nn.functional as Fn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import torch_ref  # noqa: E402
import urnn_amd.weights as uw  # noqa: E402
from urnn_amd.dataset import preprocess_inputs  # noqa: E402
from urnn_amd.net_config import load_net_config  # noqa: E402
from urnn_amd.networks import ED, get_network_params  # noqa: E402

H = W = 500
NUMS, RAIN_MAX, CUM_MAX = 30, 6.0, 250.0
dev = torch.device("cuda:0")
C = 2 * NUMS + 3
sd = uw.make_state_dict(H, W, C, seed=0)
ep, dp = get_network_params(False, H, W, C, load_net_config())
net = ED(False, ep, dp, 0.5, False, H, W)
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net = net.to(dev).eval()
ev = uw.make_event(360, H, W, RAIN_MAX, seed=42)
pt = {k: torch.from_numpy(v).to(dev) for k, v in sd.items()}
ptd = {k: v.double() for k, v in pt.items()}
shapes = [(1, 64, H, W), (1, 96, H // 2, W // 2), (1, 96, H // 4, W // 4), (1, 96, H // 4, W // 4), (1, 96, H // 2, W // 2), (1, 64, H, W)]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
lr = torch_ref._lrelu
conv = lambda name, xx: Fn.conv2d(xx, pt[name + ".weight"], pt[name + ".bias"])
dc = lambda name, xx: Fn.conv_transpose2d(xx, pt[name + ".weight"], pt[name + ".bias"], stride=2)


def errs(mod, prefix, x, e, h):
    ref = torch_ref.cell(ptd, prefix, x.double(), None if e is None else e.double(), h.double())
    hip = mod.step(x, e, h).double() - ref
    t32 = torch_ref.cell(pt, prefix, x, e, h).double() - ref
    f = lambda d: (float(d.pow(2).mean().sqrt()), float(d.abs().max()), float(d.mean(dim=(2, 3)).pow(2).mean().sqrt()))
    return f(hip), f(t32), float(ref.abs().max()), float(h.abs().max())


st = [torch.zeros(s, device=dev) for s in shapes]
print("frame | rain | enc1 HIP rms / max / chan-offset | enc1 torch rms / max / chan-offset | dec1 HIP rms / max / off | dec1 torch rms / max / off | max|h'| enc1 dec1")
with torch.no_grad():
    for t in range(N):
        x = preprocess_inputs(t, ev, dev, nums=NUMS, rain_max=RAIN_MAX, cumsum_rain_max=CUM_MAX)[:, 0].contiguous()
        e1, e2, e3, d1, d2, d3 = st
        a1 = lr(conv("encoder.stage1.conv1_leaky_1", x)).contiguous()
        r1 = errs(net.encoder.rnn1, "encoder.rnn1", a1, None, e1.contiguous())
        e1n = torch_ref.cell(pt, "encoder.rnn1", a1, None, e1)
        a2 = Fn.avg_pool2d(lr(conv("encoder.stage2.conv2_leaky_1", e1n)), 2)
        e2n = torch_ref.cell(pt, "encoder.rnn2", a2, None, e2)
        a3 = Fn.avg_pool2d(lr(conv("encoder.stage3.conv3_leaky_1", e2n)), 2)
        e3n = torch_ref.cell(pt, "encoder.rnn3", a3, None, e3)
        d1n = torch_ref.cell(pt, "decoder.rnn3", None, e3n, d1)
        u3 = lr(dc("decoder.stage3.deconv1_leaky_1", d1n))
        d2n = torch_ref.cell(pt, "decoder.rnn2", u3, e2n, d2)
        u2 = lr(dc("decoder.stage2.deconv2_leaky_1", d2n)).contiguous()
        r5 = errs(net.decoder.rnn1, "decoder.rnn1", u2, e1n.contiguous(), d3.contiguous())
        d3n = torch_ref.cell(pt, "decoder.rnn1", u2, e1n, d3)
        st = [e1n, e2n, e3n, d1n, d2n, d3n]
        if t % 5 == 0 or t == N - 1:
            rain = float(ev["rainfall"].reshape(-1)[t]) if hasattr(ev["rainfall"], "reshape") else 0.0
            g = lambda r: f"{r[0]:.1e} / {r[1]:.1e} / {r[2]:.1e}"
            print(f"{t:5d} | {rain:5.2f} | {g(r1[0])} | {g(r1[1])} | {g(r5[0])} | {g(r5[1])} | {r1[2]:.2f} {r5[2]:.2f}", flush=True)
