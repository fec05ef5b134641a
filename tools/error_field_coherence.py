"""Is the rounding error of the HIP cell GEMMs coherent in TIME?  enc1 at 500x500 on two consecutive heavy-rain frames of configs[1]
(states from the fp32 torch trajectory): error fields (kernel output minus a float64 GEMM on the same inputs) of the HIP gate GEMM,
the HIP candidate GEMM and torch's own fp32 convs, and the correlation of each field between frame T and frame T+1 / T+10.
Independent roundings decorrelate as soon as the inputs change; an error that is a fixed function of position or of slowly varying
inputs does not -- and a recurrence integrates what does not decorrelate.
usage (GPU box): python tools/error_field_coherence.py [T]"""
import os
import sys

import torch
import torch.nn.functional as Fn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import torch_ref  # noqa: E402
import urnn_amd.weights as uw  # noqa: E402
from urnn_amd import ops  # noqa: E402
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
shapes = [(1, 64, H, W), (1, 96, H // 2, W // 2), (1, 96, H // 4, W // 4), (1, 96, H // 4, W // 4), (1, 96, H // 2, W // 2), (1, 64, H, W)]
T0 = int(sys.argv[1]) if len(sys.argv) > 1 else 70
F, P = 64, H * W
prefix = "encoder.rnn1"
W1, b1 = pt[f"{prefix}.conv1.0.weight"], pt[f"{prefix}.conv1.0.bias"]
W2, b2 = pt[f"{prefix}.conv2.0.weight"], pt[f"{prefix}.conv2.0.bias"]
g1w, be1 = pt[f"{prefix}.conv1.1.weight"], pt[f"{prefix}.conv1.1.bias"]


def fields_dec1(u2, e1n, h):
    pre = "decoder.rnn1"
    V1, c1 = pt[f"{pre}.conv1.0.weight"], pt[f"{pre}.conv1.0.bias"]
    V2, c2 = pt[f"{pre}.conv2.0.weight"], pt[f"{pre}.conv2.0.bias"]
    gw, gb = pt[f"{pre}.conv1.1.weight"], pt[f"{pre}.conv1.1.bias"]
    ws = ops.workspace(ops.gru_cell_workspace_bytes(1, F, H, W), dev)
    net.decoder.rnn1.step(u2, e1n, h, ws=ws)
    torch.cuda.synchronize()
    g1 = ws[:2 * F * P * 4].view(torch.float32).reshape(1, 2 * F, H, W).clone()
    cx = ws[2 * F * P * 4:3 * F * P * 4].view(torch.float32).reshape(1, F, H, W).clone()
    cat1 = torch.cat([u2, e1n, h], dim=1)
    ref_g = Fn.conv2d(cat1.double(), V1.double(), c1.double())
    r = torch.sigmoid(Fn.group_norm(g1.double(), 4, gw.double(), gb.double(), 1e-5)[:, F:])
    cat2 = torch.cat([u2.double(), e1n.double(), r * h.double()], dim=1)
    ref_c = Fn.conv2d(cat2, V2.double(), c2.double())
    out = {"dec1 HIP gate GEMM": (g1.double() - ref_g), "dec1 HIP candidate GEMM": (cx.double() - ref_c),
           "dec1 torch fp32 conv (gates)": (Fn.conv2d(cat1, V1, c1).double() - ref_g),
           "dec1 torch fp32 conv (candidate)": (Fn.conv2d(cat2.float(), V2, c2).double() - ref_c)}
    return {k: v.float() for k, v in out.items()}


def torch_until_u2(x, st):
    e1, e2, e3, d1, d2, d3 = st
    conv = lambda name, xx: Fn.conv2d(xx, pt[name + ".weight"], pt[name + ".bias"])
    lr = torch_ref._lrelu
    a1 = lr(conv("encoder.stage1.conv1_leaky_1", x))
    e1n = torch_ref.cell(pt, "encoder.rnn1", a1, None, e1)
    a2 = Fn.avg_pool2d(lr(conv("encoder.stage2.conv2_leaky_1", e1n)), 2)
    e2n = torch_ref.cell(pt, "encoder.rnn2", a2, None, e2)
    a3 = Fn.avg_pool2d(lr(conv("encoder.stage3.conv3_leaky_1", e2n)), 2)
    e3n = torch_ref.cell(pt, "encoder.rnn3", a3, None, e3)
    d1n = torch_ref.cell(pt, "decoder.rnn3", None, e3n, d1)
    dc = lambda name, xx: Fn.conv_transpose2d(xx, pt[name + ".weight"], pt[name + ".bias"], stride=2)
    u3 = lr(dc("decoder.stage3.deconv1_leaky_1", d1n))
    d2n = torch_ref.cell(pt, "decoder.rnn2", u3, e2n, d2)
    u2 = lr(dc("decoder.stage2.deconv2_leaky_1", d2n))
    return u2.contiguous(), e1n.contiguous()


ptd = {k: v.double() for k, v in pt.items()}


def fields_cell(name, mod, prefix_, x, e, h):
    """error fields of the WHOLE cell's new state against the float64 cell on the same inputs"""
    ref = torch_ref.cell(ptd, prefix_, None if x is None else x.double(), None if e is None else e.double(), h.double())
    hip = mod.step(x, e, h)
    t32 = torch_ref.cell(pt, prefix_, x, e, h)
    return {f"{name} HIP cell h'": (hip.double() - ref).float(), f"{name} torch fp32 cell h'": (t32.double() - ref).float()}


def fields(a1, h):
    ws = ops.workspace(ops.gru_cell_workspace_bytes(1, F, H, W), dev)
    net.encoder.rnn1.step(a1, None, h, ws=ws)
    torch.cuda.synchronize()
    g1 = ws[:2 * F * P * 4].view(torch.float32).reshape(1, 2 * F, H, W).clone()
    cx = ws[2 * F * P * 4:3 * F * P * 4].view(torch.float32).reshape(1, F, H, W).clone()
    cat1 = torch.cat([a1, h], dim=1)
    ref_g = Fn.conv2d(cat1.double(), W1.double(), b1.double())
    r = torch.sigmoid(Fn.group_norm(g1.double(), 4, g1w.double(), be1.double(), 1e-5)[:, F:])
    cat2 = torch.cat([a1.double(), r * h.double()], dim=1)
    ref_c = Fn.conv2d(cat2, W2.double(), b2.double())
    out = {"HIP gate GEMM": (g1.double() - ref_g), "HIP candidate GEMM": (cx.double() - ref_c),
           "torch fp32 conv (gates)": (Fn.conv2d(cat1, W1, b1).double() - ref_g),
           "torch fp32 conv (candidate)": (Fn.conv2d(cat2.float(), W2, b2).double() - ref_c)}
    return {k: v.float() for k, v in out.items()}


st = [torch.zeros(s, device=dev) for s in shapes]
snap = {}
with torch.no_grad():
    for t in range(T0 + 11):
        x = preprocess_inputs(t, ev, dev, nums=NUMS, rain_max=RAIN_MAX, cumsum_rain_max=CUM_MAX)[:, 0].contiguous()
        if t in (T0, T0 + 1, T0 + 10):
            snap[t] = fields(net.encoder.stage1(x), st[0].contiguous())
            u2_, e1n_ = torch_until_u2(x, st)
            snap[t].update(fields_dec1(u2_, e1n_, st[5].contiguous()))
            snap[t].update(fields_cell("enc1", net.encoder.rnn1, "encoder.rnn1", net.encoder.stage1(x), None, st[0].contiguous()))
            snap[t].update(fields_cell("dec1", net.decoder.rnn1, "decoder.rnn1", u2_, e1n_, st[5].contiguous()))
        _, _, _, st = torch_ref.step(pt, x, st, H, W)


def corr(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    a, b = a - a.mean(), b - b.mean()
    return float((a * b).sum() / torch.sqrt((a * a).sum() * (b * b).sum()))


for k in snap[T0]:
    e0, e1, e10 = snap[T0][k], snap[T0 + 1][k], snap[T0 + 10][k]
    cm = lambda e: e - e.mean(dim=(2, 3), keepdim=True)
    print(f"{k:34s} max {float(e0.abs().max()):.2e}  per-channel offsets removed: corr({T0},{T0 + 1}) = {corr(cm(e0), cm(e1)):+.3f}", end="  ")
    print(f"rms {float(e0.double().pow(2).mean().sqrt()):.2e}   corr(frame {T0}, {T0 + 1}) = {corr(e0, e1):+.3f}   corr(frame {T0}, {T0 + 10}) = {corr(e0, e10):+.3f}")
