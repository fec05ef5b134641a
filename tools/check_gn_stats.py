"""Are the GroupNorm statistics the HIP cell folds from its per-tile partials EXACT?  One enc1 / dec1 step at 500x500 on a heavy-rain
state (frame 70 of configs[1], states from the fp32 torch trajectory); (mean, rstd) per group and (scale, shift) per channel from the
cell's workspace against float64 statistics of the raw planes the same workspace holds.
usage (GPU box): python tools/check_gn_stats.py"""
import os
import sys

import numpy as np
import torch

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
st = [torch.zeros(s, device=dev) for s in shapes]
T0 = int(sys.argv[1]) if len(sys.argv) > 1 else 70
with torch.no_grad():
    for t in range(T0):
        x = preprocess_inputs(t, ev, dev, nums=NUMS, rain_max=RAIN_MAX, cumsum_rain_max=CUM_MAX)[:, 0]
        _, _, _, st = torch_ref.step(pt, x, st, H, W)
    x = preprocess_inputs(T0, ev, dev, nums=NUMS, rain_max=RAIN_MAX, cumsum_rain_max=CUM_MAX)[:, 0].contiguous()
    a1 = net.encoder.stage1(x)
    e1 = st[0].contiguous()


def al(n):
    return (n + 255) // 256 * 256


def check(name, mod, xx, ee, hh, prefix):
    F, P = hh.shape[1], hh.shape[2] * hh.shape[3]
    ws = ops.workspace(ops.gru_cell_workspace_bytes(1, F, hh.shape[2], hh.shape[3]), dev)
    ws.view(torch.float32).fill_(float("nan"))
    out = mod.step(xx, ee, hh, ws=ws)
    torch.cuda.synchronize()
    tiles = max(2, (P + 31) // 32)
    off, views = 0, {}
    for key, n in (("g1", 2 * F * P), ("cx", F * P), ("part1", (2 * F // 32) * tiles * 2), ("part2", (F // 32) * tiles * 2), ("ss1", 2 * F * 2),
                   ("ss2", F * 2), ("st1", (2 * F // 32) * 2), ("st2", (F // 32) * 2)):
        views[key] = ws[off:off + 4 * n].view(torch.float32)
        off += al(4 * n)
    for tag, raw, Cn, stt, ss, gam, bet in (("gates", views["g1"].reshape(2 * F, P), 2 * F, views["st1"], views["ss1"], pt[f"{prefix}.conv1.1.weight"], pt[f"{prefix}.conv1.1.bias"]),
                                            ("candidate", views["cx"].reshape(F, P), F, views["st2"], views["ss2"], pt[f"{prefix}.conv2.1.weight"], pt[f"{prefix}.conv2.1.bias"])):
        G = Cn // 32
        rd = raw.double().reshape(G, -1)
        mean = rd.mean(dim=1)
        var = rd.var(dim=1, unbiased=False)
        rstd = 1.0 / torch.sqrt(var + 1e-5)
        got = stt.reshape(G, 2).double()
        sc = gam.double() * rstd.repeat_interleave(32)
        sh = bet.double() - mean.repeat_interleave(32) * sc
        gss = ss.reshape(Cn, 2).double()
        print(f"{name} {tag}: |mean|/std per group {[round(float(abs(m) * r), 1) for m, r in zip(mean, rstd)]}")
        print(f"   mean rel err {[f'{float(abs(a - b) / abs(b)):.1e}' for a, b in zip(got[:, 0], mean)]}  rstd rel err {[f'{float(abs(a - b) / b):.1e}' for a, b in zip(got[:, 1], rstd)]}")
        print(f"   scale: max rel err {float(((gss[:, 0] - sc).abs() / sc.abs()).max()):.1e}   shift: max |err| {float((gss[:, 1] - sh).abs().max()):.1e} (max |shift| {float(sh.abs().max()):.1f})")
    # the candidate GEMM on its own: float64 W2 . [x; e; sigmoid(GN(raw r)) * h] + b2 from the SAME raw reset gate, against the raw
    # candidate plane in the workspace -- where (which pixels / channels) and how large are the differences?
    print(f"   NaN left in raw gates: {int(torch.isnan(views['g1']).sum())}, raw candidate: {int(torch.isnan(views['cx']).sum())}, output: {int(torch.isnan(out).sum())}")
    raw = views["g1"].reshape(1, 2 * F, hh.shape[2], hh.shape[3]).double()
    g1w, be1 = pt[f"{prefix}.conv1.1.weight"].double(), pt[f"{prefix}.conv1.1.bias"].double()
    gates = torch.nn.functional.group_norm(raw, 2 * F // 32, g1w, be1, 1e-5)
    r = torch.sigmoid(gates[:, F:])
    cat = [t.double() for t in (xx, ee) if t is not None] + [r * hh.double()]
    W2, b2 = pt[f"{prefix}.conv2.0.weight"].double(), pt[f"{prefix}.conv2.0.bias"].double()
    ref = torch.nn.functional.conv2d(torch.cat(cat, dim=1), W2, b2)[0].reshape(F, P)
    got = views["cx"].reshape(F, P).double()
    d = (got - ref)
    print(f"   candidate GEMM vs float64: max |err| {float(d.abs().max()):.2e}, rms {float(d.pow(2).mean().sqrt()):.2e}, mean signed {float(d.mean()):+.2e}; |ref| rms {float(ref.pow(2).mean().sqrt()):.2e}")
    sg = torch.sign(ref)
    ul = torch.pow(2.0, torch.floor(torch.log2(ref.abs().clamp_min(1e-30))) - 23)          # ulp of the float32 result
    print(f"   candidate GEMM: mean of err * sign(ref) {float((d * sg).mean()):+.2e} (toward-zero truncation would be negative), in ulps of the result {float((d * sg / ul).mean()):+.3f}; rms in ulps {float((d / ul).pow(2).mean().sqrt()):.2f}")
    t32 = torch.nn.functional.conv2d(torch.cat(cat, dim=1).float(), W2.float(), b2.float())[0].reshape(F, P).double() - ref
    print(f"   torch fp32 conv:  mean of err * sign(ref) {float((t32 * sg).mean()):+.2e}, in ulps {float((t32 * sg / ul).mean()):+.3f}; rms in ulps {float((t32 / ul).pow(2).mean().sqrt()):.2f}")
    per_ch = d.mean(dim=1)
    print(f"   per-channel mean signed error: max |.| {float(per_ch.abs().max()):.2e} (a plane-wide offset would show here); rms of the per-channel means {float(per_ch.pow(2).mean().sqrt()):.2e}")
    worst = int(d.abs().max(dim=0).values.argmax())
    print(f"   worst pixel {worst} (tile of 128: {worst // 128}, of {P // 128}); error at the last 32 pixels: {float(d[:, -32:].abs().max()):.2e}")
    graw = views["g1"].reshape(2 * F, P).double()
    refg = torch.nn.functional.conv2d(torch.cat([t.double() for t in (xx, ee, hh) if t is not None], dim=1), pt[f"{prefix}.conv1.0.weight"].double(), pt[f"{prefix}.conv1.0.bias"].double())[0].reshape(2 * F, P)
    dg = graw - refg
    print(f"   gate GEMM vs float64:      max |err| {float(dg.abs().max()):.2e}, rms {float(dg.pow(2).mean().sqrt()):.2e}, mean signed {float(dg.mean()):+.2e}; per-channel means rms {float(dg.mean(dim=1).pow(2).mean().sqrt()):.2e}")
    return out


with torch.no_grad():
    e1n = check("enc1", net.encoder.rnn1, a1, None, e1, "encoder.rnn1")
