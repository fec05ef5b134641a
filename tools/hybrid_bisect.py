"""Which part of the HIP step carries the SYSTEMATIC error of a long rollout?  (tools/noise_floor.py showed that the HIP engine's
error against the oracle in the heavy-rain part of configs[1] does not move under one-ulp perturbations -- it is a deterministic
bias, not amplified roundoff.)  Hybrid rollouts from zero states: every layer is computed either by the HIP module or by the
plain-fp32 torch restatement (tests/torch_ref.py), per variant; error of the pre-mask regression against the oracle.
usage (GPU box): python tools/hybrid_bisect.py [--n 100]"""
import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as Fn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import rel_err  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=100)
ap.add_argument("--variants", default="")
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
ev = uw.make_event(T_EVENT, H, W, RAIN_MAX, seed=42)
N = a.n
pt = {k: torch.from_numpy(v).to(dev) for k, v in sd.items()}
shapes = [(1, 64, H, W), (1, 96, H // 2, W // 2), (1, 96, H // 4, W // 4), (1, 96, H // 4, W // 4), (1, 96, H // 2, W // 2), (1, 64, H, W)]

t0 = time.time()
onet = orc.OracleNet(sd)
ost = [np.zeros(s, dtype=np.float32) for s in shapes]
o_raw = []
for t in range(N):
    _, ost, aux = onet.step(orc.preprocess_inputs(t, ev, NUMS, RAIN_MAX, CUM_MAX)[:, 0], ost, True)
    o_raw.append(aux["reg_raw"].reshape(1, H, W))
print(f"# oracle: {N} frames in {time.time() - t0:.0f} s")

enc, dec = net.encoder, net.decoder
conv_t = lambda name, x: torch_ref._lrelu(Fn.conv2d(x, pt[name + ".weight"], pt[name + ".bias"]))
dc_t = lambda name, x: torch_ref._lrelu(Fn.conv_transpose2d(x, pt[name + ".weight"], pt[name + ".bias"], stride=2))


def gn_custom(x, G, gamma, beta, mode):
    """GroupNorm with statistics in double and the affine evaluated in float32 the way a kernel would: mode 'fold' = x * fl(sc) + fl(beta
    - mean * sc) (one fma; the shift from the UNROUNDED scale: what the HIP cell kernels did in round 3), 'fold_consistent' = the
    shift from the rounded scale (PyTorch's fused parameters), 'centred' = (x - fl(mean)) * fl(sc) + fl(beta - (mean - fl(mean)) * sc)."""
    B, Cc, Hh, Ww = x.shape
    xd = x.double().reshape(B, G, -1)
    mean = xd.mean(dim=2)
    var = xd.var(dim=2, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + torch_ref.EPS)
    mean_c = mean.repeat_interleave(Cc // G, dim=1)[:, :, None, None]
    sc = (gamma.double()[None, :] * rstd.repeat_interleave(Cc // G, dim=1))[:, :, None, None]
    bt = beta.double()[None, :, None, None]
    fsc = sc.float()
    if mode == "fold":
        fsh = (bt - mean_c * sc).float()
        return (x.double() * fsc.double() + fsh.double()).float()
    if mode == "fold_consistent":
        fsh = (bt - mean_c * fsc.double()).float()
        return (x.double() * fsc.double() + fsh.double()).float()
    m0 = mean_c.float()
    fsh = (bt - (mean_c - m0.double()) * sc).float()
    return ((x - m0).double() * fsc.double() + fsh.double()).float()


def cell_custom(prefix, x, e, h, mode):
    F = h.shape[1]
    W1, b1, g1, be1 = (pt[f"{prefix}.conv1.{k}"] for k in ("0.weight", "0.bias", "1.weight", "1.bias"))
    W2, b2, g2, be2 = (pt[f"{prefix}.conv2.{k}"] for k in ("0.weight", "0.bias", "1.weight", "1.bias"))
    cat = lambda *t: torch.cat([u for u in t if u is not None], dim=1)
    gates = gn_custom(Fn.conv2d(cat(x, e, h), W1, b1), 2 * F // 32, g1, be1, mode)
    z, r = torch.sigmoid(gates[:, :F]), torch.sigmoid(gates[:, F:])
    n = torch.tanh(gn_custom(Fn.conv2d(cat(x, e, r * h), W2, b2), F // 32, g2, be2, mode))
    return (1 - z) * h + z * n


def cell_hip_gates(mod, prefix, x, e, h):
    """raw gates from the HIP gate GEMM (phase 1 of the cell, read back from its workspace), everything after them in torch"""
    from urnn_amd import ops
    F = h.shape[1]
    P = h.shape[2] * h.shape[3]
    ws = ops.workspace(ops.gru_cell_workspace_bytes(1, F, h.shape[2], h.shape[3]), dev)
    mod.step(x, e, h, phases=ops.PHASE_GATES, ws=ws)
    raw = ws[:2 * F * P * 4].view(torch.float32).reshape(1, 2 * F, h.shape[2], h.shape[3]).clone()
    g1, be1 = pt[f"{prefix}.conv1.1.weight"], pt[f"{prefix}.conv1.1.bias"]
    W2, b2, g2, be2 = (pt[f"{prefix}.conv2.{k}"] for k in ("0.weight", "0.bias", "1.weight", "1.bias"))
    cat = lambda *t: torch.cat([u for u in t if u is not None], dim=1)
    gates = Fn.group_norm(raw, 2 * F // 32, g1, be1, torch_ref.EPS)
    z, r = torch.sigmoid(gates[:, :F]), torch.sigmoid(gates[:, F:])
    n = torch.tanh(Fn.group_norm(Fn.conv2d(cat(x, e, r * h), W2, b2), F // 32, g2, be2, torch_ref.EPS))
    return (1 - z) * h + z * n


def cell_hip_gates_cand(mod, prefix, x, e, h):
    """raw gates AND raw candidate from the HIP kernels (phases 1-2: gate GEMM, gated candidate GEMM), norms / activations / blend in torch"""
    from urnn_amd import ops
    F = h.shape[1]
    P = h.shape[2] * h.shape[3]
    ws = ops.workspace(ops.gru_cell_workspace_bytes(1, F, h.shape[2], h.shape[3]), dev)
    mod.step(x, e, h, phases=ops.PHASE_GATES, ws=ws)
    mod.step(x, e, h, phases=ops.PHASE_CAND, ws=ws)
    raw = ws[:2 * F * P * 4].view(torch.float32).reshape(1, 2 * F, h.shape[2], h.shape[3]).clone()
    off = ((2 * F * P * 4 + 255) // 256) * 256
    cx = ws[off:off + F * P * 4].view(torch.float32).reshape(1, F, h.shape[2], h.shape[3]).clone()
    g1, be1 = pt[f"{prefix}.conv1.1.weight"], pt[f"{prefix}.conv1.1.bias"]
    g2, be2 = pt[f"{prefix}.conv2.1.weight"], pt[f"{prefix}.conv2.1.bias"]
    gates = Fn.group_norm(raw, 2 * F // 32, g1, be1, torch_ref.EPS)
    z = torch.sigmoid(gates[:, :F])
    n = torch.tanh(Fn.group_norm(cx, F // 32, g2, be2, torch_ref.EPS))
    return (1 - z) * h + z * n


def f16_pair_weights(w):
    """the value the f16 x 3 k-loop multiplies by: hi + lo of w * 2^10 as two RNE f16 pieces (urnn_common.h), back in float32 (exact)"""
    ws_ = w.double() * 1024.0
    hi = ws_.to(torch.float16)
    lo = (ws_ - hi.double()).to(torch.float16)
    return ((hi.double() + lo.double()) / 1024.0).float()


def f16_pair_act(x, scale):
    """what the f16 x 3 k-loop multiplies instead of an activation x: fl32(x * scale) as hi + lo RNE f16 pieces, over scale"""
    xs = (x * scale)                                 # float32 product (exact for a power of two)
    hi = xs.to(torch.float16)
    lo = (xs - hi.float()).to(torch.float16)
    return ((hi.double() + lo.double()) / scale).float()


def conv_f16x3(x, w, b, scale, drop=True):
    """what the f16 x 3 k-loop computes, in float64: (w_hi + w_lo)(x_hi + x_lo) WITHOUT the lo * lo products (three MFMAs per 16 k: lo*hi,
    hi*lo, hi*hi), then one float32 rounding (the accumulate's own roundings are left out: this isolates the dropped term)"""
    xs = (x * scale)
    xh = xs.to(torch.float16)
    xl = (xs - xh.float()).to(torch.float16)
    ws_ = w.double() * 1024.0
    wh = ws_.to(torch.float16)
    wl = (ws_ - wh.double()).to(torch.float16)
    xh, xl, wh, wl = xh.double(), xl.double(), wh.double(), wl.double()
    acc = Fn.conv2d(xh, wh) + Fn.conv2d(xh, wl) + Fn.conv2d(xl, wh)
    if not drop:
        acc = acc + Fn.conv2d(xl, wl)
    return (acc / (scale * 1024.0) + b.double()[None, :, None, None]).float()


def cell_f16x3(prefix, x, e, h, drop):
    F = h.shape[1]
    W1, b1, g1, be1 = (pt[f"{prefix}.conv1.{k}"] for k in ("0.weight", "0.bias", "1.weight", "1.bias"))
    W2, b2, g2, be2 = (pt[f"{prefix}.conv2.{k}"] for k in ("0.weight", "0.bias", "1.weight", "1.bias"))
    cat = lambda *t: torch.cat([u for u in t if u is not None], dim=1)
    gates = Fn.group_norm(conv_f16x3(cat(x, e, h), W1, b1, 32.0, drop), 2 * F // 32, g1, be1, torch_ref.EPS)
    z, r = torch.sigmoid(gates[:, :F]), torch.sigmoid(gates[:, F:])
    n = torch.tanh(Fn.group_norm(conv_f16x3(cat(x, e, r * h), W2, b2, 32.0, drop), F // 32, g2, be2, torch_ref.EPS))
    return (1 - z) * h + z * n


def cell_repr(prefix, x, e, h, scale):
    """torch cell whose two convs see the f16-pair representation of their inputs (and weights)"""
    F = h.shape[1]
    W1, b1, g1, be1 = (pt[f"{prefix}.conv1.{k}"] for k in ("0.weight", "0.bias", "1.weight", "1.bias"))
    W2, b2, g2, be2 = (pt[f"{prefix}.conv2.{k}"] for k in ("0.weight", "0.bias", "1.weight", "1.bias"))
    cat = lambda *t: torch.cat([u for u in t if u is not None], dim=1)
    q = lambda v: f16_pair_act(v, scale)
    gates = Fn.group_norm(Fn.conv2d(q(cat(x, e, h)), f16_pair_weights(W1), b1), 2 * F // 32, g1, be1, torch_ref.EPS)
    z, r = torch.sigmoid(gates[:, :F]), torch.sigmoid(gates[:, F:])
    n = torch.tanh(Fn.group_norm(Fn.conv2d(q(cat(x, e, r * h)), f16_pair_weights(W2), b2), F // 32, g2, be2, torch_ref.EPS))
    return (1 - z) * h + z * n


FRAME = [0]


def run(which, gn_mode=None):
    """which: set of parts computed by the HIP modules: 'stage1', 'cells_full' (enc1, dec1), 'cells_rest', 'convs' (stage 2/3 convs,
    deconvs, decoder stage 1), 'head'."""
    st = [torch.zeros(s, device=dev) for s in shapes]
    errs = []
    with torch.no_grad():
        for t in range(N):
            x = preprocess_inputs(t, ev, dev, nums=NUMS, rain_max=RAIN_MAX, cumsum_rain_max=CUM_MAX)[:, 0].contiguous()
            e1, e2, e3, d1, d2, d3 = st
            FRAME[0] = t
            hc = lambda key, mod, name, xx, ee, hh: (cell_f16x3(name, xx, ee, hh, True) if (gn_mode == "x3_drop" and key == "cells_full") else cell_f16x3(name, xx, ee, hh, False) if (gn_mode == "x3_full" and key == "cells_full") else cell_repr(name, xx, ee, hh, 32.0) if (gn_mode == "repr" and key == "cells_full") else cell_repr(name, xx, ee, hh, 32.0 + 2.0 * (FRAME[0] & 7)) if (gn_mode == "repr_dither" and key == "cells_full") else cell_hip_gates(mod, name, xx.contiguous(), None if ee is None else ee.contiguous(), hh.contiguous()) if (gn_mode == "hip_gates" and key == "cells_full") else cell_hip_gates_cand(mod, name, xx.contiguous(), None if ee is None else ee.contiguous(), hh.contiguous()) if (gn_mode == "hip_gates_cand" and key == "cells_full") else mod.step(xx, ee, hh) if key in which else
                                                     (cell_custom(name, xx, ee, hh, gn_mode) if (gn_mode and key == "cells_full") else torch_ref.cell(pt, name, xx, ee, hh)))
            a1 = enc.stage1(x) if "stage1" in which else conv_t("encoder.stage1.conv1_leaky_1", x)
            e1n = hc("cells_full", enc.rnn1, "encoder.rnn1", a1, None, e1)
            a2 = enc.stage2(e1n) if "convs" in which else Fn.avg_pool2d(conv_t("encoder.stage2.conv2_leaky_1", e1n), 2)
            e2n = hc("cells_rest", enc.rnn2, "encoder.rnn2", a2, None, e2)
            a3 = enc.stage3(e2n) if "convs" in which else Fn.avg_pool2d(conv_t("encoder.stage3.conv3_leaky_1", e2n), 2)
            e3n = hc("cells_rest", enc.rnn3, "encoder.rnn3", a3, None, e3)
            d1n = hc("cells_rest", dec.rnn3, "decoder.rnn3", None, e3n, d1)
            u3 = dec.stage3(d1n) if "convs" in which else dc_t("decoder.stage3.deconv1_leaky_1", d1n)
            d2n = hc("cells_rest", dec.rnn2, "decoder.rnn2", u3, e2n, d2)
            u2 = dec.stage2(d2n) if "convs" in which else dc_t("decoder.stage2.deconv2_leaky_1", d2n)
            d3n = hc("cells_full", dec.rnn1, "decoder.rnn1", u2, e1n, d3)
            feat = dec.stage1(d3n) if "convs" in which else conv_t("decoder.stage1.conv3_leaky_1", d3n)
            if "head" in which:
                _, _, raw = net.head.run(feat.contiguous(), want_raw=True)
            else:
                _, _, raw = torch_ref.head(pt, feat, H, W)
            st = [v.contiguous() for v in (e1n, e2n, e3n, d1n, d2n, d3n)]
            errs.append(rel_err(raw.cpu().numpy().reshape(1, H, W), o_raw[t]))
    return np.array(errs)


def engine():
    eng = RolloutEngine(net, H, W, NUMS, RAIN_MAX, CUM_MAX, max_frames=T_EVENT, keep_raw=True, overlap=True, use_graph=True)
    eng.load_event(ev)
    eng.reset()
    eng.run(N)
    torch.cuda.synchronize()
    raw = eng.out_raw[:N].cpu().numpy()
    return np.array([rel_err(raw[t].reshape(1, H, W), o_raw[t]) for t in range(N)])


def show(name, e):
    lo = min(50, N - 1)
    print(f"{name:58s} worst {e.max():.2e} (frame {int(e.argmax()):3d}); mean frames {lo}..{N - 1}: {e[lo:].mean():.2e}; frames 0..39 max {e[:40].max():.2e}", flush=True)


ALL = {"stage1", "cells_full", "cells_rest", "convs", "head"}
show("engine (scalar-rain fold, fused cells, graph)", engine())
show("all torch-fp32", run(set()))
variants = [("all torch-fp32", set()), ("all HIP modules (generic stage 1, three-pass cells)", ALL),
            ("HIP: head only", {"head"}), ("HIP: stage 1 only", {"stage1"}), ("HIP: stage convs / deconvs only", {"convs"}),
            ("HIP: full-resolution cells only", {"cells_full"}), ("HIP: half / quarter-resolution cells only", {"cells_rest"}),
            ("HIP: everything but the cells", ALL - {"cells_full", "cells_rest"})]
show("torch, but the full-res cells' raw gates AND raw candidate from HIP", run(set(), "hip_gates_cand"))
show("HIP: full-resolution cells only (three-pass)", run({"cells_full"}))
if a.variants == "x3":
    show("torch, full-res cells' convs = float64 sum of hi*hi + hi*lo + lo*hi (lo*lo DROPPED)", run(set(), "x3_drop"))
    show("torch, full-res cells' convs = float64 sum of all four piece products", run(set(), "x3_full"))
if a.variants == "repr":
    show("torch, full-res cells' conv inputs + weights as f16 hi + lo pairs (scale 32)", run(set(), "repr"))
    show("torch, same with the scale 32 + 2 (t mod 8) changing every frame", run(set(), "repr_dither"))
if a.variants == "weights":
    keep = {k: v.clone() for k, v in pt.items()}
    for label, keys in (("candidate conv (W2) of enc1 / dec1", ["encoder.rnn1.conv2.0.weight", "decoder.rnn1.conv2.0.weight"]),
                        ("gate conv (W1) of enc1 / dec1", ["encoder.rnn1.conv1.0.weight", "decoder.rnn1.conv1.0.weight"]),
                        ("every conv / deconv weight of the network", [k for k in pt if k.endswith(".weight") and pt[k].dim() == 4 and not k.startswith("head.")])):
        for k in keys:
            pt[k] = f16_pair_weights(keep[k])
        rel = max(float(((pt[k] - keep[k]).abs().max() / keep[k].abs().max())) for k in keys)
        show(f"all torch-fp32, weights as hi + lo f16: {label} (max rel change {rel:.1e})", run(set()))
        for k in keys:
            pt[k] = keep[k]
if a.variants == "gn":
    for mode in ("fold", "fold_consistent", "centred"):
        show(f"all torch-fp32, full-res cells' GroupNorm affine: {mode}", run(set(), mode))
if a.variants == "all":
    for name, which in variants:
        show(name, run(which))
print(f"# {time.time() - t0:.0f} s")
