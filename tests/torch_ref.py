"""Plain-PyTorch restatement of one U-RNN timestep for tests that need autograd or a GPU-speed float64 reference at grids
where the C oracle is too slow (test infrastructure, like oracle/: never imported by the product).  Written from the equations
of DESIGN.md section 1 with the torch.nn.functional ops the reference's modules wrap (ConvRNN.py:94-104,150-189;
utils.py:87-124; network_blocks.py:74-101,129-171; flood_head.py:131-202); parameters are addressed by the reference's
state_dict names.  Pinned against the reference-generated goldens by tests/test_oracle.py::test_torch_ref_vs_reference."""
import torch
import torch.nn.functional as Fn

SLOPE = 0.2
EPS = 1e-5


def _lrelu(u, like=None):
    """LeakyReLU(0.2).  ``like``: a tensor whose SIGN decides the branch instead of u's own (the fp32 path's output: a value
    within fp32 rounding of the kink may legitimately sit on the other side there, and one such pixel moves a weight gradient
    by O(1e-3) of its size)."""
    if like is None:
        return Fn.leaky_relu(u, SLOPE)
    return u * torch.where(like > 0, 1.0, SLOPE).to(u.dtype)


def cell(p, prefix, x, e, h):
    """ConvGRU (e None) / Skip-ConvGRU step.  x may be None: a block of zeros (ConvRNN.py:143-146)."""
    F = h.shape[1]
    W1, b1, g1, be1 = (p[f"{prefix}.conv1.{k}"] for k in ("0.weight", "0.bias", "1.weight", "1.bias"))
    W2, b2, g2, be2 = (p[f"{prefix}.conv2.{k}"] for k in ("0.weight", "0.bias", "1.weight", "1.bias"))
    if x is None:
        x = torch.zeros((h.shape[0], W1.shape[1] - F - (F if e is not None else 0)) + tuple(h.shape[2:]), dtype=h.dtype, device=h.device)
    cat = lambda *t: torch.cat([u for u in t if u is not None], dim=1)
    gates = Fn.group_norm(Fn.conv2d(cat(x, e, h), W1, b1), 2 * F // 32, g1, be1, EPS)
    z, r = torch.sigmoid(gates[:, :F]), torch.sigmoid(gates[:, F:])
    n = torch.tanh(Fn.group_norm(Fn.conv2d(cat(x, e, r * h), W2, b2), F // 32, g2, be2, EPS))
    return (1 - z) * h + z * n


def head(p, feat, H, W, hip=None, cls_thred=0.5):
    """-> (masked, cls, raw).  hip: optional dict with the fp32 path's 'raw' and 'cls' (B,H,W): their LeakyReLU branch / wet-dry
    decision is used instead of the restatement's own (flip tolerance)."""
    C = feat.shape[1]

    def base(x, name):
        u = Fn.conv2d(x, p[f"head.{name}.conv.weight"])
        return Fn.silu(Fn.layer_norm(u, (C, H, W), p[f"head.{name}.ln.weight"], p[f"head.{name}.ln.bias"], EPS))
    t = base(feat, "stems")
    c = base(base(t, "cls_convs.0"), "cls_convs.1")
    q = base(base(t, "reg_convs.0"), "reg_convs.1")
    cls = torch.sigmoid(Fn.conv2d(c, p["head.cls_preds.conv.weight"], p["head.cls_preds.conv.bias"]))[:, 0]
    pre = Fn.conv2d(q, p["head.reg_preds.conv.weight"], p["head.reg_preds.conv.bias"])[:, 0]
    raw = _lrelu(pre, None if hip is None else hip["raw"])
    wet = ((cls if hip is None else hip["cls"]) >= cls_thred).to(raw.dtype)
    return raw * wet, cls, raw


def step(p, x_in, states, H, W, hip=None):
    """One ED.forward (model.py:65-121) on an assembled input x_in (B,C,H,W).  states = [e1,e2,e3,d1,d2,d3] with d1 the
    deepest.  hip: optional dict of the fp32 path's activations of the same step (keys a1, u3, u2, feat, raw, cls) used for
    branch decisions only.  Returns (masked, cls, raw, new states)."""
    g = (lambda k: None) if hip is None else (lambda k: hip.get(k))
    e1, e2, e3, d1, d2, d3 = states
    conv = lambda name, x: Fn.conv2d(x, p[name + ".weight"], p[name + ".bias"])
    a1 = _lrelu(conv("encoder.stage1.conv1_leaky_1", x_in), g("a1"))
    e1n = cell(p, "encoder.rnn1", a1, None, e1)
    a2 = Fn.avg_pool2d(_lrelu(conv("encoder.stage2.conv2_leaky_1", e1n)), 2)
    e2n = cell(p, "encoder.rnn2", a2, None, e2)
    a3 = Fn.avg_pool2d(_lrelu(conv("encoder.stage3.conv3_leaky_1", e2n)), 2)
    e3n = cell(p, "encoder.rnn3", a3, None, e3)
    d1n = cell(p, "decoder.rnn3", None, e3n, d1)
    dc = lambda name, x: Fn.conv_transpose2d(x, p[name + ".weight"], p[name + ".bias"], stride=2)
    u3 = _lrelu(dc("decoder.stage3.deconv1_leaky_1", d1n), g("u3"))
    d2n = cell(p, "decoder.rnn2", u3, e2n, d2)
    u2 = _lrelu(dc("decoder.stage2.deconv2_leaky_1", d2n), g("u2"))
    d3n = cell(p, "decoder.rnn1", u2, e1n, d3)
    feat = _lrelu(conv("decoder.stage1.conv3_leaky_1", d3n), g("feat"))
    masked, cls, raw = head(p, feat, H, W, hip)
    return masked, cls, raw, [e1n, e2n, e3n, d1n, d2n, d3n]


def wmse(reg, tgt):
    """The differentiable part of FocalBCE_and_WMSE (losses.py:150-189): 20 * mean(err^2 | wet) + mean(err^2 | dry)."""
    wet = tgt > 0
    err2 = (reg - tgt) ** 2
    return 20.0 * err2[wet].mean() + err2[~wet].mean()
