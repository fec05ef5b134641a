"""Training building blocks at BASELINE's full grid (500x500 and its 250x250 / 125x125 levels -- the last one an odd plane of
15 625 pixels, i.e. the unaligned load paths), where the numpy oracle is too slow: the reference of each test is the same
mathematics written with plain torch ops in FLOAT64 on the GPU (F.conv2d / F.group_norm / F.avg_pool2d / F.conv_transpose2d,
the modules the reference network is made of: ConvRNN.py:73-194, utils.py:73-125) and differentiated by autograd."""
import numpy as np
import pytest
import torch
import torch.nn.functional as Fn

from conftest import assert_close

pytestmark = pytest.mark.gpu
GRAD_TOL = 2e-4     # same bar as tests/test_hip_train.py: relative to each tensor's max (floor 0.1 * max, conftest.rel_err)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _off_kink(dy, u64, pool=False):
    """LeakyReLU'(u) jumps at u = 0: where the float64 pre-activation is within fp32 rounding of zero the fp32 path may
    legitimately take the other slope (a handful of values in 10^7, but each moves a weight gradient by O(1)).  The upstream
    gradient is zeroed there, so neither path depends on the branch (cf. the threshold flips of conftest.masked_parity)."""
    near = (u64.detach().abs() < 1e-5).double()
    near = Fn.max_pool2d(near, 2) if pool else near
    assert float(near.mean()) < 1e-3
    return (dy * (1.0 - near).to(dy.dtype)).contiguous()


def _rand(gen, *shape, scale=1.0, dev=None):
    return (torch.randn(*shape, generator=gen, device=dev) * scale).contiguous()


def _cell64(x, e, h, W1, b1, g1, be1, W2, b2, g2, be2, dout, eps=1e-5):
    """CGRU_cell step (ConvRNN.py:150-189) in float64 with autograd; returns (h', {gradients})."""
    d = lambda t: None if t is None else t.double().clone().requires_grad_(True)
    xs, es, hs = d(x), d(e), d(h)
    P = {k: d(v) for k, v in dict(W1=W1, b1=b1, g1=g1, be1=be1, W2=W2, b2=b2, g2=g2, be2=be2).items()}
    F = h.shape[1]
    I = W1.shape[1] - F - (F if e is not None else 0)
    # a missing input (decoder stage 3) is a block of zeros (ConvRNN.py:136)
    x_in = xs if xs is not None else torch.zeros((h.shape[0], I) + tuple(h.shape[2:]), dtype=torch.float64, device=h.device)
    cat = lambda *t: torch.cat([u for u in t if u is not None], dim=1)
    gates = Fn.group_norm(Fn.conv2d(cat(x_in, es, hs), P["W1"], P["b1"]), 2 * F // 32, P["g1"], P["be1"], eps)
    z, r = torch.sigmoid(gates[:, :F]), torch.sigmoid(gates[:, F:])
    n = torch.tanh(Fn.group_norm(Fn.conv2d(cat(x_in, es, r * hs), P["W2"], P["b2"]), F // 32, P["g2"], P["be2"], eps))
    out = (1 - z) * hs + z * n
    (out * dout.double()).sum().backward()
    g = {"dh": hs.grad, "dW1": P["W1"].grad, "db1": P["b1"].grad, "dg1": P["g1"].grad, "dbe1": P["be1"].grad,
         "dW2": P["W2"].grad, "db2": P["b2"].grad, "dg2": P["g2"].grad, "dbe2": P["be2"].grad}
    if xs is not None:
        g["dx"] = xs.grad
    if es is not None:
        g["de"] = es.grad
    return out.detach(), g


@pytest.mark.parametrize("name,I,F,skip,with_x,H,W", [
    ("decoder.rnn1", 96, 64, 1, 1, 500, 500),        # K = 224: the largest GEMMs of the network
    ("encoder.rnn1", 16, 64, 0, 1, 500, 500),
    ("decoder.rnn2", 96, 96, 1, 1, 250, 250),        # 192 x 288 weight gradient
    ("decoder.rnn3", 96, 96, 1, 0, 125, 125),        # x = None, odd plane
    ("encoder.rnn3", 96, 96, 0, 1, 125, 125),
])
def test_cell_backward_full_size(dev, name, I, F, skip, with_x, H, W):
    from urnn_amd import ops, train_ops
    gen = torch.Generator(device=dev).manual_seed(1000 + I + F + H)
    K = I + (2 * F if skip else F)
    W1, W2 = _rand(gen, 2 * F, K, 1, 1, scale=K ** -0.5, dev=dev), _rand(gen, F, K, 1, 1, scale=K ** -0.5, dev=dev)
    b1, b2, be1, be2 = (_rand(gen, n, scale=0.1, dev=dev) for n in (2 * F, F, 2 * F, F))
    g1 = (torch.rand(2 * F, generator=gen, device=dev) + 0.5).contiguous()
    g2 = (torch.rand(F, generator=gen, device=dev) + 0.5).contiguous()
    x = _rand(gen, 1, I, H, W, dev=dev) if with_x else None
    e = _rand(gen, 1, F, H, W, scale=0.5, dev=dev) if skip else None
    h = _rand(gen, 1, F, H, W, scale=0.5, dev=dev)
    dout = _rand(gen, 1, F, H, W, dev=dev)
    packed = ops.pack_gru(W1, b1, W2, b2, I, F, bool(skip))
    ws = ops.workspace(ops.gru_cell_workspace_bytes(1, F, H, W), dev)
    out = ops.gru_cell(x, e, h, packed, g1, be1, g2, be2, I, ws=ws)
    got = train_ops.gru_cell_backward(x, e, h, W1, W2, g1, g2, dout, I, ws)
    want_out, want = _cell64(x, e, h, W1, b1, g1, be1, W2, b2, g2, be2, dout)
    assert_close(out.cpu().numpy(), want_out.cpu().numpy(), 1e-4, f"{name}: forward")
    for k, ref in want.items():
        assert_close(got[k].reshape(ref.shape).cpu().numpy(), ref.cpu().numpy(), GRAD_TOL, f"{name} {H}x{W}: {k}")


@pytest.mark.parametrize("name,Cin,Cout,pool,H,W", [("encoder.stage1", 63, 16, 0, 500, 500), ("encoder.stage2", 64, 64, 1, 500, 500),
                                                    ("encoder.stage3", 96, 96, 1, 250, 250), ("decoder.stage1", 64, 16, 0, 500, 500)])
def test_stage_conv_backward_full_size(dev, name, Cin, Cout, pool, H, W):
    from urnn_amd import ops, train_ops
    gen = torch.Generator(device=dev).manual_seed(2000 + Cin + Cout + H)
    w, b = _rand(gen, Cout, Cin, 1, 1, scale=Cin ** -0.5, dev=dev), _rand(gen, Cout, scale=0.1, dev=dev)
    x = _rand(gen, 1, Cin, H, W, dev=dev)
    y = ops.stage_conv(x, ops.pack_conv(w, b), Cout, bool(pool))
    xs, ws, bs = (t.double().clone().requires_grad_(True) for t in (x, w, b))
    u = Fn.conv2d(xs, ws, bs)
    ref = Fn.leaky_relu(u, 0.2)
    ref = Fn.avg_pool2d(ref, 2) if pool else ref
    dy = _off_kink(_rand(gen, *y.shape, dev=dev), u, bool(pool))
    dx, dw, db = train_ops.stage_conv_backward(x, w, b, dy, bool(pool))
    (ref * dy.double()).sum().backward()
    assert_close(y.cpu().numpy(), ref.detach().cpu().numpy(), 1e-4, f"{name}: forward")
    for k, a, r in (("dx", dx, xs.grad), ("dw", dw, ws.grad), ("db", db, bs.grad)):
        assert_close(a.cpu().numpy(), r.cpu().numpy(), GRAD_TOL, f"{name} {H}x{W}: {k}")


@pytest.mark.parametrize("name,Cin,Cout,H,W", [("decoder.stage2", 96, 96, 250, 250), ("decoder.stage3", 96, 96, 125, 125)])
def test_deconv_backward_full_size(dev, name, Cin, Cout, H, W):
    from urnn_amd import ops, train_ops
    gen = torch.Generator(device=dev).manual_seed(3000 + Cin + H)
    w, b = _rand(gen, Cin, Cout, 2, 2, scale=Cin ** -0.5, dev=dev), _rand(gen, Cout, scale=0.1, dev=dev)
    x = _rand(gen, 1, Cin, H, W, dev=dev)
    y = ops.deconv2x2(x, ops.pack_deconv(w, b), Cout)
    xs, ws, bs = (t.double().clone().requires_grad_(True) for t in (x, w, b))
    u = Fn.conv_transpose2d(xs, ws, bs, stride=2)
    ref = Fn.leaky_relu(u, 0.2)
    dy = _off_kink(_rand(gen, *y.shape, dev=dev), u)
    dx, dw, db = train_ops.deconv2x2_backward(x, w, y, dy)
    (ref * dy.double()).sum().backward()
    assert_close(y.cpu().numpy(), ref.detach().cpu().numpy(), 1e-4, f"{name}: forward")
    for k, a, r in (("dx", dx, xs.grad), ("dw", dw, ws.grad), ("db", db, bs.grad)):
        assert_close(a.cpu().numpy(), r.cpu().numpy(), GRAD_TOL, f"{name} {H}x{W}: {k}")


def test_head_backward_full_size(dev):
    """YOLOXHead (flood_head.py:131-202) at 500x500: BaseConv = conv1x1 (no bias) + LayerNorm([16,H,W]) + SiLU; the gradient
    flows through the regression branch only (the mask and classify_outputs are comparisons).  The float64 restatement uses
    the HIP forward's own wet/dry mask and LeakyReLU branch of the prediction layer, so that threshold / kink flips of single
    pixels (conftest.masked_parity) cannot enter the comparison."""
    from urnn_amd import ops, train_ops
    H = W = 500
    C = 16
    gen = torch.Generator(device=dev).manual_seed(4242)
    conv_w = _rand(gen, 5, C, C, scale=0.4, dev=dev)
    ln_w = (torch.rand(5, C, H, W, generator=gen, device=dev) + 0.5).contiguous()
    ln_b = _rand(gen, 5, C, H, W, scale=0.1, dev=dev)
    cls_w, cls_b, reg_w, reg_b = _rand(gen, C, scale=0.3, dev=dev), _rand(gen, 1, scale=0.1, dev=dev), _rand(gen, C, scale=0.3, dev=dev), \
        _rand(gen, 1, scale=0.1, dev=dev)
    feat = _rand(gen, 1, C, H, W, dev=dev)
    dout = _rand(gen, 1, H, W, dev=dev)
    ws = ops.workspace(ops.head_workspace_bytes(1, C, H, W), dev)
    masked, cls, raw = ops.head(feat, conv_w, ln_w, ln_b, cls_w, cls_b, reg_w, reg_b, 0.5, want_raw=True, ws=ws)
    g = train_ops.head_backward(feat, conv_w, ln_w, ln_b, reg_w, raw, cls, dout, 0.5, ws)

    d = lambda t: t.double().clone().requires_grad_(True)
    f64, cw, lw, lb, rw, rb = d(feat), d(conv_w), d(ln_w), d(ln_b), d(reg_w), d(reg_b)

    def base(x, i):
        u = Fn.conv2d(x, cw[i].reshape(C, C, 1, 1))
        return Fn.silu(Fn.layer_norm(u, (C, H, W), lw[i], lb[i], 1e-5))
    q = base(base(base(f64, 0), 3), 4)
    pre = Fn.conv2d(q, rw.reshape(1, C, 1, 1), rb)[:, 0]
    slope = torch.where(raw > 0, 1.0, 0.2).double()            # the branch the fp32 forward took
    reg = pre * slope
    wet = (cls >= 0.5).double()
    assert_close((reg * wet).detach().cpu().numpy(), masked.cpu().numpy(), 1e-4, "head forward")
    (reg * wet * dout.double()).sum().backward()
    assert_close(g["dfeat"].cpu().numpy(), f64.grad.cpu().numpy(), GRAD_TOL, "head 500x500: dfeat")
    for i, n in ((0, "stems"), (3, "reg_convs.0"), (4, "reg_convs.1")):
        assert_close(g["dconv_w"][i].cpu().numpy(), cw.grad[i].cpu().numpy(), GRAD_TOL, f"head 500x500: d{n}.conv")
        assert_close(g["dln_w"][i].cpu().numpy(), lw.grad[i].cpu().numpy(), GRAD_TOL, f"head 500x500: d{n}.ln.weight")
        assert_close(g["dln_b"][i].cpu().numpy(), lb.grad[i].cpu().numpy(), GRAD_TOL, f"head 500x500: d{n}.ln.bias")
    for i in (1, 2):
        assert float(g["dconv_w"][i].abs().max()) == 0.0 and float(g["dln_w"][i].abs().max()) == 0.0
    assert_close(g["dreg_w"].cpu().numpy(), rw.grad.cpu().numpy(), GRAD_TOL, "head 500x500: dreg_w")
    assert_close(g["dreg_b"].cpu().numpy(), rb.grad.cpu().numpy(), GRAD_TOL, "head 500x500: dreg_b")


def _bench_net(dev, H, W, C, seed=0):
    import urnn_amd.weights as uw
    from urnn_amd.net_config import load_net_config
    from urnn_amd.networks import ED, get_network_params
    sd = uw.make_state_dict(H, W, C, seed=seed)
    ep, dp = get_network_params(False, H, W, C, load_net_config())
    net = ED(False, ep, dp, 0.5, False, H, W)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return net.to(dev).eval(), sd


def _train_inputs(dev, H, W, rain_max, frames, seed=7, batch=1):
    import urnn_amd.weights as uw
    ev = uw.make_event(frames, H, W, rain_max, seed=42, batch=batch)
    g = torch.Generator(device=dev).manual_seed(seed)
    label = torch.rand(batch, frames, H, W, device=dev, generator=g) ** 3
    label[label < 0.1] = 0
    return ev, label


def _zero_states(dev, B, H, W):
    from oracle import oracle as orc
    return [torch.zeros(s.shape, device=dev) for s in orc.zero_states(B, H, W)]


def _window_setup(dev, H, W, nums, rain_max, cum_max, steps, t0, B):
    """Net, event, labels and the states after a gradient-free pre-roll of t0 frames (the window starts from real states)."""
    from urnn_amd.dataset import event_to_device
    from urnn_amd.training import WindowGradients
    net, sd = _bench_net(dev, H, W, 2 * nums + 3)
    ev_np, label = _train_inputs(dev, H, W, rain_max, t0 + steps, batch=B)
    ev = event_to_device(ev_np, dev)
    wg = WindowGradients(net, H, W, nums, rain_max, cum_max)
    states = _zero_states(dev, B, H, W)
    for t in range(t0):
        _, states = wg._forward_step(ev, t, states, 0)
    return net, sd, ev, label, wg, [s.clone() for s in states]


@pytest.mark.parametrize("name,H,W,nums,rain_max,cum_max,steps,t0,B", [
    # BASELINE configs[3]'s grid (location1: 500x500, historical_nums 30 -> C = 63; location1_scratch.yaml:56-58)
    ("location1", 500, 500, 30, 6.0, 250.0, 4, 2, 1),
    # BASELINE configs[2]: the lightweight grid with 8 events per GPU and the reference's seq_num = 12 windows (lite.yaml:31-36,56-58)
    ("lite128xB8", 128, 128, 3, 60.0, 250.0, 12, 2, 8),
])
def test_window_full_size(dev, name, H, W, nums, rain_max, cum_max, steps, t0, B):
    """One SWP window (main.py:598-768) at a BASELINE training configuration's own size, from non-zero states, through
    `WindowGradients.run`: every one of the 79 parameter gradients against float64 torch autograd of tests/torch_ref.py on the
    GPU (the restatement takes the fp32 path's LeakyReLU / wet-dry branches where that path materialises the activation, so
    threshold pixels cannot enter), and everything finite."""
    import torch_ref
    net, sd, ev, label, wg, start = _window_setup(dev, H, W, nums, rain_max, cum_max, steps, t0, B)
    out = wg.run(ev, label[:, t0:t0 + steps], t0, steps, states=[s.clone() for s in start])
    torch.cuda.synchronize()
    for pname, gr in out["grads"].items():
        assert bool(torch.isfinite(gr).all()), f"non-finite gradient in {pname}"

    # float64 autograd of the same window; branch decisions from the fp32 forward of each step
    p = {k: torch.from_numpy(v).to(dev).double().requires_grad_(True) for k, v in sd.items()}
    st64 = [s.double() for s in start]
    regs, st_hip = [], [s.clone() for s in start]
    for s in range(steps):
        S, st_hip = wg._forward_step(ev, t0 + s, st_hip, 0)
        hip = {k: S[k].double() for k in ("a1", "u3", "u2", "feat", "raw", "cls")}
        masked, _, _, st64 = torch_ref.step(p, S["x_in"].double(), st64, H, W, hip=hip)
        assert_close(masked.detach().cpu().numpy(), S["masked"].cpu().numpy(), 1e-4, f"{name} step {s}: forward output")
        regs.append(masked)
    loss = torch_ref.wmse(torch.stack(regs, dim=1), label[:, t0:t0 + steps].double())
    loss.backward()
    assert float(out["loss"][1]) == pytest.approx(float(loss), rel=2e-4)
    worst = ("", 0.0)
    for pname in sd:
        ref = p[pname].grad
        got = out["grads"][pname].double().reshape(p[pname].shape)
        if ref is None:                                   # classification branch: cut off by the wet/dry comparison
            assert float(got.abs().max()) == 0.0, pname
            continue
        scale = float(ref.abs().max())
        err = float((got - ref).abs().max()) / max(scale, 1e-30)
        worst = max(worst, (pname, err), key=lambda t: t[1])
        assert err <= 1e-3, (pname, err, scale)           # the bar of the 16x16 window test (test_hip_train.py)
    print(f"worst gradient, {name} {H}x{W} B={B} seq {steps}:", worst)


def test_bf16_window_full_size(dev):
    """BASELINE configs[3]'s bf16 arm at its own grid (500x500, C = 63): the reference only declares --amp (config.py:179), so
    the bf16 compute mode (urnn_set_matrix_mode: GEMM operands rounded to bf16 -- forward, input- and weight-gradient GEMMs --
    fp32 accumulation, fp32 norms / states / loss / Adam) is judged against this build's own fp32 window, whose 79 gradients
    test_window_full_size pins to float64 autograd.  Stated bounds: bf16 keeps 8 significant bits (2^-9 = 2e-3 per rounded
    operand); through four recurrent steps and ~40 GEMMs the loss stays within 2 %, and every gradient TENSOR stays at cosine
    >= 0.99 of the fp32 gradient with a relative L2 error <= 15 % (single elements of the 4-million-element LayerNorm affines
    move by up to half of the tensor's max: measured 0.52 at head.stems.ln.bias; worst cosine 0.9971 and worst relative L2 0.076 at
    head.reg_convs.1.ln.bias, printed); the fp32 mode is
    bit-unchanged afterwards."""
    from urnn_amd import ops
    H = W = 500
    nums, rain_max, cum_max, steps, t0 = 30, 6.0, 250.0, 4, 2
    net, sd, ev, label, wg, start = _window_setup(dev, H, W, nums, rain_max, cum_max, steps, t0, 1)
    ref = wg.run(ev, label[:, t0:t0 + steps], t0, steps, states=[s.clone() for s in start])
    ref_g = {k: v.clone() for k, v in ref["grads"].items()}
    ref_loss = float(ref["loss"][1])
    with ops.matrix_mode("bf16"):
        low = wg.run(ev, label[:, t0:t0 + steps], t0, steps, states=[s.clone() for s in start])
        low_g = {k: v.clone() for k, v in low["grads"].items()}
        low_loss = float(low["loss"][1])
    again = wg.run(ev, label[:, t0:t0 + steps], t0, steps, states=[s.clone() for s in start])
    torch.cuda.synchronize()
    assert all(torch.equal(again["grads"][k], ref_g[k]) for k in ref_g), "fp32 mode must be bit-unchanged after the bf16 scope"
    assert low_loss == pytest.approx(ref_loss, rel=2e-2), (low_loss, ref_loss)
    worst_rel, worst_cos, worst_l2, differs = ("", 0.0), ("", 1.0), ("", 0.0), 0
    for k, r in ref_g.items():
        g = low_g[k]
        assert bool(torch.isfinite(g).all()), k
        scale = float(r.abs().max())
        if scale == 0.0:                                   # classification branch: exactly zero in both modes
            assert float(g.abs().max()) == 0.0, k
            continue
        rel = float((g - r).abs().max()) / scale
        cos = float((g.double() * r.double()).sum() / (g.double().norm() * r.double().norm()))
        l2 = float((g.double() - r.double()).norm() / r.double().norm())
        differs += int(rel > 1e-5)
        worst_rel = max(worst_rel, (k, rel), key=lambda t: t[1])
        worst_cos = min(worst_cos, (k, cos), key=lambda t: t[1])
        worst_l2 = max(worst_l2, (k, l2), key=lambda t: t[1])
        assert cos >= 0.99 and l2 <= 0.15 and rel <= 0.6, (k, rel, cos, l2)
    print(f"bf16 vs fp32 window at 500x500: loss {low_loss:.6f} vs {ref_loss:.6f}; worst |dg|/max {worst_rel}; worst cosine {worst_cos}; "
          f"worst relative L2 {worst_l2}")
    assert differs > 40, "the bf16 mode must actually change the arithmetic"


def test_trainer_graph_equals_eager_full_size(dev):
    """Three SWP windows of the training bench at 500x500 through `Trainer` with and without hipGraph capture: the clip
    coefficient / gradient norm of every window is finite and the two paths agree bit for bit (round 1 printed
    grad_norm = Infinity from the captured path: its memset nodes did not run on replay)."""
    from urnn_amd.training import Trainer
    H = W = 500
    nums, rain_max, cum_max, S, nwin = 30, 6.0, 250.0, 4, 3
    ev, label = _train_inputs(dev, H, W, rain_max, S * nwin)
    res = {}
    for graph in (False, True):
        net, _ = _bench_net(dev, H, W, 2 * nums + 3)
        tr = Trainer(net, H, W, nums, rain_max, cum_max, lr=1e-4, grad_clip=1.0, use_graph=graph)
        states, clips = None, []
        for w in range(nwin):
            loss, states = tr.train_window(ev, label[:, w * S:(w + 1) * S], w * S, S, states)
            torch.cuda.synchronize()
            clip = tr.last["clip"].cpu()
            assert bool(torch.isfinite(clip).all()) and 0.0 < float(clip[1]) < 1e3, (graph, w, clip)
            assert bool(torch.isfinite(tr.gflat).all()), (graph, w)
            clips.append(clip.clone())
        res[graph] = (clips, tr.flat.clone())
        del tr, net
    for a, b in zip(res[False][0], res[True][0]):
        assert torch.equal(a, b), (a, b)
    assert torch.equal(res[False][1], res[True][1])


def test_paper_mode_prewarm_vs_the_oracle_trace(dev):
    """Paper mode (main.py:542-595, 655-672) rolls gradient-free from frame 0 to a window's start THROUGH THE TRAINING FORWARD -- the
    three-pass cells, whose recurrent product W2[:, h].(r * h) stays on f16 pieces, where the inference engine's fused candidate
    kernel puts it on the fp32 matrix instruction.  The first 181 frames of BASELINE configs[1] (500x500, C = 63: up to
    and through the rain peak, where float32 roundoff is amplified most) through `Trainer`'s forward, against the committed oracle
    trace of the whole event (tests/golden/whole_event_500x500_T360.npz: random AND adversarial pixels of every sampled frame) under
    the rollout's own bar: every sampled frame within max(1e-4, 3 x what plain float32 torch is away from the oracle there).
    (Against the inference engine's states the two HIP paths differ by up to 4e-4 of a state's range at frame 120 -- both carry
    amplified roundoff, which is why the yardstick is the oracle and the reference's own arithmetic, not each other.)"""
    import os
    import urnn_amd.weights as uw
    from urnn_amd.dataset import event_to_device
    from urnn_amd.general import initialize_states
    from urnn_amd.training import Trainer
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "whole_event_500x500_T360.npz"))
    H, W, nums, T = int(g["H"]), int(g["W"]), int(g["nums"]), int(g["T"])
    rain_max, cum_max = float(g["rain_max"]), float(g["cumsum_max"])
    last = 180
    ev = event_to_device(uw.make_event(T, H, W, rain_max, seed=int(g["event_seed"])), dev)
    net, _ = _bench_net(dev, H, W, 2 * nums + 3, seed=int(g["weights_seed"]))
    tr = Trainer(net, H, W, nums, rain_max, cum_max, lr=1e-4, use_graph=False)
    states = [s.to(dev) for s in initialize_states(dev, H, W)]
    fr = {int(t): i for i, t in enumerate(g["frames"])}
    pix = torch.from_numpy(g["pixels"]).long().to(dev)

    def sub_err(got, want, plane_max):
        got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
        return float((np.abs(got - want) / np.maximum(np.abs(want), 0.1 * max(float(plane_max), 1e-30))).max())

    worst, wr, wt = 0.0, 0.0, 0.0
    for t in range(last + 1):
        S, states = tr.wg._forward_step(ev, t, states, 0)
        if t in fr:
            i = fr[t]
            adv = torch.from_numpy(g["adv_idx"][i].astype(np.int64)).to(dev)
            raw, cls = S["raw"].reshape(-1), S["cls"].reshape(-1)
            for got, want, yard, pmax in ((raw[pix], g["oracle_raw"][i], g["torch32_reg_err"][i], g["oracle_raw_plane_max"][i]),
                                          (raw[adv], g["adv_oracle_raw"][i], g["torch32_reg_err_adv"][i], g["oracle_raw_plane_max"][i]),
                                          (cls[pix], g["oracle_cls"][i], g["torch32_cls_err"][i], g["oracle_cls_plane_max"][i]),
                                          (cls[adv], g["adv_oracle_cls"][i], g["torch32_cls_err_adv"][i], g["oracle_cls_plane_max"][i])):
                e = sub_err(got.cpu().numpy(), want, pmax)
                worst = max(worst, e / max(1e-4, 3 * float(yard)))
            wr = max(wr, sub_err(raw[adv].cpu().numpy(), g["adv_oracle_raw"][i], g["oracle_raw_plane_max"][i]))
            wt = max(wt, float(g["torch32_reg_err_adv"][i]))
    print(f"training forward, frames 0-{last} vs the oracle trace: worst pre-mask regression on adversarial pixels {wr:.2e} "
          f"(torch-fp32 there {wt:.2e}); worst error / bar = {worst:.2f}")
    assert worst <= 1.0
