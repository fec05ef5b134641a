#!/usr/bin/env python
"""Goldens for the training rows, produced by running the REFERENCE with autograd in this container:

  * the loss `FocalBCE_and_WMSE` (losses.py:44-249) on seeded predictions/targets: components and d(loss)/d(reg);
  * one SWP window of `seq_num` = 2 timesteps from zero states (main.py:598-700 `process_window` semantics: pred["reg"] is the
    network output, pred["cls"] = where(output >= cls_thred, 1, 0), loss on the concatenated steps, one backward through both
    steps and the recurrent states): loss, the per-step outputs, and the gradient of every one of the 79 unique parameter
    tensors (`None` gradients -- the cls branch of the head is cut off by the non-differentiable mask -- are stored as flags).

Needs /root/reference; never shipped to the GPU box.  Usage: python tests/golden/make_train_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (reference imports, the .cuda shim, ref_net)

from src.lib.model.networks.losses import FocalBCE_and_WMSE  # noqa: E402
from src.lib.dataset.Dynamic2DFlood import preprocess_inputs  # noqa: E402
from src.lib.utils.general import initialize_states  # noqa: E402
import urnn_amd.weights as uw  # noqa: E402


def gen_loss(out):
    rs = np.random.RandomState(5)
    tgt = (rs.uniform(0, 1, (1, 3, 10, 12)) ** 4).astype(np.float32)
    tgt[tgt < 0.05] = 0.0                                      # ~half the cells dry, like a flood map
    reg = (tgt * rs.uniform(0.5, 1.4, tgt.shape) + (rs.uniform(0, 1, tgt.shape) > 0.8) * 0.02).astype(np.float32)
    for thr in (0.0, 0.01):
        r = torch.from_numpy(reg).clone().requires_grad_(True)
        cls = torch.where(r >= thr, 1, 0)
        res = FocalBCE_and_WMSE(gamma=2, alpha=0.25)({"reg": r, "cls": cls}, torch.from_numpy(tgt), epoch=0)
        res["loss"].backward()
        tag = f"loss_thr{thr}"
        for k, v in res.items():
            out[f"{tag}_{k}"] = np.float64(v.item())
        out[f"{tag}_dreg"] = r.grad.numpy()
    out["loss_reg"], out["loss_tgt"] = reg, tgt


def gen_window(out, H=16, W=16, nums=3, steps=2, wseed=21, eseed=8):
    C = 2 * nums + 3
    net, sd = mg.ref_net(H, W, C, wseed)
    net.train()                                               # nothing in the model depends on the mode (no dropout / BN)
    ev = uw.make_event(steps + 1, H, W, 60.0, seed=eseed)
    tev = mg.event_to_torch(ev)
    rs = np.random.RandomState(eseed)
    tgt = (rs.uniform(0, 1, (1, steps, H, W)) ** 3).astype(np.float32)
    tgt[tgt < 0.1] = 0.0
    states = initialize_states(torch.device("cpu"), input_height=H, input_width=W, net_cfg=mg.CFG)
    pred = None
    for t in range(steps):
        x = preprocess_inputs(t, tev, torch.device("cpu"), nums=nums, rain_max=60.0, cumsum_rain_max=250.0)
        res = net(x, *states)
        o, states = res[0], res[1:]
        cls = torch.where(o >= 0, 1, 0)
        pred = {"reg": o, "cls": cls} if pred is None else {"reg": torch.cat((pred["reg"], o), 1), "cls": torch.cat((pred["cls"], cls), 1)}
    res = FocalBCE_and_WMSE(gamma=2, alpha=0.25)(pred, torch.from_numpy(tgt), epoch=0)
    res["loss"].backward()
    out.update({"win_H": H, "win_W": W, "win_nums": nums, "win_steps": steps, "win_weights_seed": wseed, "win_event_seed": eseed,
                "win_rain_max": 60.0, "win_cumsum_max": 250.0, "win_target": tgt, "win_reg": pred["reg"].detach().numpy()})
    for k, v in res.items():
        out[f"win_{k}"] = np.float64(v.item())
    named = dict(net.named_parameters())                       # named_parameters() de-duplicates the alias registrations
    seen = set()
    for key in sd:                                            # the 79 canonical names
        p = named.get(key)
        if p is None:                                         # registered only under an alias name: find the same storage
            p = next(v for k, v in net.state_dict(keep_vars=True).items()
                     if k.replace("_wrapper.module.", ".").replace(".conv1_module.", ".conv1.").replace(".conv2_module.", ".conv2.") == key)
        assert id(p) not in seen
        seen.add(id(p))
        out[f"win_hasgrad_{key}"] = np.int32(p.grad is not None)
        if p.grad is not None:
            out[f"win_grad_{key}"] = p.grad.numpy()
    for i, s in enumerate(states):
        out[f"win_state{i}"] = s.detach().numpy()


def gen_cell_backward(out):
    """One ConvGRU / Skip-ConvGRU cell (ConvRNN.py:73-194, 1x1 gates) with seeded parameters and inputs: the gradients of a
    random upstream dL/dh' w.r.t. every input and parameter, from reference autograd."""
    from src.lib.model.networks.ConvRNN import CGRU_cell
    for tag, (I, F, module, H, W, B, with_x) in {
        "enc": (16, 64, "encoder", 16, 16, 1, True),
        "dec": (96, 64, "decoder", 8, 12, 2, True),
        "dec0": (96, 96, "decoder", 4, 4, 1, False),          # x == None (decoder stage 3)
    }.items():
        rs = np.random.RandomState(300 + I + F)
        cell = CGRU_cell(False, (H, W), I, 1, F, module)
        with torch.no_grad():
            for prm in cell.parameters():
                if prm.ndim == 4:
                    prm.copy_(torch.from_numpy(rs.normal(0, 1 / np.sqrt(prm.shape[1]), prm.shape).astype(np.float32)))
                else:
                    prm.copy_(torch.from_numpy(rs.uniform(0.5, 1.5, prm.shape).astype(np.float32)
                                               if prm is cell.conv1[1].weight or prm is cell.conv2[1].weight
                                               else rs.normal(0, 0.1, prm.shape).astype(np.float32)))
        x = torch.from_numpy(rs.normal(0, 1, (B, I, H, W)).astype(np.float32)).requires_grad_(True)
        h = torch.from_numpy(rs.normal(0, 0.5, (B, F, H, W)).astype(np.float32)).requires_grad_(True)
        e = torch.from_numpy(rs.normal(0, 0.5, (B, F, H, W)).astype(np.float32)).requires_grad_(True) if module == "decoder" else None
        hidden = torch.cat((e, h), 1) if e is not None else h
        y = cell(x[None] if with_x else None, hidden)[0]
        dout = torch.from_numpy(rs.normal(0, 1, y.shape).astype(np.float32))
        y.backward(dout)
        out.update({f"cell_{tag}_I": I, f"cell_{tag}_F": F, f"cell_{tag}_skip": int(e is not None), f"cell_{tag}_with_x": int(with_x),
                    f"cell_{tag}_x": x.detach().numpy(), f"cell_{tag}_h": h.detach().numpy(), f"cell_{tag}_dout": dout.numpy(),
                    f"cell_{tag}_out": y.detach().numpy(), f"cell_{tag}_dh": h.grad.numpy()})
        if with_x:
            out[f"cell_{tag}_dx"] = x.grad.numpy()
        if e is not None:
            out[f"cell_{tag}_e"], out[f"cell_{tag}_de"] = e.detach().numpy(), e.grad.numpy()
        for name, prm in (("W1", cell.conv1[0].weight), ("b1", cell.conv1[0].bias), ("g1", cell.conv1[1].weight), ("be1", cell.conv1[1].bias),
                          ("W2", cell.conv2[0].weight), ("b2", cell.conv2[0].bias), ("g2", cell.conv2[1].weight), ("be2", cell.conv2[1].bias)):
            out[f"cell_{tag}_{name}"] = prm.detach().numpy()
            out[f"cell_{tag}_d{name}"] = prm.grad.numpy()


def gen_layers_backward(out, H=16, W=16, C=9, seed=31):
    """Stage convs (conv1x1 + LeakyReLU [+ AvgPool]), a transposed conv stage and the head, each with a random upstream
    gradient, from reference autograd (modules of a seeded reference ED)."""
    net, sd = mg.ref_net(H, W, C, seed)
    rs = np.random.RandomState(seed)
    out.update({"lay_H": H, "lay_W": W, "lay_C": C, "lay_seed": seed})

    def one(tag, module, x_shape, params):
        for prm in module.parameters():
            prm.grad = None
        x = torch.from_numpy(rs.normal(0, 1, x_shape).astype(np.float32)).requires_grad_(True)
        y = module(x)
        dy = torch.from_numpy(rs.normal(0, 1, y.shape).astype(np.float32))
        y.backward(dy)
        out[f"lay_{tag}_x"], out[f"lay_{tag}_y"], out[f"lay_{tag}_dy"], out[f"lay_{tag}_dx"] = x.detach().numpy(), y.detach().numpy(), dy.numpy(), x.grad.numpy()
        for name, prm in params.items():
            out[f"lay_{tag}_{name}"] = prm.detach().numpy()
            out[f"lay_{tag}_d{name}"] = prm.grad.numpy()

    e, d = net.encoder, net.decoder
    one("s1", e.stage1, (2, C, H, W), {"w": e.stage1[0].weight, "b": e.stage1[0].bias})
    one("s2", e.stage2, (1, 64, H, W), {"w": e.stage2[0].weight, "b": e.stage2[0].bias})
    one("s3", e.stage3, (2, 96, H // 2, W // 2), {"w": e.stage3[0].weight, "b": e.stage3[0].bias})
    one("dc3", d.stage3, (1, 96, H // 4, W // 4), {"w": d.stage3[0].weight, "b": d.stage3[0].bias})
    one("dc2", d.stage2, (2, 96, H // 2, W // 2), {"w": d.stage2[0].weight, "b": d.stage2[0].bias})
    one("st1", d.stage1, (1, 64, H, W), {"w": d.stage1[0].weight, "b": d.stage1[0].bias})
    # head: input (S=1, B=1, 16, H, W) as the decoder hands it over; the loss only sees channel 0 (masked reg)
    hd = net.head
    for prm in hd.parameters():
        prm.grad = None
    f = torch.from_numpy(rs.normal(0, 1, (1, 1, 16, H, W)).astype(np.float32)).requires_grad_(True)
    o = hd(f)
    dreg = torch.from_numpy(rs.normal(0, 1, o[:, :, 0].shape).astype(np.float32))
    (o[:, :, 0] * dreg).sum().backward()
    out["lay_head_f"], out["lay_head_out"], out["lay_head_dreg"], out["lay_head_df"] = f.detach().numpy(), o.detach().numpy(), dreg.numpy(), f.grad.numpy()
    for name, prm in hd.named_parameters():
        if "_wrapper" in name:
            continue
        out[f"lay_head_p_{name}"] = prm.detach().numpy()
        out[f"lay_head_g_{name}"] = prm.grad.numpy() if prm.grad is not None else np.zeros_like(prm.detach().numpy())


def gen_training_loop(out, H=16, W=16, nums=3, seq_num=2, windows=3, wseed=21, eseed=8, lr=1e-3, grad_clip=1.0):
    """The reference SWP loop in fast mode (main.py:700-768: per window zero_grad, seq_num steps from the previous window's
    detached states, loss, backward, clip_grad_norm_, Adam step) on one seeded event: per-window losses and gradient norms,
    a (sum, sum of squares) fingerprint of every parameter after every window, and the final value of the small tensors."""
    C = 2 * nums + 3
    net, sd = mg.ref_net(H, W, C, wseed)
    net.train()
    T = seq_num * windows
    ev = uw.make_event(T, H, W, 60.0, seed=eseed)
    tev = mg.event_to_torch(ev)
    rs = np.random.RandomState(eseed + 1)
    label = (rs.uniform(0, 1, (1, T, H, W)) ** 3).astype(np.float32)
    label[label < 0.1] = 0.0
    named = {}
    for key in sd:
        named[key] = next(v for k, v in net.state_dict(keep_vars=True).items()
                          if k.replace("_wrapper.module.", ".").replace(".conv1_module.", ".conv1.").replace(".conv2_module.", ".conv2.") == key)
    params = list({id(p): p for p in named.values()}.values())
    opt = torch.optim.Adam(params, lr=lr)
    lossf = FocalBCE_and_WMSE(gamma=2, alpha=0.25)
    states = None
    out.update({"loop_H": H, "loop_W": W, "loop_nums": nums, "loop_seq_num": seq_num, "loop_windows": windows, "loop_weights_seed": wseed,
                "loop_event_seed": eseed, "loop_lr": lr, "loop_grad_clip": grad_clip, "loop_label": label, "loop_rain_max": 60.0,
                "loop_cumsum_max": 250.0})
    for wdx in range(windows):
        ind = wdx * seq_num
        opt.zero_grad()
        st = states if states is not None else initialize_states(torch.device("cpu"), input_height=H, input_width=W, net_cfg=mg.CFG)
        pred = None
        for t in range(ind, ind + seq_num):
            x = preprocess_inputs(t, tev, torch.device("cpu"), nums=nums, rain_max=60.0, cumsum_rain_max=250.0)
            res = net(x, *st)
            o, st = res[0], res[1:]
            cls = torch.where(o >= 0, 1, 0)
            pred = {"reg": o, "cls": cls} if pred is None else {"reg": torch.cat((pred["reg"], o), 1), "cls": torch.cat((pred["cls"], cls), 1)}
        states = [s.detach() for s in st]
        losses = lossf(pred, torch.from_numpy(label[:, ind:ind + seq_num]), epoch=0)
        losses["loss"].backward()
        norm = torch.nn.utils.clip_grad_norm_(params, grad_clip)
        opt.step()
        out[f"loop_w{wdx}_loss"] = np.float64(losses["loss"].item())
        out[f"loop_w{wdx}_gradnorm"] = np.float64(norm.item())
        out[f"loop_w{wdx}_fingerprint"] = np.array([[float(named[k].detach().double().sum()), float((named[k].detach().double() ** 2).sum())]
                                                    for k in sd])
    for k in sd:
        if named[k].numel() <= 4096:
            out[f"loop_final_{k}"] = named[k].detach().numpy()
    for i, s_ in enumerate(states):
        out[f"loop_state{i}"] = s_.numpy()


def gen_ddp_loop(out, H=16, W=16, nums=3, seq_num=2, windows=2, wseed=21, lr=1e-3, grad_clip=1.0, world=2):
    """DistributedDataParallel (main.py:384-387) emulated in one process: `world` ranks hold the same parameters, every rank runs
    the reference window (main.py:700-768) on ITS OWN event / labels, the gradients are averaged over the ranks (what DDP's
    all-reduce leaves in .grad), then clip_grad_norm_ and one Adam step on the shared parameters.  Per window: the per-rank losses
    and the averaged gradient of every parameter; at the end the parameters.  Events / labels as tests/test_hip_train.py
    _ddp_worker makes them (event seed 100 + rank, labels of the training-loop golden scaled by 1 + rank / 2)."""
    C = 2 * nums + 3
    net, sd = mg.ref_net(H, W, C, wseed)
    net.train()
    T = seq_num * windows
    rs = np.random.RandomState(8 + 1)                      # the label recipe of gen_training_loop (event seed 8)
    label = (rs.uniform(0, 1, (1, seq_num * 3, H, W)) ** 3).astype(np.float32)
    label[label < 0.1] = 0.0
    label = label[:, :T]
    named = {}
    for key in sd:
        named[key] = next(v for k, v in net.state_dict(keep_vars=True).items()
                          if k.replace("_wrapper.module.", ".").replace(".conv1_module.", ".conv1.").replace(".conv2_module.", ".conv2.") == key)
    params = list({id(p): p for p in named.values()}.values())
    opt = torch.optim.Adam(params, lr=lr)
    lossf = FocalBCE_and_WMSE(gamma=2, alpha=0.25)
    tevs = [mg.event_to_torch(uw.make_event(T, H, W, 60.0, seed=100 + r)) for r in range(world)]
    labels = [torch.from_numpy(label * (1.0 + 0.5 * r)) for r in range(world)]
    states = [None] * world
    out.update({"ddp_H": H, "ddp_W": W, "ddp_nums": nums, "ddp_seq_num": seq_num, "ddp_windows": windows, "ddp_weights_seed": wseed,
                "ddp_lr": lr, "ddp_grad_clip": grad_clip, "ddp_world": world, "ddp_label": label})
    for wdx in range(windows):
        ind = wdx * seq_num
        mean_grad = {k: torch.zeros_like(named[k]) for k in sd}
        for r in range(world):
            opt.zero_grad()
            st = states[r] if states[r] is not None else initialize_states(torch.device("cpu"), input_height=H, input_width=W, net_cfg=mg.CFG)
            pred = None
            for t in range(ind, ind + seq_num):
                x = preprocess_inputs(t, tevs[r], torch.device("cpu"), nums=nums, rain_max=60.0, cumsum_rain_max=250.0)
                res = net(x, *st)
                o, st = res[0], res[1:]
                cls = torch.where(o >= 0, 1, 0)
                pred = {"reg": o, "cls": cls} if pred is None else {"reg": torch.cat((pred["reg"], o), 1), "cls": torch.cat((pred["cls"], cls), 1)}
            states[r] = [s.detach() for s in st]
            losses = lossf(pred, labels[r][:, ind:ind + seq_num], epoch=0)
            losses["loss"].backward()
            out[f"ddp_w{wdx}_rank{r}_loss"] = np.float64(losses["loss"].item())
            for k in sd:
                if named[k].grad is not None:
                    mean_grad[k] += named[k].grad / world
        for k in sd:                                            # what every rank holds after DDP's all-reduce
            named[k].grad = mean_grad[k].clone()
            out[f"ddp_w{wdx}_grad_{k}"] = mean_grad[k].numpy().copy()
        norm = torch.nn.utils.clip_grad_norm_(params, grad_clip)
        out[f"ddp_w{wdx}_gradnorm"] = np.float64(norm.item())
        opt.step()
    for k in sd:
        out[f"ddp_final_{k}"] = named[k].detach().numpy().copy()


if __name__ == "__main__":
    sys.modules.setdefault("wandb", types.ModuleType("wandb"))
    if len(sys.argv) > 1 and sys.argv[1] == "ddp":          # only the DDP golden (the other files stay byte-identical)
        ddp = {}
        gen_ddp_loop(ddp)
        np.savez_compressed(os.path.join(HERE, "train_ddp_16x16.npz"), **ddp)
        print("wrote train_ddp_16x16.npz", os.path.getsize(os.path.join(HERE, "train_ddp_16x16.npz")) // 1024, "KiB")
        sys.exit(0)
    out = {}
    gen_loss(out)
    gen_window(out)
    cells = {}
    gen_cell_backward(cells)
    np.savez_compressed(os.path.join(HERE, "train_cell_backward.npz"), **cells)
    layers = {}
    gen_layers_backward(layers)
    np.savez_compressed(os.path.join(HERE, "train_layers_backward.npz"), **layers)
    loop = {}
    gen_training_loop(loop)
    np.savez_compressed(os.path.join(HERE, "train_loop_16x16.npz"), **loop)
    ddp = {}
    gen_ddp_loop(ddp)
    np.savez_compressed(os.path.join(HERE, "train_ddp_16x16.npz"), **ddp)
    path = os.path.join(HERE, "train_window_16x16.npz")
    np.savez_compressed(path, **out)
    ng = sum(1 for k in out if k.startswith("win_grad_"))
    nn_ = sum(1 for k in out if k.startswith("win_hasgrad_"))
    print("wrote", path, os.path.getsize(path) // 1024, "KiB;", ng, "of", nn_, "parameters receive a gradient; loss", out["win_loss"])
