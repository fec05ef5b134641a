"""One SWP training window on the HIP path (main.py:598-768 `process_window` + loss + backward):
`seq_num` timesteps forward with every activation the backward needs kept, the loss on the concatenated outputs, then
back-propagation through the steps and through the six recurrent states, giving the gradient of all 79 parameter tensors.

`Trainer` adds global-norm clipping + Adam on flat buffers, the SWP loop (fast mode and pre-warming) and the DDP gradient mean
(RCCL, overlapped with the backward pass)."""
import os

import torch

from . import ops, train_ops
from .ops import tuning_env as _tuning_env
from .dataset import event_to_device


def _cell_params(cell):
    c1, g1, c2, g2 = cell.conv1[0], cell.conv1[1], cell.conv2[0], cell.conv2[1]
    return c1, g1, c2, g2


class WindowGradients:
    """gradients of one window:  wg = WindowGradients(net, H, W, nums, rain_max, cumsum_max);  out = wg.run(event, targets, t0,
    steps, states)  ->  dict(loss=5 floats, grads={reference parameter name: tensor}, states=[6 tensors], reg=(B,steps,H,W))."""

    def __init__(self, net, H, W, nums, rain_max, cumsum_max, cls_thred_train=0.0):
        self.net, self.H, self.W, self.nums = net, H, W, int(nums)
        self.rain_max, self.cumsum_max = float(rain_max), float(cumsum_max)
        self.cls_thred_train = float(cls_thred_train)          # classify_outputs threshold of the loop (main.py:598: 0)
        self.device = next(net.parameters()).device
        self._bwd_packed = {}      # per cell: packed weights of the input-gradient GEMMs, valid until the parameters change
        # scratch owned by this object: ("fwd", step, layer) keeps the forward scratch of layer `layer` of window step `step`
        # until its backward ran, "bwd" is the backward kernels' scratch
        self.arena = ops.Arena(self.device)
        self._fwd_streams = None
        self.forward_chains = _tuning_env("URNN_TUNE_TRAIN_CHAINS", "1") != "0"      # _forward_window instead of step by step
        self.backward_chains = _tuning_env("URNN_TUNE_TRAIN_BWD_CHAINS", "1") != "0"  # _backward_window likewise
        self._arena_d, self._arena_e = ops.Arena(self.device), ops.Arena(self.device)    # scratch of the decoder / encoder backward chains

    # -- forward of one timestep, keeping what the backward reads ---------------------------------------------------
    # The forward of a timestep in the three parts that only meet through the tensors they hand on (model.py:65-121): encoder(s + 1)
    # reads the encoder states of step s, decoder(s) the encoder outputs of step s and its own states, head(s) the decoder's last
    # feature map.  _forward_step runs them in order; _forward_window runs them as three kernel chains.
    def _cell_fwd(self, S, step, k, mod, x, e, h):
        S["ws"][k] = self.arena.get(("fwd", step, k), ops.gru_cell_workspace_bytes(h.shape[0], h.shape[1], h.shape[2], h.shape[3]))
        return mod.step(x, e, h, ws=S["ws"][k])

    def _forward_enc(self, ev, t, enc_states, step, t_dev=None):
        enc = self.net.encoder
        e1, e2, e3 = enc_states
        S = {"prev": list(enc_states) + [None] * 3, "ws": {}}
        S["x_in"] = ops.preprocess(ev["rain"], ev["cumsum"], ev["dem"], ev["imperv"], ev["manhole"], ev["dem_min"], ev["dem_max"], int(t),
                                   self.nums, self.rain_max, self.cumsum_max, t_dev=t_dev)
        S["a1"] = enc.stage1(S["x_in"])
        S["e1"] = self._cell_fwd(S, step, 0, enc.rnn1, S["a1"], None, e1)
        S["a2"] = enc.stage2(S["e1"])
        S["e2"] = self._cell_fwd(S, step, 1, enc.rnn2, S["a2"], None, e2)
        S["a3"] = enc.stage3(S["e2"])
        S["e3"] = self._cell_fwd(S, step, 2, enc.rnn3, S["a3"], None, e3)
        return S

    def _forward_dec(self, S, dec_states, step):
        dec = self.net.decoder
        d1, d2, d3 = dec_states
        S["prev"][3:] = list(dec_states)
        S["d1"] = self._cell_fwd(S, step, 3, dec.rnn3, None, S["e3"], d1)
        S["u3"] = dec.stage3(S["d1"])
        S["d2"] = self._cell_fwd(S, step, 4, dec.rnn2, S["u3"], S["e2"], d2)
        S["u2"] = dec.stage2(S["d2"])
        S["d3"] = self._cell_fwd(S, step, 5, dec.rnn1, S["u2"], S["e1"], d3)
        S["feat"] = dec.stage1(S["d3"])

    def _forward_head(self, S, step):
        f = S["feat"]
        S["ws"][6] = self.arena.get(("fwd", step, 6), ops.head_workspace_bytes(f.shape[0], f.shape[1], f.shape[2], f.shape[3]))
        S["masked"], S["cls"], S["raw"] = self.net.head.run(f, want_raw=True, ws=S["ws"][6])

    def _forward_step(self, ev, t, states, step, t_dev=None):
        S = self._forward_enc(ev, t, states[:3], step, t_dev)
        self._forward_dec(S, states[3:], step)
        self._forward_head(S, step)
        return S, [S["e1"], S["e2"], S["e3"], S["d1"], S["d2"], S["d3"]]

    def _forward_window(self, ev, t0, steps, states, t_devs=None):
        """The window's forward as a software pipeline, the rollout engine's schedule (rollout.py): iteration i runs
        encoder(i + 1) || decoder(i) || head(i - 1) on three streams forked from and joined into the current one.  Same kernels on the
        same operands as _forward_step step by step, so the same bits; what every part allocates is kept in `saved` until the window's
        backward has run, and the streams are joined before anything on the current stream reads it."""
        cur = torch.cuda.current_stream(self.device)
        if self._fwd_streams is None:
            self._fwd_streams = [torch.cuda.Stream(device=self.device) for _ in range(3)]
        s1, s2, s3 = self._fwd_streams
        td = (lambda s: None) if t_devs is None else (lambda s: t_devs[s])
        saved = [self._forward_enc(ev, t0, states[:3], 0, td(0))]
        dstates = list(states[3:])
        for i in range(steps):
            for s in (s1, s2, s3):
                s.wait_stream(cur)
            with torch.cuda.stream(s2):
                self._forward_dec(saved[i], dstates, i)
            if i + 1 < steps:
                with torch.cuda.stream(s1):
                    S = saved[i]
                    saved.append(self._forward_enc(ev, t0 + i + 1, [S["e1"], S["e2"], S["e3"]], i + 1, td(i + 1)))
            if i >= 1:
                with torch.cuda.stream(s3):
                    self._forward_head(saved[i - 1], i - 1)
            for s in (s1, s2, s3):
                cur.wait_stream(s)
            dstates = [saved[i]["d1"], saved[i]["d2"], saved[i]["d3"]]
        self._forward_head(saved[steps - 1], steps - 1)
        S = saved[steps - 1]
        return saved, [S["e1"], S["e2"], S["e3"], S["d1"], S["d2"], S["d3"]]

    # -- backward of one timestep -----------------------------------------------------------------------------------------
    # The backward of a timestep in the three parts that only meet through the gradients they hand on: head (d loss / d feature map),
    # decoder (gradients of the decoder's parameters and states + what flows into the encoder states through the skip connections),
    # encoder.  _backward_step runs them in order on one scratch arena; _backward_window as three kernel chains, an arena each.
    def _bwd_ops(self, S, G, acc, arena):
        def conv_bwd(name, mod, x, dy):
            L = mod.layer
            key = f"{name}.{mod._pname}"
            dx, dw, db = train_ops.stage_conv_backward(x, L.weight.detach(), L.bias.detach(), dy, mod.pool, dweight=G.get(key + ".weight"),
                                                       dbias=G.get(key + ".bias"), accumulate=acc and (key + ".weight") in G,
                                                       scratch=arena, packed=self._bwd_packed.setdefault(key, []))
            G[key + ".weight"], G[key + ".bias"] = dw, db
            return dx

        def deconv_bwd(name, mod, x, out, dy):
            L = mod.layer
            key = f"{name}.{mod._pname}"
            dx, dw, db = train_ops.deconv2x2_backward(x, L.weight.detach(), out, dy, dweight=G.get(key + ".weight"),
                                                      dbias=G.get(key + ".bias"), accumulate=acc and (key + ".weight") in G,
                                                      scratch=arena, packed=self._bwd_packed.setdefault(key, []))
            G[key + ".weight"], G[key + ".bias"] = dw, db
            return dx

        def cell_bwd(k, name, mod, x, e, h, *terms):
            """terms: tensors / tuples of tensors / None -- the pieces of dL/dh'.  Returns dx, de, (dh, dh2)."""
            flat = []
            for t in terms:
                if t is None:
                    continue
                flat.extend(v for v in (t if isinstance(t, (tuple, list)) else (t,)) if v is not None)
            flat = [v.contiguous() for v in flat]
            while len(flat) > 4:                    # (never with this network: at most layer above + skip + two from the next step)
                flat = [flat[0] + flat[1]] + flat[2:]
            flat += [None] * (4 - len(flat))
            c1, g1, c2, g2 = _cell_params(mod)
            names = {"dW1": "conv1.0.weight", "db1": "conv1.0.bias", "dg1": "conv1.1.weight", "dbe1": "conv1.1.bias",
                     "dW2": "conv2.0.weight", "db2": "conv2.0.bias", "dg2": "conv2.1.weight", "dbe2": "conv2.1.bias"}
            have = f"{name}.conv1.0.weight" in G
            prev = {k_: (G[f"{name}.{v}"].reshape(-1) if k_ in ("db1", "dg1", "dbe1", "db2", "dg2", "dbe2") else
                         G[f"{name}.{v}"].reshape(G[f"{name}.{v}"].shape[0], -1)) for k_, v in names.items()} if have else None
            g = train_ops.gru_cell_backward(x, e, h, c1.weight.detach(), c2.weight.detach(), g1.weight.detach(), g2.weight.detach(),
                                            flat[0], mod.input_channels, S["ws"][k], grads=prev, accumulate=acc and have,
                                            packed=self._bwd_packed.setdefault(name, []), dh_out2=flat[1], dh_out3=flat[2], dh_out4=flat[3],
                                            split_dh=True, scratch=arena)
            for k_, v in names.items():
                ref = dict(mod.named_parameters())[v]
                G[f"{name}.{v}"] = g[k_].reshape(ref.shape)
            return g.get("dx"), g.get("de"), (g["dh"], g["dh2"])
        return conv_bwd, deconv_bwd, cell_bwd

    def _bwd_head(self, S, dout, G, acc, arena, after_head=None):
        """-> d loss / d feature map of this step; the head's parameter gradients accumulate in G["_head"]."""
        head = self.net.head
        fp = head.flat_params()
        hg_prev = G.get("_head") if acc else None
        hg = train_ops.head_backward(S["feat"], fp["conv_w"], fp["ln_w"], fp["ln_b"], head.reg_preds.conv.weight.detach().reshape(-1),
                                     S["raw"], S["cls"], dout.contiguous(), head.cls_thred, S["ws"][6], grads=hg_prev,
                                     accumulate=hg_prev is not None, scratch=arena)
        G["_head"] = hg
        if after_head is not None:      # the head's gradients of this window are final here (when this is the first timestep)
            after_head(hg)
        return hg["dfeat"]

    def _bwd_dec(self, S, dfeat, dD, G, acc, arena):
        """dD: the terms arriving at this step's three NEW decoder states from the following step.  -> (the three terms the skip
        connections send into this step's new encoder states, the terms for the decoder states this step STARTED from)."""
        dec = self.net.decoder
        conv_bwd, deconv_bwd, cell_bwd = self._bwd_ops(S, G, acc, arena)
        d1p, d2p, d3p = S["prev"][3:]
        dD1, dD2, dD3 = dD
        du2, dE1_dec, dD3n = cell_bwd(5, "decoder.rnn1", dec.rnn1, S["u2"], S["e1"], d3p, conv_bwd("decoder.stage1", dec.stage1, S["d3"], dfeat), dD3)
        du3, dE2_dec, dD2n = cell_bwd(4, "decoder.rnn2", dec.rnn2, S["u3"], S["e2"], d2p, deconv_bwd("decoder.stage2", dec.stage2, S["d2"], S["u2"], du2), dD2)
        _, dE3_dec, dD1n = cell_bwd(3, "decoder.rnn3", dec.rnn3, None, S["e3"], d1p, deconv_bwd("decoder.stage3", dec.stage3, S["d1"], S["u3"], du3), dD1)
        return (dE1_dec, dE2_dec, dE3_dec), (dD1n, dD2n, dD3n)

    def _bwd_enc(self, S, dE_dec, dE, G, acc, arena):
        """dE_dec: from this step's decoder (skip connections); dE: from the following step's encoder.  -> the terms for the encoder
        states this step STARTED from."""
        enc = self.net.encoder
        conv_bwd, _, cell_bwd = self._bwd_ops(S, G, acc, arena)
        e1p, e2p, e3p = S["prev"][:3]
        dE1_dec, dE2_dec, dE3_dec = dE_dec
        dE1, dE2, dE3 = dE
        da3, _, dE3n = cell_bwd(2, "encoder.rnn3", enc.rnn3, S["a3"], None, e3p, dE3_dec, dE3)
        da2, _, dE2n = cell_bwd(1, "encoder.rnn2", enc.rnn2, S["a2"], None, e2p, conv_bwd("encoder.stage3", enc.stage3, S["e2"], da3), dE2_dec, dE2)
        da1, _, dE1n = cell_bwd(0, "encoder.rnn1", enc.rnn1, S["a1"], None, e1p, conv_bwd("encoder.stage2", enc.stage2, S["e1"], da2), dE1_dec, dE1)
        conv_bwd("encoder.stage1", enc.stage1, S["x_in"], da1)
        return (dE1n, dE2n, dE3n)

    def _backward_step(self, S, step, dout, dstate, G, acc, after_head=None):
        """dout: d loss / d masked output of this step (B,H,W); dstate: gradients arriving at this step's six NEW states from
        the following step (or None); returns the gradients w.r.t. the six states this step STARTED from."""
        # a state's gradient arrives as a TUPLE of terms (None entries dropped): the layer above in this timestep, the skip
        # connection's reader, and the same cell in the next timestep -- which itself hands over two (blend / reset-gate part, and the
        # gates' GEMM output).  The cell backward sums up to four terms on the fly; nothing is added by a pass of its own.
        dE1, dE2, dE3, dD1, dD2, dD3 = dstate if dstate is not None else [()] * 6
        # (the same scratch arenas as the pipelined form: whichever ran in the eager warm-up has sized them for the other's capture)
        dfeat = self._bwd_head(S, dout, G, acc, self.arena, after_head)
        dE_dec, dDn = self._bwd_dec(S, dfeat, (dD1, dD2, dD3), G, acc, self._arena_d)
        dEn = self._bwd_enc(S, dE_dec, (dE1, dE2, dE3), G, acc, self._arena_e)
        return list(dEn) + list(dDn)

    def _backward_window(self, saved, dreg, steps, G, on_head_final=None):
        """The window's backward as a software pipeline over the timesteps, last to first: iteration k runs head(S-1-k) ||
        decoder(S-k) || encoder(S+1-k) on three streams forked from and joined into the current one (the forward's schedule,
        mirrored).  Each chain has its own scratch arena and touches its own parameters' gradients; what a chain hands to another
        (dfeat, the skip connections' terms) is kept alive until the window's end, so no block is reused while a stream still reads it.
        on_head_final(head gradients): called on the current stream once the iteration with the FIRST timestep's head backward is
        joined -- the head's gradients are final there, with decoder(0), encoder(1) and encoder(0) still to run."""
        cur = torch.cuda.current_stream(self.device)
        if self._fwd_streams is None:
            self._fwd_streams = [torch.cuda.Stream(device=self.device) for _ in range(3)]
        sE, sD, sH = self._fwd_streams
        dfeat, dE_dec = {}, {}
        dD, dE = ((), (), ()), ((), (), ())
        for k in range(steps + 2):
            sh, sd, se = steps - 1 - k, steps - k, steps + 1 - k
            for s in (sE, sD, sH):
                s.wait_stream(cur)
            # (the enqueue order of the three parts makes no difference here: 4.09-4.12 ms for all of them)
            if 0 <= sd < steps:
                with torch.cuda.stream(sD):
                    dE_dec[sd], dD = self._bwd_dec(saved[sd], dfeat[sd], dD, G, sd != steps - 1, self._arena_d)
            if 0 <= se < steps:
                with torch.cuda.stream(sE):
                    dE = self._bwd_enc(saved[se], dE_dec[se], dE, G, se != steps - 1, self._arena_e)
            if 0 <= sh < steps:
                with torch.cuda.stream(sH):
                    dfeat[sh] = self._bwd_head(saved[sh], dreg[:, sh], G, sh != steps - 1, self.arena)
            for s in (sE, sD, sH):
                cur.wait_stream(s)
            if sh == 0 and on_head_final is not None:
                on_head_final(self.head_gradients(G["_head"]))
        return list(dE) + list(dD)

    def arena_generation(self):
        return self.arena.generation + self._arena_d.generation + self._arena_e.generation

    def head_gradients(self, hg):
        """{reference parameter name: tensor} views of the head backward's stacked buffers."""
        head, grads = self.net.head, {}
        for i, blk in enumerate(["stems", "cls_convs.0", "cls_convs.1", "reg_convs.0", "reg_convs.1"]):
            grads[f"head.{blk}.conv.weight"] = hg["dconv_w"][i].reshape(head.channels, head.channels, 1, 1)
            grads[f"head.{blk}.ln.weight"] = hg["dln_w"][i]
            grads[f"head.{blk}.ln.bias"] = hg["dln_b"][i]
        grads["head.reg_preds.conv.weight"] = hg["dreg_w"].reshape(1, -1, 1, 1)
        grads["head.reg_preds.conv.bias"] = hg["dreg_b"]
        # the classification branch gets no gradient (cut off by the wet/dry comparison, main.py:500): explicit zeros, so that the
        # single-GPU and the DDP path (which all-reduces these slots with the head's tail) write the same regions every window
        grads["head.cls_preds.conv.weight"] = torch.zeros_like(head.cls_preds.conv.weight)
        grads["head.cls_preds.conv.bias"] = torch.zeros_like(head.cls_preds.conv.bias)
        return grads

    def run(self, event, targets, t0, steps, states=None, t_devs=None, grad_buffers=None, on_head_final=None):
        """event: reference-layout event dict (or already on the device); targets (B,steps,H,W) normalised depths of frames
        t0 .. t0+steps-1; states: six (B,C,h,w) tensors or None (zeros); t_devs: optional list of int32 device scalars holding
        the frame index of every step (hipGraph replay); grad_buffers: optional {parameter name: tensor} the encoder / decoder
        gradients are written into directly (views of a flat gradient buffer); on_head_final(head gradient dict): called as soon as
        the head's gradients of the window are complete -- right after the head backward of the FIRST timestep, before that
        timestep's decoder / encoder backward (DDP starts their all-reduce there).  See the class docstring for the result."""
        ev = event if "rain" in event else event_to_device(event, self.device)
        B = ev["B"]
        if states is None:
            from .general import initialize_states
            states = [s.to(self.device).repeat(B, 1, 1, 1) for s in initialize_states(self.device, self.H, self.W)]
        targets = torch.as_tensor(targets, dtype=torch.float32, device=self.device).contiguous()
        for v in self._bwd_packed.values():      # the parameters may have changed since the last window: re-pack once per window
            v.clear()
        if self.forward_chains and self.device.type == "cuda":
            saved, states = self._forward_window(ev, t0, steps, states, t_devs)
        else:
            saved = []
            for s in range(steps):
                S, states = self._forward_step(ev, t0 + s, states, s, None if t_devs is None else t_devs[s])
                saved.append(S)
        reg = torch.stack([S["masked"] for S in saved], dim=1).contiguous()            # (B,steps,H,W) as main.py concatenates
        comps, dreg = train_ops.loss(reg, targets, cls_thred=self.cls_thred_train, scratch=self.arena)
        G, dstate = ({k: v for k, v in grad_buffers.items() if not k.startswith("head.")} if grad_buffers else {}), None
        if self.backward_chains and self.device.type == "cuda":
            dstate = self._backward_window(saved, dreg, steps, G, on_head_final)
        else:
            for s in reversed(range(steps)):
                hook = (lambda hg: on_head_final(self.head_gradients(hg))) if (on_head_final is not None and s == 0) else None
                dstate = self._backward_step(saved[s], s, dreg[:, s], dstate, G, acc=(s != steps - 1), after_head=hook)
        grads = {k: v for k, v in G.items() if not k.startswith("_")}
        grads.update(self.head_gradients(G["_head"]))
        # state_grads: per state the TWO terms of dL/d(initial state) (their sum is the gradient; left unsummed: nobody in the loop reads it)
        return {"loss": comps, "grads": grads, "states": [s.detach() for s in states], "reg": reg, "state_grads": dstate}


def window_starts(loc, seq_num, window_size):
    """Start indices of the SWP windows of one sample (split_iter_index, main.py:162-178): loc, loc + seq_num, ...; when
    window_size is not a multiple of seq_num the last window is shifted back to end exactly at loc + window_size."""
    loc, seq_num, window_size = int(loc), int(seq_num), int(window_size)
    idx = list(range(loc, loc + window_size, seq_num))
    if window_size % seq_num > 0:
        idx[-1] = loc + window_size - seq_num
    return idx


# ---- SWP window planning (get_window, main.py:415-443 with main.py:106-159) -------------------------------------------------------
def correction_seq_num(seq_num, window_size, full_window_size=False):
    """main.py:106-118."""
    return window_size if full_window_size else min(seq_num, window_size)


def correction_window_size(window_size, rain_len, event_len, all_seq_train=False, train_event=False):
    """main.py:121-136."""
    sample_length = event_len if train_event else rain_len
    return sample_length if all_seq_train else min(window_size, sample_length)


def get_start_loc(rain_len, window_size, event_len, train_event=False):
    """main.py:139-159: a random start when the sample is longer than the window (numpy's global generator, as there)."""
    import numpy as np
    sample_length = event_len if train_event else rain_len
    loc = 0
    if sample_length - window_size > 0:
        loc = int(np.random.randint(0, sample_length - window_size, size=1, dtype=int)[0])
    return loc


def plan_windows(rain_len, event_len, seq_num, window_size, all_seq_train=False, train_event=True, full_window_size=False,
                 wind_random=True):
    """``get_window`` (main.py:415-443).  Returns (loc, seq_num, window_size, window start indices in processing order);
    like the reference, the corrected seq_num / window_size replace the configured ones for the rest of the run.  Feed the
    starts to ``Trainer.train_event(..., starts=...)``."""
    import random
    window_size = correction_window_size(window_size, rain_len, event_len, all_seq_train, train_event)
    loc = get_start_loc(rain_len, window_size, event_len, train_event)
    seq_num = correction_seq_num(seq_num, window_size, full_window_size)
    starts = window_starts(loc, seq_num, window_size)
    if wind_random:
        random.shuffle(starts)
    return loc, seq_num, window_size, starts


class Trainer:
    """SWP training on the HIP path: the loop of ``model_forward`` (main.py:700-768) in fast mode -- per window: zero the
    gradients, ``seq_num`` timesteps from the previous window's (detached) states, loss, backward, global-norm clipping, Adam.

    The 79 parameter tensors are re-homed as views of ONE flat float32 buffer (same for the gradients and Adam's moments), so
    the optimizer is a single kernel over 40 M floats at 500x500 and the DDP exchange a single all-reduce of that buffer
    (``torch.distributed``, RCCL over xGMI with the nccl backend) -- mean over ranks, as DistributedDataParallel does
    (main.py:384-387)."""

    def __init__(self, net, H, W, nums, rain_max, cumsum_max, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, grad_clip=0.0,
                 cls_thred_train=0.0, process_group=None, distributed=False, use_graph=False, matrix_mode="fp32"):
        self.net = net
        # "fp32": the reference's arithmetic (default).  "bf16": BASELINE configs[3] -- every GEMM (forward, input gradients and
        # weight gradients) takes its operands rounded to bf16 and accumulates in fp32 (ops.matrix_mode); master weights, the
        # stored gradients, Adam, norms and the loss stay fp32 (tests/test_hip_train_fullsize.py::test_bf16_window_full_size)
        if matrix_mode not in ops.MATRIX_MODES:
            raise ValueError(f"matrix_mode must be one of {sorted(ops.MATRIX_MODES)}")
        self.matrix_mode = matrix_mode
        # capture one window as hipGraphs: single GPU -- one graph (forward, loss, backward, clip + Adam); DDP -- three graphs with
        # the two halves of the gradient mean enqueued between them (forward + backward up to the point where the head's gradients
        # are final | the rest of the backward | clip + Adam), main.py:384-387,750-762
        self.use_graph = bool(use_graph)
        self._graphs = {}           # captured windows by key (one per catchment / shape / lr): see _train_window_graph
        self.wg = WindowGradients(net, H, W, nums, rain_max, cumsum_max, cls_thred_train)
        self.lr, self.betas, self.eps, self.grad_clip = float(lr), tuple(betas), float(eps), float(grad_clip)
        self.distributed, self.pg = bool(distributed), process_group
        self.names = [n for n, _ in net.named_parameters()]
        params = dict(net.named_parameters())
        total = sum(p.numel() for p in params.values())
        dev = next(net.parameters()).device
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
        self.gflat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self.views, off = {}, 0
        for n in self.names:
            p = params[n]
            k = p.numel()
            self.flat[off:off + k].copy_(p.detach().reshape(-1))
            p.data = self.flat[off:off + k].view(p.shape)                 # parameters now live in the flat buffer
            self.views[n] = (off, k, tuple(p.shape))
            off += k
        self.grad_views = {n: self.gflat[off:off + k].view(shape) for n, (off, k, shape) in self.views.items()}
        # net.named_parameters() lists encoder, decoder, head in this order: the head is the contiguous tail of the flat buffers
        self.head_offset = min(off for n, (off, k, shape) in self.views.items() if n.startswith("head."))
        if any(off >= self.head_offset and not n.startswith("head.") for n, (off, k, shape) in self.views.items()):
            raise RuntimeError("Trainer: expected the head's parameters to be the tail of net.named_parameters()")
        self.step_count = 0
        self.last = None

    def _invalidate_packed(self):
        """The kernels read packed copies of the weights; an in-place optimizer step does not bump torch's version counters."""
        for mod in self.net.modules():
            cache = getattr(mod, "_cache", None)
            if cache is not None and hasattr(cache, "clear"):
                cache.clear()
        self.net.head._stamp = None
        # engines that captured a rollout graph of this net (inference.Inference, RolloutEngine) compare this counter
        self.net._urnn_generation = getattr(self.net, "_urnn_generation", 0) + 1


    def refresh_weight_ranges(self):
        """The per-layer choice between the f16-piece k-loops and the exact fp32 matrix instruction (|weight| >= 64: include/urnn_hip.h
        "Operand range") for the windows that follow -- called once per event, outside any capture.  One reduction over the flat
        parameter buffer and one host read; only a network that has a large weight somewhere pays the per-layer check.  A captured
        window bakes the choice in, so a flag that flips drops the captured windows."""
        from .networks._packing import F16_WEIGHT_LIMIT
        caches = [(mod, mod._cache) for mod in self.net.modules() if hasattr(getattr(mod, "_cache", None), "owner_checks")]
        small = ops.max_abs(self.flat) < F16_WEIGHT_LIMIT
        flipped = False
        for mod, cache in caches:
            cache.owner_checks = True
            wide = False
            if not small:
                ws = [p.detach().contiguous() for p in mod.parameters() if p.dim() == 4]     # the conv / deconv weights below this module
                wide = bool(ws) and not (max(ops.max_abs(w) for w in ws) < F16_WEIGHT_LIMIT)
            flipped |= wide != cache.wide
            cache.wide = wide
        if flipped:
            self._graphs.clear()

    def release_weight_ranges(self):
        """Hand the weight-range check back to the layers (``PackedCache.get`` re-reads max |w| when a layer is repacked): the flags
        ``refresh_weight_ranges`` set are this trainer's promise for ONE event; anything else that runs this network afterwards --
        ``Inference`` / ``RolloutEngine``, an eager window, another ``load_state_dict`` -- must not inherit it."""
        for mod in self.net.modules():
            cache = getattr(mod, "_cache", None)
            if hasattr(cache, "owner_checks"):
                cache.owner_checks = False

    def set_lr(self, lr):
        """Learning rate of the following windows (the epoch loop calls this once per epoch; the captured window is re-captured)."""
        lr = float(lr)
        if lr != self.lr:
            # a captured window bakes its learning rate in: graphs of another rate can never be replayed again -- release their private
            # pools (a whole window's activations each) now instead of leaving them to the LRU eviction
            for key in [k for k in self._graphs if k[6] != lr]:
                del self._graphs[key]
        self.lr = lr

    # -- checkpoints: the reference's file content (earlystopping.py:40-44) ----------------------------------------------------
    def state_dict(self):
        """The network's state dict (reference key names), detached copies on the host."""
        return {k: v.detach().cpu().clone() for k, v in self.net.state_dict().items()}

    def load_state_dict(self, sd):
        """Copy a reference-layout state dict into the flat parameter buffer (missing / unexpected keys raise as in torch)."""
        own = self.net.state_dict()
        missing, unexpected = [k for k in own if k not in sd], [k for k in sd if k not in own]
        if missing or unexpected:
            raise KeyError(f"load_state_dict: missing {missing[:3]}{'...' if len(missing) > 3 else ''}, unexpected {unexpected[:3]}")
        with torch.no_grad():
            for k, v in own.items():
                v.copy_(torch.as_tensor(sd[k]).to(device=v.device, dtype=v.dtype).reshape(v.shape))
        self._invalidate_packed()
        self.refresh_weight_ranges()
        self.release_weight_ranges()

    def optimizer_state_dict(self):
        """torch.optim.Adam.state_dict() of the equivalent optimizer: parameters numbered in ``net.parameters()`` order."""
        state = {}
        if self.step_count > 0:
            for i, n in enumerate(self.names):
                off, k, shape = self.views[n]
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.m[off:off + k].view(shape).detach().cpu().clone(),
                            "exp_avg_sq": self.v[off:off + k].view(shape).detach().cpu().clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "params": list(range(len(self.names)))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, osd):
        group = osd["param_groups"][0]
        if len(group["params"]) != len(self.names):
            raise ValueError(f"optimizer state has {len(group['params'])} parameters, the network {len(self.names)}")
        self.lr, self.betas, self.eps = float(group["lr"]), tuple(group["betas"]), float(group["eps"])
        self.m.zero_()
        self.v.zero_()
        steps = 0
        for i, pid in enumerate(group["params"]):
            st = osd["state"].get(pid)
            if st is None:
                continue
            off, k, _ = self.views[self.names[i]]
            self.m[off:off + k].copy_(torch.as_tensor(st["exp_avg"]).reshape(-1))
            self.v[off:off + k].copy_(torch.as_tensor(st["exp_avg_sq"]).reshape(-1))
            steps = max(steps, int(float(st["step"])))
        self.step_count = steps

    def _scatter(self, grads):
        """Copy gradients that were not written in place (the head's stacked buffers) into the flat gradient buffer."""
        for n, g in grads.items():
            off, k, _ = self.views[n]
            if g.data_ptr() != self.grad_views[n].data_ptr():
                self.gflat[off:off + k].copy_(g.reshape(-1))

    def _window_body(self, ev, targets, t0, steps, states, t_devs=None, step_dev=None):
        reducer = None
        if self.distributed:
            # DDP (main.py:384-387): the head's gradients (the flat buffer's tail, 99 % of the bytes) are all-reduced over RCCL
            # while the first timestep's decoder / encoder backward still runs; the remainder after the backward
            from .distributed import OverlappedGradientMean
            reducer = OverlappedGradientMean(self.gflat, self.head_offset, group=self.pg)

            def head_final(head_grads):
                self._scatter(head_grads)
                reducer.start_tail()
        out = self.wg.run(ev, targets, t0, steps, states, t_devs=t_devs, grad_buffers=self.grad_views,
                          on_head_final=head_final if reducer is not None else None)
        if reducer is None:
            self._scatter(out["grads"])
        else:
            self._scatter({n: g for n, g in out["grads"].items() if not n.startswith("head.")})
            reducer.finish()
        clip = train_ops.adam_step(self.flat, self.gflat, self.m, self.v, max(self.step_count, 1), lr=self.lr, betas=self.betas,
                                   eps=self.eps, max_grad_norm=self.grad_clip, step_dev=step_dev, scratch=self.wg.arena)
        self._invalidate_packed()
        return out, clip

    def _window_body_ddp_graphs(self, sev, G, steps):
        """Capture the DDP window as three graphs on the current (side) stream: A ends where the head's gradients of the window are
        complete (right after the head backward of the first timestep), B holds that timestep's decoder / encoder backward and
        the hand-over of the remaining gradients, C the clipped Adam step.  Nothing is reduced during capture."""
        gA, gB, gC = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        pool = torch.cuda.graph_pool_handle()        # one memory pool: activations kept by A are read by B
        gA.capture_begin(pool=pool)

        def head_final(head_grads):
            self._scatter(head_grads)
            gA.capture_end()
            gB.capture_begin(pool=pool)
        out = self.wg.run(sev, G["tgt"], 0, steps, G["states"], t_devs=G["t_devs"], grad_buffers=self.grad_views, on_head_final=head_final)
        self._scatter({n: g for n, g in out["grads"].items() if not n.startswith("head.")})
        gB.capture_end()
        gC.capture_begin(pool=pool)
        clip = train_ops.adam_step(self.flat, self.gflat, self.m, self.v, 1, lr=self.lr, betas=self.betas, eps=self.eps,
                                   max_grad_norm=self.grad_clip, step_dev=G["step_dev"], scratch=self.wg.arena)
        gC.capture_end()
        return (gA, gB, gC), out, clip

    MAX_CACHED_WINDOWS = 4

    def _train_window_graph(self, ev, targets, t0, steps, states):
        """hipGraph path: static input buffers (targets, the six states, the per-step frame indices, the Adam step counter) are
        refreshed on the stream, then the captured window -- ~250 launches per timestep -- replays as one graph."""
        dev = self.wg.device
        B = ev["B"]
        # the DEM normalisation bounds are kernel ARGUMENTS (frozen into the graph): one capture per (shape, bounds), i.e. per
        # catchment; the event's tensors are copied into static buffers before every replay
        key = (steps, B, ev["T"], tuple(ev["rain"].shape), ev["dem_min"], ev["dem_max"], self.lr, self.matrix_mode)   # lr: one re-capture per epoch
        # (a scratch buffer that grew since the capture -- an eager call with a larger batch through the same arena -- leaves the
        # graph pointing at freed memory: ops.Arena.generation tells)
        G = self._graphs.get(key)
        if G is not None and G["arena_gen"] != self.wg.arena_generation():
            # a scratch buffer grew since some capture: EVERY cached graph may point at freed memory
            self._graphs.clear()
            G = None
        if G is None:
            from .general import initialize_states
            zero = [s.to(dev).repeat(B, 1, 1, 1) for s in initialize_states(dev, self.wg.H, self.wg.W)]
            sev = dict(ev)
            for k in ("rain", "cumsum", "dem", "imperv", "manhole"):
                sev[k] = ev[k].clone()
            G = {"key": key, "ev": sev, "tgt": torch.zeros((B, steps, self.wg.H, self.wg.W), device=dev),
                 "states": [torch.zeros_like(z) for z in zero], "zero": zero,
                 "t_devs": [torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(steps)],
                 "step_dev": torch.zeros(1, dtype=torch.int32, device=dev)}
            keep = (self.flat.clone(), self.m.clone(), self.v.clone())
            G["step_dev"].fill_(1)
            self._invalidate_packed()
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            # Eager warm-up (sizes every workspace, packs every weight) and capture are PER RANK decisions -- under DDP each rank
            # sees its own catchments (DistributedSampler), so one rank may (re)capture while another only replays.  Both must
            # therefore be collective-free: a reduce issued here would pair with another rank's real gradient reduce and the
            # sequences would stay out of step from then on.  The only collectives of a window are the two between the replays.
            was_distributed, self.distributed = self.distributed, False
            try:
                with torch.cuda.stream(side):
                    self._window_body(sev, G["tgt"], 0, steps, G["states"], t_devs=G["t_devs"], step_dev=G["step_dev"])
            finally:
                self.distributed = was_distributed
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            for dst, src in zip((self.flat, self.m, self.v), keep):
                dst.copy_(src)
            self._invalidate_packed()
            if self.distributed:
                cap = torch.cuda.Stream(device=dev)
                cap.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(cap):
                    graphs, out, clip = self._window_body_ddp_graphs(sev, G, steps)
                torch.cuda.current_stream(dev).wait_stream(cap)
                G.update(graph=graphs, out=out, clip=clip)
            else:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    out, clip = self._window_body(sev, G["tgt"], 0, steps, G["states"], t_devs=G["t_devs"], step_dev=G["step_dev"])
                G.update(graph=g, out=out, clip=clip)
            if any(h["arena_gen"] != self.wg.arena_generation() for h in self._graphs.values()):
                self._graphs.clear()            # the warm-up grew a buffer older graphs point into
            G["arena_gen"] = self.wg.arena_generation()
            # alternating catchments replay their own graphs instead of re-capturing every window; each entry owns a memory pool
            # with the window's activations, so the cache is small and least-recently-used entries go first
            while len(self._graphs) >= self.MAX_CACHED_WINDOWS:
                self._graphs.pop(next(iter(self._graphs)))
            self._graphs[key] = G
        else:
            self._graphs[key] = self._graphs.pop(key)      # most recently used last
        if ev is not G["ev"]:
            for k in ("rain", "cumsum", "dem", "imperv", "manhole"):
                if G["ev"][k].data_ptr() != ev[k].data_ptr():
                    G["ev"][k].copy_(ev[k])
        G["tgt"].copy_(targets)
        for dst, src in zip(G["states"], states if states is not None else G["zero"]):
            dst.copy_(src)
        for s, td in enumerate(G["t_devs"]):
            td.fill_(int(t0) + s)
        G["step_dev"].fill_(self.step_count)
        if self.distributed:
            # graph | the head's share of the gradient mean starts (RCCL: asynchronous, on its own stream) | graph | the rest of the
            # mean, join | graph: the same kernels and the same reduction order as the eager DDP window
            from .distributed import OverlappedGradientMean
            gA, gB, gC = G["graph"]
            reducer = OverlappedGradientMean(self.gflat, self.head_offset, group=self.pg)
            gA.replay()
            reducer.start_tail()
            gB.replay()
            reducer.finish()
            gC.replay()
        else:
            G["graph"].replay()
        self._invalidate_packed()       # host-side caches of packed weights now describe the previous parameters
        return G["out"], G["clip"]

    def train_window(self, event, targets, t0, steps, states=None):
        """One window: returns (loss components, final states); the parameters have been updated."""
        ev = event if "rain" in event else event_to_device(event, self.wg.device)
        self.step_count += 1
        with ops.matrix_mode(self.matrix_mode):
            if self.use_graph:
                targets = torch.as_tensor(targets, dtype=torch.float32, device=self.wg.device)
                out, clip = self._train_window_graph(ev, targets, t0, steps, states)
                self.last = {"loss": out["loss"], "clip": clip, "reg": out["reg"]}        # static buffers: valid until the next window
                return out["loss"].clone(), [s.clone() for s in out["states"]]
            out, clip = self._window_body(ev, targets, t0, steps, states)
        self.last = {"loss": out["loss"], "clip": clip, "reg": out["reg"]}
        return out["loss"], out["states"]

    def prewarm(self, event, ind):
        """SWP pre-warming (main.py:540-595): the six states after a gradient-free rollout of frames 0 .. ind-1 from zeros."""
        ev = event if "rain" in event else event_to_device(event, self.wg.device)
        from .general import initialize_states
        states = [s.to(self.wg.device).repeat(ev["B"], 1, 1, 1) for s in initialize_states(self.wg.device, self.wg.H, self.wg.W)]
        with ops.matrix_mode(self.matrix_mode):
            for t in range(int(ind)):
                _, states = self.wg._forward_step(ev, t, states, 0)
        return states

    def train_event(self, event, label, seq_num, window_size=None, loc=0, prewarming=False, starts=None):
        """All windows of one sample in order (or in the order of ``starts``, e.g. the shuffled plan of ``plan_windows``).  Fast mode (default): states carried between windows; ``prewarming=True``: the
        paper's schedule, every window starts from a gradient-free rollout from frame 0 (main.py:655-672).  label (B,T,H,W)
        normalised depths.  Returns (per-window loss components, final states)."""
        ev = event if "rain" in event else event_to_device(event, self.wg.device)
        label = torch.as_tensor(label, dtype=torch.float32, device=self.wg.device)
        T = label.shape[1]
        window_size = T - loc if window_size is None else window_size
        states, losses = None, []
        self.refresh_weight_ranges()
        try:
            for ind in (window_starts(loc, seq_num, window_size) if starts is None else starts):
                if prewarming:
                    states = self.prewarm(ev, ind) if ind > 0 else None
                loss, states = self.train_window(ev, label[:, ind:ind + seq_num], ind, seq_num, states)
                losses.append(loss)
        finally:
            # after the event's last optimizer step: flags for the weights as they are NOW, then the layers check for themselves again
            self.refresh_weight_ranges()
            self.release_weight_ranges()
        return losses, states
