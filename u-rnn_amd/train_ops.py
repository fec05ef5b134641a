"""ctypes wrappers of the training entry points of the C ABI: backward of the ConvGRU / Skip-ConvGRU cell, the
stage convs, the transposed convs and the head; the loss; the clipped Adam step.  ``urnn_amd.training`` assembles them into SWP
windows and the training loop.  A backward call reads the scratch buffer its forward ran on (``fwd_ws``)."""
import torch

from . import ops
from ._lib import check, lib


def _bwd_scratch(scratch, nbytes, dev):
    """Backward scratch: ``scratch`` is the owner's ``ops.Arena`` (buffer "bwd", shared by all backward calls of a stream) or None
    (a fresh buffer for this call)."""
    return ops._scratch(scratch, nbytes, dev, key="bwd")


def gru_cell_backward(x, e, h, W1, W2, gn1_w, gn2_w, dh_out, I, fwd_ws, grads=None, accumulate=False, packed=None, dh_out2=None,
                      scratch=None, dh_out3=None, dh_out4=None, split_dh=False):
    """Gradients of one cell step.  ``fwd_ws`` is the scratch buffer the forward ``ops.gru_cell(x, e, h, ..., ws=fwd_ws)`` of the
    SAME inputs ran on (untouched since): it holds the raw gates / candidate and the GroupNorm statistics.
    W1 (2F,K[,1,1]) / W2 (F,K[,1,1]): conv weights in the reference layout.  Returns a dict with dx, de (when given), dh and
    dW1, db1, dg1, dbe1, dW2, db2, dg2, dbe2; pass ``grads`` (the dict of a previous call) with ``accumulate=True`` to add
    the parameter gradients of another timestep.  ``packed``: a one-element list the caller keeps per cell for the packed
    weights of the input-gradient GEMMs -- filled on the first call, reused until the caller empties it (after an optimizer
    step); None: packed on every call.  ``dh_out2`` .. ``dh_out4``: optional further terms of dL/dh' (added on the fly).  ``split_dh``:
    dL/dh is returned as two terms, ``dh`` + ``dh2`` (the second straight out of its GEMM), for a consumer that sums its terms itself --
    the previous timestep's call."""
    ops._dev_check(x, e, h, W1, W2, gn1_w, gn2_w, dh_out, dh_out2, dh_out3, dh_out4)
    B, F, H, W = h.shape
    K = I + (F if e is not None else 0) + F
    if W1.numel() != 2 * F * K or W2.numel() != F * K:
        raise RuntimeError(f"gru_cell_backward: weights do not match I={I}, F={F}, skip={e is not None}")
    L = lib()
    dev = h.device
    fwd = ops._scratch(fwd_ws, L.urnn_gru_cell_workspace_bytes(B, F, H, W), dev)
    ws = _bwd_scratch(scratch, L.urnn_gru_cell_backward_workspace_bytes(B, I, F, int(e is not None), H, W), dev)
    f32 = dict(dtype=torch.float32, device=dev)
    g = grads if grads is not None else {}
    if "dW1" not in g:          # caller-provided buffers (e.g. views of a flat gradient buffer) are written in place
        g.update(dW1=torch.empty(2 * F, K, **f32), db1=torch.empty(2 * F, **f32), dg1=torch.empty(2 * F, **f32),
                 dbe1=torch.empty(2 * F, **f32), dW2=torch.empty(F, K, **f32), db2=torch.empty(F, **f32),
                 dg2=torch.empty(F, **f32), dbe2=torch.empty(F, **f32))
        accumulate = False
    g["dh"] = torch.empty_like(h)
    g["dh2"] = torch.empty_like(h) if split_dh else None
    if x is not None and e is not None and B == 1:        # one block: a single GEMM writes dx | de in place
        dxe = torch.empty((1, I + F, H, W), **f32)
        g["dx"], g["de"] = dxe[:, :I], dxe[:, I:]
    else:
        g["dx"] = torch.empty_like(x) if x is not None else None
        g["de"] = torch.empty_like(e) if e is not None else None
    repack = packed is None or not packed
    if repack:
        pk = torch.empty(L.urnn_gru_cell_backward_packed_floats(I, F, int(e is not None)), **f32)
        if packed is not None:
            packed.append(pk)
    else:
        pk = packed[0]
    p = ops._ptr
    check(L.urnn_gru_cell_backward_f32(p(x), p(e), p(h), p(W1), p(W2), p(gn1_w), p(gn2_w), p(fwd), p(dh_out), p(dh_out2), p(dh_out3),
                                       p(dh_out4), p(g["dx"]), p(g["de"]),
                                       p(g["dh"]), p(g["dh2"]), p(g["dW1"]), p(g["db1"]), p(g["dg1"]), p(g["dbe1"]), p(g["dW2"]), p(g["db2"]),
                                       p(g["dg2"]), p(g["dbe2"]), p(pk), int(repack), p(ws), ws.numel(), B, I, F, H, W,
                                       int(bool(accumulate)), ops._stream()), "urnn_gru_cell_backward_f32")
    return g


def weight_gradient(dy, segs, dW=None, db=None, accumulate=False, scratch=None):
    """dW (N,K) = sum over pixels of dy (B,N,H,W) x cat(segs) (B,K,H,W) (1x1-conv weight gradient), db (N) likewise; segs: one to
    three (B,C_i,H,W) tensors."""
    L = lib()
    B, N, H, W = dy.shape
    Cs = [int(t.shape[1]) for t in segs] + [0] * (3 - len(segs))
    K = sum(Cs)
    dev = dy.device
    dW = torch.empty(N, K, device=dev) if dW is None else dW
    db = torch.empty(N, device=dev) if db is None else db
    ws = _bwd_scratch(scratch, L.urnn_weight_gradient_workspace_bytes(B, N, K, H, W), dev)
    p = ops._ptr
    ptrs = [p(t) for t in segs] + [None] * (3 - len(segs))
    check(L.urnn_weight_gradient_f32(p(dy), ptrs[0], Cs[0], ptrs[1], Cs[1], ptrs[2], Cs[2], p(dW), p(db), p(ws), ws.numel(), B, N,
                                     H, W, int(accumulate), ops._stream()), "urnn_weight_gradient_f32")
    return dW, db


def _layer_packs(packed, nfloats, dev):
    """Caller-kept packed weights of a layer's backward GEMMs: ``packed`` is None (pack per call) or a list that receives the
    buffer on first use and hands it back afterwards (the caller clears it when the parameters change).  -> (buffer, repack)"""
    if packed is None:
        return None, 1
    if not packed:
        packed.append(torch.empty(nfloats, dtype=torch.float32, device=dev))
        return packed[0], 1
    return packed[0], 0


def stage_conv_backward(x, weight, bias, dout, pool, dweight=None, dbias=None, accumulate=False, slope=ops.LRELU_SLOPE, scratch=None,
                        packed=None):
    """Backward of ``ops.stage_conv`` (conv1x1 + LeakyReLU [+ AvgPool2]).  weight (Cout,Cin[,1,1]), bias (Cout).
    Returns (dx, dweight, dbias)."""
    ops._dev_check(x, weight, bias, dout)
    B, Cin, H, W = x.shape
    Cout = weight.shape[0]
    L = lib()
    ws = _bwd_scratch(scratch, L.urnn_stage_conv_backward_workspace_bytes(B, Cin, Cout, H, W), x.device)
    if dweight is None:
        dweight, dbias, accumulate = torch.empty_like(weight), torch.empty_like(bias), False
    dx = torch.empty_like(x)
    p = ops._ptr
    pk, repack = _layer_packs(packed, L.urnn_stage_conv_backward_packed_floats(Cin, Cout), x.device)
    check(L.urnn_stage_conv_backward_f32(p(x), p(weight), p(bias), p(dout), p(dx), p(dweight), p(dbias), p(pk), repack, p(ws), ws.numel(),
                                         B, Cin, Cout, H, W, int(bool(pool)), slope, int(bool(accumulate)), ops._stream()),
          "urnn_stage_conv_backward_f32")
    return dx, dweight, dbias


def deconv2x2_backward(x, weight, out, dout, dweight=None, dbias=None, accumulate=False, slope=ops.LRELU_SLOPE, scratch=None,
                       packed=None):
    """Backward of ``ops.deconv2x2``.  weight (Cin,Cout,2,2); ``out`` is the forward output.  Returns (dx, dweight, dbias)."""
    ops._dev_check(x, weight, out, dout)
    B, Cin, H, W = x.shape
    Cout = weight.shape[1]
    L = lib()
    ws = _bwd_scratch(scratch, L.urnn_deconv2x2_backward_workspace_bytes(B, Cin, Cout, H, W), x.device)
    if dweight is None:
        dweight, dbias, accumulate = torch.empty_like(weight), torch.empty(Cout, dtype=torch.float32, device=x.device), False
    dx = torch.empty_like(x)
    p = ops._ptr
    pk, repack = _layer_packs(packed, L.urnn_deconv2x2_backward_packed_floats(Cin, Cout), x.device)
    check(L.urnn_deconv2x2_backward_f32(p(x), p(weight), p(out), p(dout), p(dx), p(dweight), p(dbias), p(pk), repack, p(ws), ws.numel(), B,
                                        Cin, Cout, H, W, slope, int(bool(accumulate)), ops._stream()), "urnn_deconv2x2_backward_f32")
    return dx, dweight, dbias


def head_backward(feat, conv_w, ln_w, ln_b, reg_w, out_raw, out_cls, dout, cls_thred, fwd_ws, grads=None, accumulate=False,
                  slope=ops.LRELU_SLOPE, scratch=None):
    """Backward of ``ops.head`` w.r.t. the masked depth.  ``fwd_ws``: the scratch the forward on the same ``feat`` ran on
    (``ops.head(..., want_raw=True, ws=fwd_ws)``, untouched since: it holds the five LayerNorm statistics).  conv_w (5,C,C), ln_w / ln_b (5,C,H,W), reg_w (C).
    Returns a dict: dfeat, dconv_w, dln_w, dln_b, dreg_w, dreg_b."""
    ops._dev_check(feat, conv_w, ln_w, ln_b, reg_w, out_raw, out_cls, dout)
    B, C, H, W = feat.shape
    L = lib()
    fwd = ops._scratch(fwd_ws, L.urnn_head_workspace_bytes(B, C, H, W), feat.device)
    ws = _bwd_scratch(scratch, L.urnn_head_backward_workspace_bytes(B, H, W), feat.device)
    f32 = dict(dtype=torch.float32, device=feat.device)
    g = grads if grads is not None else {}
    if grads is None or not accumulate:
        g.update(dconv_w=torch.empty_like(conv_w), dln_w=torch.empty_like(ln_w), dln_b=torch.empty_like(ln_b),
                 dreg_w=torch.empty(C, **f32), dreg_b=torch.empty(1, **f32))
    g["dfeat"] = torch.empty_like(feat)
    p = ops._ptr
    check(L.urnn_head_backward_f32(p(feat), p(conv_w), p(ln_w), p(ln_b), p(reg_w), p(fwd), p(out_raw), p(out_cls), p(dout), p(g["dfeat"]),
                                   p(g["dconv_w"]), p(g["dln_w"]), p(g["dln_b"]), p(g["dreg_w"]), p(g["dreg_b"]), p(ws), ws.numel(),
                                   B, C, H, W, float(cls_thred), slope, int(bool(accumulate)), ops._stream()), "urnn_head_backward_f32")
    return g


def loss(reg, target, cls_thred=0.0, want_grad=True, scratch=None):
    """FocalBCE_and_WMSE on a window's outputs.  Returns (components: 5-float device tensor [loss, loss_reg, wet MSE, dry MSE,
    loss_cls], dreg or None)."""
    ops._dev_check(reg, target)
    n = reg.numel()
    if target.numel() != n:
        raise RuntimeError("loss: reg and target differ in size")
    L = lib()
    ws = _bwd_scratch(scratch, L.urnn_loss_workspace_bytes(n), reg.device)
    comps = torch.empty(5, dtype=torch.float32, device=reg.device)
    dreg = torch.empty_like(reg) if want_grad else None
    check(L.urnn_loss_f32(ops._ptr(reg), ops._ptr(target), float(cls_thred), ops._ptr(comps), ops._ptr(dreg), ops._ptr(ws), ws.numel(), n,
                          ops._stream()), "urnn_loss_f32")
    return comps, dreg


def adam_step(params, grads, exp_avg, exp_avg_sq, step, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=0.0, step_dev=None,
              scratch=None):
    """In-place Adam step on flat float32 buffers (with optional global-norm clipping).  ``step_dev`` (int32 device scalar)
    overrides ``step`` for graph replay.  Returns a 2-float device tensor: (clip coefficient, gradient norm)."""
    ops._dev_check(params, grads, exp_avg, exp_avg_sq)
    if step_dev is not None and (not step_dev.is_cuda or step_dev.dtype != torch.int32):
        raise RuntimeError("adam_step: step_dev must be an int32 device scalar")
    n = params.numel()
    L = lib()
    ws = _bwd_scratch(scratch, L.urnn_adam_workspace_bytes(n), params.device)
    out = torch.empty(2, dtype=torch.float32, device=params.device)
    p = ops._ptr
    check(L.urnn_adam_step_f32(p(params), p(grads), p(exp_avg), p(exp_avg_sq), n, float(lr), float(betas[0]), float(betas[1]), float(eps),
                               int(step), p(step_dev), float(max_grad_norm), p(out), p(ws), ws.numel(), ops._stream()), "urnn_adam_step_f32")
    return out
