"""``urnn_step_f32``: one inference timestep of ``ED.forward`` (reference model.py:65-121) behind ONE call of the C ABI.

This module is the Python view of what a non-Python host does with ``include/urnn_hip.h``: describe the network once as a
``urnn_net_f32`` (packed slabs + channel counts), size and initialise one workspace, then enqueue ``urnn_step_f32`` per frame -- directly
or inside a captured hipGraph.  The launches are the ones ``RolloutEngine(overlap=False)`` makes through the per-module entries, so the
results are bit-identical to it (tests/test_hip_rollout.py::test_c_abi_step_equals_the_one_chain_engine)."""
import ctypes

import torch

from . import ops
from ._lib import check, lib

_P3 = ctypes.c_void_p * 3


class UrnnNet(ctypes.Structure):
    """``urnn_net_f32`` of include/urnn_hip.h, field for field."""
    _fields_ = [("in_channels", ctypes.c_int), ("enc_stage_out", ctypes.c_int * 3), ("enc_features", ctypes.c_int * 3),
                ("dec_zero_input_channels", ctypes.c_int), ("dec_features", ctypes.c_int * 3), ("dec_stage_out", ctypes.c_int * 2),
                ("feat_channels", ctypes.c_int),
                ("enc_stage", _P3), ("enc_cell", _P3), ("enc_gn1_w", _P3), ("enc_gn1_b", _P3), ("enc_gn2_w", _P3), ("enc_gn2_b", _P3),
                ("dec_cell", _P3), ("dec_gn1_w", _P3), ("dec_gn1_b", _P3), ("dec_gn2_w", _P3), ("dec_gn2_b", _P3), ("dec_stage", _P3),
                ("head_conv_w", ctypes.c_void_p), ("head_ln_w", ctypes.c_void_p), ("head_ln_b", ctypes.c_void_p),
                ("cls_w", ctypes.c_void_p), ("cls_b", ctypes.c_void_p), ("reg_w", ctypes.c_void_p), ("reg_b", ctypes.c_void_p)]


class StepNet:
    """The ``urnn_net_f32`` of an ``ED`` network plus the workspace of ``urnn_step_f32`` for (B, H, W).  Keeps every tensor the
    structure points to alive; rebuild it after the parameters change (the slabs are packed copies)."""

    def __init__(self, net, B, H, W, in_channels):
        enc, dec, head = net.encoder, net.decoder, net.head
        self.device = next(net.parameters()).device
        self.B, self.H, self.W = int(B), int(H), int(W)
        self._keep = []
        n = UrnnNet()
        n.in_channels = int(in_channels)
        ecells, dcells = [enc.rnn1, enc.rnn2, enc.rnn3], [dec.rnn3, dec.rnn2, dec.rnn1]          # decoder: deepest first
        estages, dstages = [enc.stage1, enc.stage2, enc.stage3], [dec.stage3, dec.stage2, dec.stage1]

        def keep(t):
            t = t.detach().contiguous()
            self._keep.append(t)
            return t.data_ptr()

        for k in range(3):
            n.enc_stage_out[k] = estages[k].out_channels
            n.enc_features[k] = ecells[k].num_features
            n.dec_features[k] = dcells[k].num_features
            n.enc_stage[k] = keep(estages[k]._packed())
            n.dec_stage[k] = keep(dstages[k]._packed())
            for cells, pre in ((ecells, "enc"), (dcells, "dec")):
                c = cells[k]
                getattr(n, pre + "_cell")[k] = keep(c._packed())
                getattr(n, pre + "_gn1_w")[k] = keep(c.conv1[1].weight)
                getattr(n, pre + "_gn1_b")[k] = keep(c.conv1[1].bias)
                getattr(n, pre + "_gn2_w")[k] = keep(c.conv2[1].weight)
                getattr(n, pre + "_gn2_b")[k] = keep(c.conv2[1].bias)
        if any(m._cache.wide for m in ecells + dcells + estages + dstages):      # (set when the slabs above were packed)
            raise ValueError("a layer holds a weight beyond the f16 pieces' range: run it through the per-module entries (ops.exact_matrix_if)")
        n.dec_zero_input_channels = dec.rnn3.input_channels
        n.dec_stage_out[0], n.dec_stage_out[1] = dec.stage3.out_channels, dec.stage2.out_channels
        n.feat_channels = dec.stage1.out_channels
        fp = head.flat_params()
        n.head_conv_w, n.head_ln_w, n.head_ln_b = keep(fp["conv_w"]), keep(fp["ln_w"]), keep(fp["ln_b"])
        n.cls_w, n.cls_b = keep(head.cls_preds.conv.weight.reshape(-1)), keep(head.cls_preds.conv.bias)
        n.reg_w, n.reg_b = keep(head.reg_preds.conv.weight.reshape(-1)), keep(head.reg_preds.conv.bias)
        self.net = n
        self.cls_thred, self.eps = float(head.cls_thred), float(enc.rnn1.conv1[1].eps)
        L = lib()
        nbytes = L.urnn_step_workspace_bytes(ctypes.byref(n), self.B, self.H, self.W)
        if nbytes == 0:
            raise ValueError("urnn_step_workspace_bytes: the network description or the grid is not one urnn_step_f32 takes")
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        cs, hs = ctypes.c_void_p(), ctypes.c_void_p()
        check(L.urnn_step_workspace_init(ctypes.byref(n), self.ws.data_ptr(), nbytes, self.B, self.H, self.W, ctypes.byref(cs), ctypes.byref(hs),
                                         ops._stream()), "urnn_step_workspace_init")
        self._status_ptrs = (cs.value, hs.value)

    def step(self, x_t, states, out_masked, out_cls, out_raw=None, frame_index=None):
        """Enqueue one timestep on the current stream: ``x_t`` (B,C,H,W), ``states`` six tensors updated in place, outputs
        (T,B,H,W) frame buffers (frame ``*frame_index``, 0 without one)."""
        ptrs = (ctypes.c_void_p * 6)(*[s.data_ptr() for s in states])
        check(lib().urnn_step_f32(ctypes.byref(self.net), x_t.data_ptr(), ptrs, out_masked.data_ptr(), out_cls.data_ptr(),
                                  out_raw.data_ptr() if out_raw is not None else None,
                                  frame_index.data_ptr() if frame_index is not None else None, self.ws.data_ptr(), self.ws.numel(),
                                  self.B, self.H, self.W, self.cls_thred, self.eps, ops.LRELU_SLOPE, ops._stream()), "urnn_step_f32")

    def status(self):
        """(cell status word, head status word) -- synchronises."""
        off = [p - self.ws.data_ptr() for p in self._status_ptrs]
        words = self.ws.view(torch.int32)
        return tuple(int(words[o // 4]) for o in off)
