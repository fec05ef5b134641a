"""Layer factory mirror (reference utils.py:73-125): builds the stage conv / deconv blocks from the same
OrderedDict specs, as modules whose children carry the reference's names (``conv1_leaky_1`` ...)."""
from collections import OrderedDict

import torch
from torch import nn

from .. import ops
from ._packing import PackedCache


class StageLayers(nn.Module):
    """One encoder/decoder stage: Conv1x1+LeakyReLU[+AvgPool2] or ConvTranspose2x2+LeakyReLU, fused in one
    HIP kernel.  Child module names equal the reference's ``nn.Sequential`` keys so checkpoints load."""

    def __init__(self, block):
        super().__init__()
        self.kind = None
        self.pool = False
        self._pname = None
        for layer_name, v in block.items():
            v = [int(x) for x in v]
            if "avgpool" in layer_name:
                if v != [2, 2, 0]:
                    raise NotImplementedError("only AvgPool2d(2, 2, 0) is implemented (net_params.py:84-85)")
                if self.kind != "conv":
                    raise NotImplementedError("avgpool must follow a conv layer")
                self.pool = True
                self.add_module(layer_name, nn.AvgPool2d(2, 2, 0))
            elif "deconv" in layer_name:
                if self.kind is not None or v[2:] != [2, 2, 0]:
                    raise NotImplementedError("only a single ConvTranspose2d(k=2, s=2, p=0) per stage is implemented")
                self.kind, self._pname = "deconv", layer_name
                self.add_module(layer_name, nn.ConvTranspose2d(v[0], v[1], v[2], v[3], v[4]))
                self.add_module("lrelu_" + layer_name, nn.LeakyReLU(0.2))
            elif "conv" in layer_name:
                if self.kind is not None or v[2:] != [1, 1, 0]:
                    raise NotImplementedError("only a single Conv2d(k=1, s=1, p=0) per stage is implemented")
                self.kind, self._pname = "conv", layer_name
                self.add_module(layer_name, nn.Conv2d(v[0], v[1], v[2], v[3], v[4]))
                self.add_module("lrelu_" + layer_name, nn.LeakyReLU(0.2))
            else:
                raise NotImplementedError(layer_name)
        self._cache = PackedCache()

    @property
    def layer(self):
        return getattr(self, self._pname)

    @property
    def out_channels(self):
        return self.layer.out_channels

    def _packed(self):
        L = self.layer
        fn = ops.pack_conv if self.kind == "conv" else ops.pack_deconv
        return self._cache.get((L.weight, L.bias), lambda: fn(L.weight.detach(), L.bias.detach()), weights=(L.weight,))

    @torch.no_grad()
    def forward(self, x, out=None, head_w=None, head_partial0=None):
        """``head_w`` / ``head_partial0``: ops.stage_conv (the decoder's last conv taking the head's first statistics)."""
        x = x.contiguous()
        packed = self._packed()
        with ops.exact_matrix_if(self._cache.wide):       # a weight beyond the f16 pieces' range: exact fp32 MFMA for this layer
            if self.kind == "conv":
                return ops.stage_conv(x, packed, self.out_channels, self.pool, out=out, head_w=head_w, head_partial0=head_partial0)
            return ops.deconv2x2(x, packed, self.out_channels, out=out)


def make_layers(block, norm_name="", act="lrelu"):
    if norm_name != "" or act != "lrelu":
        raise NotImplementedError("the reference only ever builds norm-free LeakyReLU stages (encoder.py:97-99)")
    return StageLayers(OrderedDict(block))
