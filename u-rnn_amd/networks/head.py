"""Dual prediction head on the HIP path (mirror of reference head/flood_head.py:39-202 and
head/network_blocks.py:74-171).

``BaseConv`` / ``finalConv`` are parameter containers with the reference's child names (``conv``, ``ln``) so the
``head.*`` checkpoint keys load unchanged; ``YOLOXHead.forward`` is one ``urnn_head_f32`` call (four streaming
passes over the 16-channel feature map and the five (16,H,W) LayerNorm affines)."""
import torch
from torch import nn

from .. import ops


class BaseConv(nn.Module):
    """Conv1x1 (no bias) -> LayerNorm([C,H,W]) -> SiLU (network_blocks.py:74-101); container only."""

    def __init__(self, in_channels, out_channels, ksize, stride, groups=1, bias=False, act="silu", height=None, width=None):
        super().__init__()
        if ksize != 1 or stride != 1 or groups != 1 or bias or act != "silu" or height is None or width is None:
            raise NotImplementedError("head blocks are Conv1x1(no bias)+LayerNorm([C,H,W])+SiLU (flood_head.py:81-108)")
        self.conv = nn.Conv2d(in_channels, out_channels, 1, 1, 0, bias=False)
        self.ln = nn.LayerNorm([out_channels, height, width])


class finalConv(nn.Module):
    """Conv1x1 (+bias) -> activation, no norm (network_blocks.py:129-171); container only."""

    def __init__(self, in_channels, out_channels, ksize, stride, groups=1, bias=False, act="leaky", norm="gn"):
        super().__init__()
        if ksize != 1 or stride != 1 or groups != 1 or norm != "":
            raise NotImplementedError("prediction convs are norm-free Conv1x1 (flood_head.py:111-118)")
        self.conv = nn.Conv2d(in_channels, out_channels, 1, 1, 0)  # the reference ignores `bias` here too
        self.act_name = act


class YOLOXHead(nn.Module):
    def __init__(self, cls_thred=0.5, in_channels=64, width=0.25, depthwise=False, use_checkpoint=True,
                 input_height=500, input_width=500):
        super().__init__()
        if depthwise:
            raise NotImplementedError("depthwise head is never built by the reference (model.py:62-63)")
        ch = int(in_channels * width)
        H, W = input_height, input_width
        self.acts = ["silu"] * 3 + ["sigmoid", "lrelu"]
        self.stems = BaseConv(ch, ch, 1, 1, act="silu", height=H, width=W)
        self.cls_convs = nn.Sequential(BaseConv(ch, ch, 1, 1, act="silu", height=H, width=W),
                                       BaseConv(ch, ch, 1, 1, act="silu", height=H, width=W))
        self.reg_convs = nn.Sequential(BaseConv(ch, ch, 1, 1, act="silu", height=H, width=W),
                                       BaseConv(ch, ch, 1, 1, act="silu", height=H, width=W))
        self.cls_preds = finalConv(ch, 1, 1, 1, act="sigmoid", norm="")
        self.reg_preds = finalConv(ch, 1, 1, 1, act="lrelu", norm="")
        self.use_checkpoint = use_checkpoint
        self.cls_thred = cls_thred
        self.channels = ch
        self._stamp = None
        self._flat = None

    def _blocks(self):
        return [self.stems, self.cls_convs[0], self.cls_convs[1], self.reg_convs[0], self.reg_convs[1]]

    def flat_params(self):
        """Stack the five blocks' parameters into the contiguous (5, ...) buffers the kernel streams."""
        blocks = self._blocks()
        params = [p for b in blocks for p in (b.conv.weight, b.ln.weight, b.ln.bias)]
        stamp = tuple((p.data_ptr(), p._version) for p in params)
        if self._flat is None or stamp != self._stamp:
            ch = self.channels
            self._flat = {
                "conv_w": torch.stack([b.conv.weight.detach().reshape(ch, ch) for b in blocks]).contiguous(),
                "ln_w": torch.stack([b.ln.weight.detach() for b in blocks]).contiguous(),
                "ln_b": torch.stack([b.ln.bias.detach() for b in blocks]).contiguous(),
            }
            self._stamp = stamp
        return self._flat

    def run(self, feat, out_masked=None, out_cls=None, out_raw=None, frame_index=None, want_raw=False, ws=None, partial0=None, coop=False,
            frame_next=None):
        """feat (B,16,H,W) -> (masked, cls, raw|None).  ``partial0``, ``frame_next``: see ops.head."""
        fp = self.flat_params()
        return ops.head(feat, fp["conv_w"], fp["ln_w"], fp["ln_b"],
                        self.cls_preds.conv.weight.detach().reshape(-1), self.cls_preds.conv.bias.detach(),
                        self.reg_preds.conv.weight.detach().reshape(-1), self.reg_preds.conv.bias.detach(),
                        self.cls_thred, out_masked=out_masked, out_cls=out_cls, out_raw=out_raw,
                        frame_index=frame_index, want_raw=want_raw, eps=self.stems.ln.eps, ws=ws, partial0=partial0, coop=coop,
                        frame_next=frame_next)

    @torch.no_grad()
    def forward(self, inputs):
        """inputs (S,B,16,H,W) [any leading two dims, as the reference only flattens them] -> (S,B,2,H,W)
        with channel 0 = masked depth, channel 1 = wet probability (flood_head.py:131-177)."""
        d0, d1, C, H, W = inputs.shape
        feat = inputs.reshape(d0 * d1, C, H, W).contiguous()
        masked, cls, _ = self.run(feat)
        return torch.stack([masked, cls], dim=1).reshape(d0, d1, 2, H, W)

    def correction_depth(self, reg_output_t, cls_output_t, flood_thres=0.5):
        """Kept for API parity (flood_head.py:179-202); the kernel applies the same mask in its last pass."""
        return reg_output_t * (cls_output_t >= flood_thres).float()
