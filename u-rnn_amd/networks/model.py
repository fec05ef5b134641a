"""``ED``: the U-RNN encoder-decoder model on the HIP path (mirror of reference model.py:22-121)."""
import torch
from torch import nn

from .decoder import Decoder
from .encoder import Encoder
from .head import YOLOXHead


def canonical_state_dict(state_dict):
    """Reduce a reference checkpoint's 254-key ``state_dict`` to the 79 unique tensors.

    The reference registers every gradient-checkpoint wrapper as a sub-module, so each parameter appears
    under alias keys (``*_wrapper.module.*``, ``conv{1,2}_module_wrapper.module.*`` -- ConvRNN.py:108-109,
    encoder.py:106-117, decoder.py:95-100, flood_head.py:121-125); DDP adds a ``module.`` prefix
    (test.py:399-403).  Aliases are dropped, the prefix stripped."""
    out = {}
    for key, value in state_dict.items():
        if key.startswith("module."):
            key = key[len("module."):]
        if "_wrapper.module." in key or "_module_wrapper." in key:
            continue
        out[key] = value
    return out


class ED(nn.Module):
    def __init__(self, clstm_flag, encoder_params, decoder_params, cls_thred=0.5, use_checkpoint=True,
                 input_height=500, input_width=500):
        super().__init__()
        self.encoder = Encoder(clstm_flag, encoder_params[0], encoder_params[1], use_checkpoint=use_checkpoint)
        self.decoder = Decoder(clstm_flag, decoder_params[0], decoder_params[1], use_checkpoint=use_checkpoint)
        self.head = YOLOXHead(cls_thred, use_checkpoint=use_checkpoint, input_height=input_height, input_width=input_width)

    def load_state_dict(self, state_dict, strict=True, **kw):
        return super().load_state_dict(canonical_state_dict(state_dict), strict=strict, **kw)

    @torch.no_grad()
    def forward(self, input_t, prev_encoder_state1, prev_encoder_state2, prev_encoder_state3,
                prev_decoder_state1, prev_decoder_state2, prev_decoder_state3):
        """Same contract as the reference (model.py:65-121): ``input_t`` (B,S=1,C,H,W) plus six states ->
        (reg (B,S,H,W), e1, e2, e3, d1, d2, d3) with d1 the deepest decoder state."""
        x = input_t.permute(1, 0, 2, 3, 4)
        enc = self.encoder(x, [prev_encoder_state1, prev_encoder_state2, prev_encoder_state3])
        feat, dec = self.decoder(enc, [prev_decoder_state1, prev_decoder_state2, prev_decoder_state3])
        out = self.head(feat)           # (B,S,2,H,W)
        reg = out[:, :, 0]
        return (reg,) + tuple(enc) + tuple(dec)
