"""Three-stage Skip-ConvGRU decoder on the HIP path (mirror of reference decoder.py:59-217).

Stage order deepest -> shallowest; decoder state list is [stage3 (quarter res), stage2, stage1 (full res)]
exactly as the reference orders it (decoder.py:198-212, general.py:82-89).  The encoder state of the SAME
timestep is the skip input; the ``torch.cat((enc, dec))`` of decoder.py:135 never materialises -- the kernel
reads the two tensors as separate K segments."""
import torch
from torch import nn

from .utils import make_layers


class Decoder(nn.Module):
    def __init__(self, clstm, subnets, rnns, use_checkpoint):
        super().__init__()
        if len(subnets) != 3 or len(rnns) != 3:
            raise NotImplementedError("the decoder has three stages (net_params.py:102-139)")
        self.blocks = 3
        self.stage3, self.stage2, self.stage1 = (make_layers(s) for s in subnets)
        self.rnn3, self.rnn2, self.rnn1 = rnns
        self.clstm = clstm
        self.use_checkpoint = use_checkpoint

    @torch.no_grad()
    def forward_by_stage(self, i, inputs, encoder_states, decoder_states=None):
        rnn, stage = getattr(self, f"rnn{i}"), getattr(self, f"stage{i}")
        e = encoder_states.contiguous()
        d = torch.zeros_like(e) if decoder_states is None else decoder_states.contiguous()
        x = None
        if inputs is not None:
            S, B, C, H, W = inputs.shape
            if S != 1:
                raise NotImplementedError("S == 1 only (see Encoder.forward_by_stage)")
            x = inputs.reshape(B, C, H, W)
        h = rnn.step(x, e, d)
        y = stage(h)
        return y.unsqueeze(0), h

    @torch.no_grad()
    def forward(self, encoder_states, decoder_states):
        states = []
        inputs, st = self.forward_by_stage(3, None, encoder_states[-1], decoder_states[0])
        states.append(st)
        for i in (2, 1):
            inputs, st = self.forward_by_stage(i, inputs, encoder_states[i - 1], decoder_states[self.blocks - i])
            states.append(st)
        return inputs.transpose(0, 1), tuple(states)
