"""Three-stage encoder on the HIP path (mirror of reference encoder.py:64-215).

Per stage: fused Conv1x1+LeakyReLU[+AvgPool2] kernel, then one ConvGRU cell call.  ``forward(inputs,
state_stages)`` keeps the reference contract: inputs (S,B,C,H,W) with S == 1, returns the three new states."""
import torch
from torch import nn

from .utils import make_layers


class Encoder(nn.Module):
    def __init__(self, clstm, subnets, rnns, use_checkpoint):
        super().__init__()
        if clstm:
            raise NotImplementedError("ConvLSTM is dead code in the reference (no LSTM cell class exists; SURVEY F9)")
        if len(subnets) != 3 or len(rnns) != 3:
            raise NotImplementedError("the encoder has three stages (net_params.py:52-100)")
        self.blocks = 3
        self.use_checkpoint = use_checkpoint
        self.clstm = clstm
        self.stage1, self.stage2, self.stage3 = (make_layers(s) for s in subnets)
        self.rnn1, self.rnn2, self.rnn3 = rnns

    @torch.no_grad()
    def forward_by_stage(self, i, inputs, hidden_state, subnet, rnn):
        S, B, C, H, W = inputs.shape
        if S != 1:
            raise NotImplementedError("the reference only ever calls the model with S == 1 timesteps per forward "
                                      "(test.py:356-365; CGRU_cell consumes inputs[0] only) -- roll out on the host")
        x = subnet(inputs.reshape(B, C, H, W))
        if hidden_state is None:
            hidden_state = torch.zeros(B, rnn.num_features, x.shape[2], x.shape[3], device=x.device)
        h = rnn.step(x, None, hidden_state.contiguous())
        return h.unsqueeze(0), h

    @torch.no_grad()
    def forward(self, inputs, state_stages):
        states = []
        for i in (1, 2, 3):
            inputs, st = self.forward_by_stage(i, inputs, state_stages[i - 1], getattr(self, f"stage{i}"), getattr(self, f"rnn{i}"))
            states.append(st)
        return tuple(states)
