"""ConvGRU / Skip-ConvGRU cell on the HIP path.

Mirror of the reference ``CGRU_cell`` (ConvRNN.py:49-194): same constructor, same parameter layout
(``conv1 = Sequential(Conv2d, GroupNorm)``, ``conv2`` likewise -> identical ``state_dict`` keys), same
``forward(inputs, hidden_state, seq_len)`` contract.  The torch modules are parameter containers only; the
arithmetic is one ``urnn_gru_cell_f32`` call (gate GEMM + GroupNorm + sigmoid, candidate GEMM on r*h +
GroupNorm + tanh, blend).
"""
import torch
from torch import nn

from .. import ops
from ._packing import PackedCache


class CGRU_cell(nn.Module):
    def __init__(self, use_checkpoint, shape, input_channels, filter_size, num_features, module):
        super().__init__()
        if int(filter_size) != 1:
            raise NotImplementedError(
                "urnn_amd implements the published architecture: every gate convolution is 1x1 "
                "(configs/network.yaml filter_size: 1); got filter_size=%r" % (filter_size,))
        if module not in ("encoder", "decoder"):
            raise ValueError("module must be 'encoder' or 'decoder'")
        self.shape = tuple(shape)
        self.input_channels = int(input_channels)
        self.filter_size = int(filter_size)
        self.num_features = int(num_features)
        self.padding = 0
        self.module = module
        self.use_checkpoint = use_checkpoint  # accepted for API parity; inference path has no autograd graph
        F = self.num_features
        k_in = self.input_channels + (2 * F if module == "decoder" else F)
        # parameter containers with the reference's names/shapes (ConvRNN.py:94-104)
        self.conv1 = nn.Sequential(nn.Conv2d(k_in, 2 * F, 1, 1, 0), nn.GroupNorm(2 * F // 32, 2 * F))
        self.conv2 = nn.Sequential(nn.Conv2d(k_in, F, 1, 1, 0), nn.GroupNorm(F // 32, F))
        self._cache = PackedCache()

    def _packed(self):
        c1, c2 = self.conv1[0], self.conv2[0]
        return self._cache.get(
            (c1.weight, c1.bias, c2.weight, c2.bias),
            lambda: ops.pack_gru(c1.weight.detach(), c1.bias.detach(), c2.weight.detach(), c2.bias.detach(),
                                 self.input_channels, self.num_features, self.module == "decoder"),
            weights=(c1.weight, c2.weight))

    def step(self, x, e, h, out=None, phases=ops.PHASE_ALL, ws=None):
        """One timestep on raw (B,C,H,W) tensors.  ``e`` is the encoder skip state (decoder cells) or None.  ``ws``: scratch
        owned by the caller (an engine's buffer); None allocates one for this call."""
        g1, g2 = self.conv1[1], self.conv2[1]
        packed = self._packed()
        with ops.exact_matrix_if(self._cache.wide):       # a weight beyond the f16 pieces' range: exact fp32 MFMA for this cell
            return ops.gru_cell(x, e, h, packed, g1.weight.detach(), g1.bias.detach(), g2.weight.detach(),
                                g2.bias.detach(), self.input_channels, out=out, eps=g1.eps, phases=phases, ws=ws)

    def step_tail(self, x, e, h, stage, conv_out, out=None, phases=ops.PHASE_ALL, ws=None, head_w=None, head_partial0=None):
        """``step`` with the stage conv that consumes the new state (a ``StageLayers`` of kind "conv") fused into the cell's last
        kernel (ops.gru_cell_tail); the caller checked ``ops.gru_cell_tail_applies`` and that neither layer is ``wide``."""
        g1, g2 = self.conv1[1], self.conv2[1]
        return ops.gru_cell_tail(x, e, h, self._packed(), g1.weight.detach(), g1.bias.detach(), g2.weight.detach(), g2.bias.detach(),
                                 self.input_channels, stage._packed(), stage.out_channels, stage.pool, conv_out=conv_out, out=out,
                                 eps=g1.eps, phases=phases, ws=ws, head_w=head_w, head_partial0=head_partial0)

    @torch.no_grad()
    def forward(self, inputs=None, hidden_state=None, seq_len=1):
        """Reference contract (ConvRNN.py:111-194): ``inputs`` (S,B,I,H,W) or None, ``hidden_state`` (B,F,H,W)
        for the encoder or cat(e, d) (B,2F,H,W) for the decoder; returns stacked states (S,B,F,H,W)."""
        F = self.num_features
        if hidden_state is None:
            if inputs is None:
                raise ValueError("CGRU_cell.forward needs inputs or hidden_state")
            hidden_state = torch.zeros(inputs.size(1), F, self.shape[0], self.shape[1], device=inputs.device)
        outs = []
        if self.module == "decoder":
            e = hidden_state[:, :F].contiguous()
            h = hidden_state[:, F:].contiguous()
        else:
            e, h = None, hidden_state.contiguous()
        for index in range(seq_len):
            x = None if inputs is None else inputs[index].contiguous()
            h = self.step(x, e, h)
            outs.append(h)
        return torch.stack(outs)
