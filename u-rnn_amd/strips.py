"""Spatial-strip rollout: ONE event's grid split over the ranks in horizontal strips, for single-event
latency once event parallelism is exhausted.  All convs are 1x1 and the pool / transposed-conv blocks 2x2, so a strip whose
height is a multiple of 4 needs no halo; the only exchange is the GroupNorm / LayerNorm statistics -- 12 (cells) + 3 (head)
all-reduces of a few doubles per timestep (RCCL with the nccl backend; latency-bound on xGMI).  Eager launches: the exchanges
sit between kernels of one cell, so a captured timestep would need graph-capturable collectives."""
import torch

from . import ops
from .dataset import event_to_device
from .general import initialize_states


def strip_rows(H, rank, world):
    """Rows [r0, r1) of rank ``rank``: multiples of 4 (two 2x2 pool levels), as even as H/4 allows."""
    if H % 4 != 0:
        raise ValueError(f"strip mode needs a grid height that is a multiple of 4 (got {H})")
    units = H // 4
    if world > units:
        raise ValueError(f"{world} strips need at least {4 * world} rows (got {H})")
    base, rem = divmod(units, world)
    r0 = 4 * (rank * base + min(rank, rem))
    return r0, r0 + 4 * (base + (1 if rank < rem else 0))


class StripRollout:
    """``sr = StripRollout(net, H, W, nums, rain_max, cumsum_max, rank, world, group)``; ``sr.load_event(event)``;
    ``strip = sr.run(T)`` -> (T, B, rows, W) masked depths of this rank's rows; ``sr.gather(strip)`` -> (T, B, H, W) on every
    rank.  ``net`` is the full-grid network (every rank holds all weights; LayerNorm affines are sliced per strip)."""

    def __init__(self, net, H, W, nums, rain_max, cumsum_max, rank=0, world=1, group=None):
        self.net, self.H, self.W, self.nums = net, H, W, int(nums)
        self.rain_max, self.cumsum_max = float(rain_max), float(cumsum_max)
        self.rank, self.world, self.group = int(rank), int(world), group
        self.r0, self.r1 = strip_rows(H, rank, world)
        self.device = next(net.parameters()).device
        fp = net.head.flat_params()
        self.ln_w = fp["ln_w"][..., self.r0:self.r1, :].contiguous()
        self.ln_b = fp["ln_b"][..., self.r0:self.r1, :].contiguous()
        self.ev = None
        self.states = None
        self.exchanges = 0

    def _exchange(self, sums):
        self.exchanges += 1
        if self.world > 1:
            import torch.distributed as dist
            if dist.get_backend(self.group) == "gloo" and sums.is_cuda:       # CPU-staged dry runs (several ranks on one GPU)
                host = sums.cpu()
                dist.all_reduce(host, group=self.group)
                sums.copy_(host)
            else:
                dist.all_reduce(sums, group=self.group)

    def load_event(self, event):
        ev = event if "rain" in event else event_to_device(event, self.device)
        s = dict(ev)
        for k in ("dem", "imperv", "manhole"):
            s[k] = ev[k][:, self.r0:self.r1].contiguous()
        if ev["rain"].dim() == 4:
            for k in ("rain", "cumsum"):
                s[k] = ev[k][:, :, self.r0:self.r1].contiguous()
        self.ev = s
        full = initialize_states(self.device, self.H, self.W)
        B = ev["B"]
        self.states = []
        for st in full:                                            # strip of each state at its own resolution
            f = self.H // st.shape[-2]
            self.states.append(st[..., self.r0 // f:self.r1 // f, :].to(self.device).repeat(B, 1, 1, 1).contiguous())

    def step(self, t):
        net, ev = self.net, self.ev
        enc, dec, head = net.encoder, net.decoder, net.head
        e1, e2, e3, d1, d2, d3 = self.states
        gp = self.H * self.W

        def cell(mod, x, e, h, level):
            g1, g2 = mod.conv1[1], mod.conv2[1]
            packed = mod._packed()
            with ops.exact_matrix_if(mod._cache.wide):
                return ops.gru_cell_strip(x, e, h, packed, g1.weight.detach(), g1.bias.detach(), g2.weight.detach(), g2.bias.detach(),
                                          mod.input_channels, gp // (level * level), self._exchange, eps=g1.eps)

        x_in = ops.preprocess(ev["rain"], ev["cumsum"], ev["dem"], ev["imperv"], ev["manhole"], ev["dem_min"], ev["dem_max"], int(t),
                              self.nums, self.rain_max, self.cumsum_max)
        e1 = cell(enc.rnn1, enc.stage1(x_in), None, e1, 1)
        e2 = cell(enc.rnn2, enc.stage2(e1), None, e2, 2)
        e3 = cell(enc.rnn3, enc.stage3(e2), None, e3, 4)
        d1 = cell(dec.rnn3, None, e3, d1, 4)
        d2 = cell(dec.rnn2, dec.stage3(d1), e2, d2, 2)
        d3 = cell(dec.rnn1, dec.stage2(d2), e1, d3, 1)
        feat = dec.stage1(d3)
        fp = head.flat_params()
        masked, cls, _ = ops.head_strip(feat, fp["conv_w"], self.ln_w, self.ln_b, head.cls_preds.conv.weight.detach().reshape(-1),
                                        head.cls_preds.conv.bias.detach(), head.reg_preds.conv.weight.detach().reshape(-1),
                                        head.reg_preds.conv.bias.detach(), head.cls_thred, gp, self._exchange, eps=head.stems.ln.eps)
        self.states = [e1, e2, e3, d1, d2, d3]
        return masked, cls

    @torch.no_grad()
    def run(self, T, t0=0):
        out = []
        for t in range(t0, t0 + T):
            out.append(self.step(t)[0])
        return torch.stack(out)

    def gather(self, strip):
        """(T,B,rows,W) per rank -> (T,B,H,W) on every rank (strips padded to the tallest one for the all-gather)."""
        if self.world == 1:
            return strip
        import torch.distributed as dist
        rows = [strip_rows(self.H, r, self.world) for r in range(self.world)]
        tall = max(b - a for a, b in rows)
        pad = torch.zeros(strip.shape[:2] + (tall, self.W), dtype=strip.dtype, device=strip.device)
        pad[:, :, :strip.shape[2]] = strip
        staged = dist.get_backend(self.group) == "gloo" and pad.is_cuda
        src = pad.cpu() if staged else pad
        bucket = [torch.empty_like(src) for _ in range(self.world)]
        dist.all_gather(bucket, src, group=self.group)
        return torch.cat([bucket[r][:, :, :b - a] for r, (a, b) in enumerate(rows)], dim=2).to(strip.device)
