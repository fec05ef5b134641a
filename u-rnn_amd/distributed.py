"""Event-parallel multi-GPU helpers (one process per GPU, torch.distributed: RCCL on GPUs, gloo on CPU tests).

Inference shards naturally over independent rainfall events and needs NO data-path collective: the reference
uses ``DistributedSampler(shuffle=False)`` over events and never gathers (test.py:741-746).  The
helpers here reproduce that partition and add the two control-plane reductions a benchmark/driver needs (max of the
per-rank wall time, optional gather of per-event results to rank 0)."""
import math
import os

import torch


def env_ranks():
    """(local_rank, rank, world_size) from the torchrun environment; (0, 0, 1) when not launched distributed
    (the reference reads the same variables: general.py:25-27)."""
    return (int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)))


def shard_events(num_events, rank, world_size):
    """Indices of the events rank ``rank`` processes: ``DistributedSampler(dataset, shuffle=False, drop_last=False)``
    semantics -- the index list is padded by wrapping around to a multiple of world_size, then strided."""
    if num_events <= 0:
        return []
    per_rank = math.ceil(num_events / world_size)
    total = per_rank * world_size
    indices = list(range(num_events))
    while len(indices) < total:
        indices += indices[: total - len(indices)]
    return indices[rank:total:world_size]


def max_over_ranks(value, device=None):
    """Max of a python float over all ranks (bench timing contract); identity when not distributed."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allreduce_mean_(flat, group=None, bucket_floats=1 << 24):
    """DDP's gradient exchange (main.py:384-387) on the trainer's flat gradient buffer: in-place mean over the ranks, in
    buckets of ``bucket_floats`` (64 MiB) so that the xGMI ring pipelines and a following optimizer kernel can start on the
    first buckets.  With the nccl backend this is RCCL on the device buffer; gloo (CPU dry runs) stages through the host."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return flat
    world = dist.get_world_size(group)
    staged = dist.get_backend(group) == "gloo" and flat.is_cuda
    buf = flat.cpu() if staged else flat
    for lo in range(0, buf.numel(), bucket_floats):
        piece = buf[lo:lo + bucket_floats]
        dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=group)
        piece.div_(world)
    if staged:
        flat.copy_(buf)
    return flat


class OverlappedGradientMean:
    """DDP's gradient mean (main.py:384-387: DistributedDataParallel overlaps its bucketed all-reduce with the rest of the
    backward pass) on the trainer's flat gradient buffer.  The buffer is cut where the backward pass finishes it: the TAIL
    ``flat[split:]`` -- the head, whose LayerNorm affines are 99 % of the bytes (160 of 161.7 MB at 500x500) -- is final as soon
    as the head backward of the window's first timestep has run, so ``start_tail()`` launches its all-reduce there
    (``async_op``: RCCL runs on its own stream, over xGMI) while the decoder / encoder backward of that timestep still computes;
    ``finish()`` reduces the small remainder, joins, and leaves the MEAN in place.  With the nccl backend the reduction is
    ``ReduceOp.AVG`` (no scaling pass); gloo (CPU dry runs, several ranks on one GPU) stages through the host and is
    synchronous -- same call order, same result."""

    def __init__(self, flat, split, group=None, bucket_floats=1 << 24, force=False):
        import torch.distributed as dist
        self.flat, self.split, self.group, self.bucket = flat, int(split), group, int(bucket_floats)
        # force: run the collectives even in a world of one (tests drive the RCCL call path and its stream ordering on a 1-GPU box)
        self.active = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or bool(force))
        self.world = dist.get_world_size(group) if self.active else 1
        self.nccl = self.active and dist.get_backend(group) == "nccl"
        self._pending = []

    def _reduce(self, piece, asynchronous):
        import torch.distributed as dist
        if self.nccl:
            work = dist.all_reduce(piece, op=dist.ReduceOp.AVG, group=self.group, async_op=asynchronous)
            if asynchronous:
                self._pending.append(work)
            return
        staged = piece.is_cuda                       # gloo: reduce a host copy, scale, copy back
        buf = piece.cpu() if staged else piece
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
        buf.div_(self.world)
        if staged:
            piece.copy_(buf)

    def start_tail(self):
        if not self.active:
            return
        tail = self.flat[self.split:]
        for lo in range(0, tail.numel(), self.bucket):
            self._reduce(tail[lo:lo + self.bucket], asynchronous=True)

    def finish(self):
        if not self.active:
            return self.flat
        if self.split > 0:
            self._reduce(self.flat[:self.split], asynchronous=False)
        for work in self._pending:
            work.wait()                               # the current stream waits for RCCL's; no host synchronisation
        self._pending = []
        return self.flat


def gather_event_results(local_results, num_events, device=None):
    """Gather per-event tensors (same shape on every rank) onto every rank in GLOBAL event order.

    ``local_results`` is the list produced by this rank for ``shard_events(num_events, rank, world)`` (padding
    duplicates included).  Returns a list of ``num_events`` tensors (on CPU); duplicates from padding are dropped."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [r.cpu() for r in local_results[:num_events]]
    world, rank = dist.get_world_size(), dist.get_rank()
    stacked = torch.stack([r.to(device) if device is not None else r for r in local_results])
    bucket = [torch.empty_like(stacked) for _ in range(world)]
    dist.all_gather(bucket, stacked)
    out = [None] * num_events
    for r in range(world):
        for k, idx in enumerate(shard_events(num_events, r, world)):
            if out[idx] is None:
                out[idx] = bucket[r][k].cpu()
    return out
