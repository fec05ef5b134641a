"""world_size-2 gloo test (CPU) of the N>1 path: event sharding without a data-path collective, the max-over-ranks
timing reduction and the optional result gather -- the same code bench.py / a multi-GPU driver run over RCCL."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, num_events, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from urnn_amd.distributed import OverlappedGradientMean, allreduce_mean_, env_ranks, gather_event_results, max_over_ranks, shard_events
    dist.init_process_group("gloo", rank=rank, world_size=world)
    assert env_ranks() == (rank, rank, world)
    mine = shard_events(num_events, rank, world)
    # stand-in for a rollout: a (T,H,W) result that encodes the event index
    results = [torch.full((3, 4, 5), float(idx)) for idx in mine]
    slowest = max_over_ranks(1.0 + rank)
    gathered = gather_event_results(results, num_events)
    # the trainer's gradient exchange: bucketed in-place mean of a flat buffer (DDP semantics)
    flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    allreduce_mean_(flat, bucket_floats=256)
    mean_ok = torch.equal(flat, torch.arange(1000, dtype=torch.float32) * (sum(range(1, world + 1)) / world))
    # the same mean in two parts, as the trainer runs it: the tail (head gradients) first, while the front is still being written
    flat2 = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    flat2[:300] = -1.0                                   # not final yet when the tail's all-reduce starts
    red = OverlappedGradientMean(flat2, 300, bucket_floats=256)
    red.start_tail()
    flat2[:300] = torch.arange(300, dtype=torch.float32) * (rank + 1)
    red.finish()
    mean_ok = mean_ok and torch.equal(flat2, flat)
    dist.barrier()
    ok = mean_ok and slowest == float(world) and len(gathered) == num_events and all(float(g[0, 0, 0]) == i for i, g in enumerate(gathered))
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), np.array([int(ok)] + mine))
    dist.destroy_process_group()


def test_two_rank_event_sharding(tmp_path):
    world, num_events = 2, 5
    port = _free_port()
    mp.spawn(_worker, args=(world, port, num_events, str(tmp_path)), nprocs=world, join=True)
    seen = []
    for r in range(world):
        data = np.load(os.path.join(str(tmp_path), f"rank{r}.npy"))
        assert data[0] == 1, f"rank {r} failed its checks"
        seen += list(data[1:])
    assert set(seen) == set(range(num_events))          # every event processed
    assert len(seen) == 6                               # padded to a multiple of world, as DistributedSampler does


def test_bench_self_launch_builds_torchrun_command(monkeypatch):
    """`python bench.py --gpus N` (no launcher environment) re-runs itself as N ranks, one per GPU, like the reference's
    `torchrun --nproc_per_node=N` (README.md:415-422); with fewer GPUs than ranks it refuses instead of printing n_gpus: 1."""
    import subprocess
    import sys

    import pytest

    import bench
    calls = []
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: calls.append((cmd, env)) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"])
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as exc:
        bench.main()
    assert exc.value.code == 0 and len(calls) == 1
    cmd, env = calls[0]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    assert os.path.basename(cmd[cmd.index("--master-port") + 2]) == "bench.py" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    with pytest.raises(SystemExit) as exc:
        bench.main()
    assert "needs 4 GPUs" in str(exc.value.code) and len(calls) == 1
