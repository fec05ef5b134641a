"""Host side of the training loop (urnn_amd.fit) against goldens produced by the reference's own functions
(tests/golden/make_fit_golden.py -> fit_host.npz) and against torch's sampler / optimizer objects.  CPU only."""
import os
import random
import tempfile

import numpy as np
import pytest
import torch

import urnn_amd.fit as fit

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "fit_host.npz"))


def _series(sched, losses):
    out = []
    for loss in losses:
        out.append(sched.lr)
        sched.step(float(loss))
    return np.asarray(out)


@pytest.mark.parametrize("key,args", [
    ("lr_cosine", ("WarmUpCosineAnneal", 0.01, 10, 60, 1e-4, 0.9, 10)),
    ("lr_cosine_short", ("WarmUpCosineAnneal", 0.003, 0, 5, 1e-5, 0.9, 10)),
    ("lr_plateau", ("ReduceLROnPlateau", 0.01, 10, 60, 1e-4, 0.9, 3)),
    ("lr_plateau_floor", ("ReduceLROnPlateau", 2e-4, 10, 60, 1e-4, 0.5, 0)),
])
def test_lr_schedules_match_reference(key, args):
    name, lr, warm, epochs, lr_min, factor, patience = args
    want = G[key]
    got = _series(fit.lr_schedule(name, lr, warm, epochs, lr_min=lr_min, factor=factor, patience=patience), G["losses"][:len(want)])
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=0)


def test_window_split_and_plans_match_reference():
    for i, (loc, seq, win) in enumerate(G["split_cases"]):
        assert fit.window_starts(loc, seq, win) == G[f"split_{i}"].tolist()
    for i, (rain_len, event_len, seq, win, allseq, tev, full, wrand) in enumerate(G["plan_cases"]):
        np.random.seed(100 + i)
        random.seed(200 + i)
        loc, seq2, win2, starts = fit.plan_windows(int(rain_len), int(event_len), int(seq), int(win), bool(allseq), bool(tev), bool(full),
                                                   bool(wrand))
        assert [loc, seq2, win2] + starts == G[f"plan_{i}"].tolist(), i


def test_epoch_order_is_the_distributed_samplers():
    from torch.utils.data.distributed import DistributedSampler
    data = list(range(11))
    for world in (1, 2, 4):
        for rank in range(world):
            s = DistributedSampler(data, num_replicas=world, rank=rank, shuffle=True, seed=3)
            for epoch in (0, 1, 7):
                s.set_epoch(epoch)
                assert fit.epoch_order(len(data), epoch, rank, world, seed=3) == list(iter(s))
    assert len(fit.epoch_order(11, 0, 0, 1, drop_last_batch=4)) == 8


class _FakeTrainer:
    def __init__(self):
        self.lin = torch.nn.Linear(2, 2)

    def state_dict(self):
        return self.lin.state_dict()

    def optimizer_state_dict(self):
        return {"state": {}, "param_groups": [{"params": [0, 1]}]}


def test_best_checkpoint_saves_what_the_reference_saves():
    saver, tr = fit.BestCheckpoint(), _FakeTrainer()
    with tempfile.TemporaryDirectory() as d:
        for epoch, (loss, want) in enumerate(zip(G["losses"][:40], G["saved_names"])):
            path = saver(float(loss), tr, epoch, d)
            assert (os.path.basename(path) if path else "") == str(want), epoch
        info = torch.load(fit.latest_checkpoint(d), map_location="cpu", weights_only=False)
        assert sorted(info.keys()) == G["ckpt_keys"].tolist()
        last = max(i for i, n in enumerate(G["saved_names"]) if n)
        assert info["epoch"] == last


def _small_trainer():
    from urnn_amd.networks.net_params import get_network_params
    from urnn_amd.networks.model import ED
    from urnn_amd.net_config import load_net_config
    import urnn_amd.weights as uw
    from urnn_amd.training import Trainer
    H = W = 16
    C = 9
    ep, dp = get_network_params(False, H, W, C, load_net_config())
    net = ED(False, ep, dp, 0.5, False, input_height=H, input_width=W)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in uw.make_state_dict(H, W, C, seed=4).items()})
    return net, Trainer(net, H, W, 3, 6.0, 250.0, lr=2e-3, grad_clip=1.0)


def test_checkpoint_round_trip_with_torch_adam():
    """Our optimizer entry loads into torch.optim.Adam (what the reference resumes with, main.py:357) and torch's loads into ours."""
    net, tr = _small_trainer()
    tr.step_count = 5
    tr.m.copy_(torch.linspace(-1, 1, tr.m.numel()))
    tr.v.copy_(torch.linspace(0, 2, tr.v.numel()))
    with tempfile.TemporaryDirectory() as d:
        fit.BestCheckpoint()(0.25, tr, 3, d)
        info = torch.load(fit.latest_checkpoint(d), map_location="cpu", weights_only=False)
        ref_opt = torch.optim.Adam(net.parameters(), lr=1.0)
        ref_opt.load_state_dict(info["optimizer"])
        assert ref_opt.param_groups[0]["lr"] == pytest.approx(2e-3)
        p0 = next(net.parameters())
        off, k, shape = tr.views[tr.names[0]]
        torch.testing.assert_close(ref_opt.state[p0]["exp_avg"], tr.m[off:off + k].view(shape).cpu())
        assert float(ref_opt.state[p0]["step"]) == 5.0
        # and back: a fresh trainer resumes from the file
        net2, tr2 = _small_trainer()
        with torch.no_grad():
            tr2.flat.mul_(0.5)
        assert fit.resume(tr2, d) == 4
        torch.testing.assert_close(tr2.flat, tr.flat)
        torch.testing.assert_close(tr2.m, tr.m)
        torch.testing.assert_close(tr2.v, tr.v)
        assert tr2.step_count == 5 and tr2.lr == pytest.approx(2e-3)
        # a torch-written optimizer state (after one real step) loads too
        for p in net.parameters():
            p.grad = torch.ones_like(p)
        ref_opt.step()
        tr2.load_optimizer_state_dict(ref_opt.state_dict())
        assert tr2.step_count == 6
    with pytest.raises(KeyError):
        tr.load_state_dict({"nope": torch.zeros(1)})
