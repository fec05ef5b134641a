"""Host side of the training loop around ``training.Trainer`` (reference ``main.py``): window planning per
sample, learning-rate schedules, the epoch loop with the reference's sample order, best-loss checkpoints in the reference's
file format, resume.  Pure Python / numpy -- every timestep of work goes through the HIP path inside ``Trainer``."""
import math
import os
import random

import numpy as np
import torch

from .training import window_starts


# ---- window planning (main.py:106-178, 415-443) ---------------------------------------------------------------------------
def correction_seq_num(seq_num, window_size, full_window_size=False):
    """main.py:106-118."""
    return window_size if full_window_size else min(seq_num, window_size)


def correction_window_size(window_size, rain_len, event_len, all_seq_train=False, train_event=False):
    """main.py:121-136."""
    sample_length = event_len if train_event else rain_len
    return sample_length if all_seq_train else min(window_size, sample_length)


def get_start_loc(rain_len, window_size, event_len, train_event=False):
    """main.py:139-159: a random start when the sample is longer than the window (numpy's global generator, as there)."""
    sample_length = event_len if train_event else rain_len
    loc = 0
    if sample_length - window_size > 0:
        loc = int(np.random.randint(0, sample_length - window_size, size=1, dtype=int)[0])
    return loc


def plan_windows(rain_len, event_len, seq_num, window_size, all_seq_train=False, train_event=True, full_window_size=False,
                 wind_random=True):
    """``get_window`` (main.py:415-443).  Returns (loc, seq_num, window_size, window start indices in processing order);
    like the reference, the corrected seq_num / window_size replace the configured ones for the rest of the run."""
    window_size = correction_window_size(window_size, rain_len, event_len, all_seq_train, train_event)
    loc = get_start_loc(rain_len, window_size, event_len, train_event)
    seq_num = correction_seq_num(seq_num, window_size, full_window_size)
    starts = window_starts(loc, seq_num, window_size)
    if wind_random:
        random.shuffle(starts)
    return loc, seq_num, window_size, starts


# ---- learning-rate schedules (main.py:181-277) ---------------------------------------------------------------------------
def warmup_cosine_factor(cur_iter, warm_up_iter, T_max, lr_max, lr_min):
    """Multiplier of ``WarmUpCosineAnneal`` (main.py:202-231): lr(epoch) = lr_max * factor(epoch)."""
    min_ratio = lr_min / lr_max
    if cur_iter < warm_up_iter:
        return float(cur_iter) / float(max(1, warm_up_iter))
    if cur_iter > T_max:
        return min_ratio
    progress = (cur_iter - warm_up_iter) / max(1, T_max - warm_up_iter)
    return min_ratio + 0.5 * (1.0 - min_ratio) * (1.0 + math.cos(math.pi * progress))


def warmup_cosine_v2_factor(cur_iter, warm_up_iter, T_max, lr_max, lr_min):
    """Multiplier of ``WarmUpCosineAnneal_v2`` (main.py:181-199; written for a base lr of 0.1)."""
    if cur_iter < warm_up_iter:
        return cur_iter / warm_up_iter * lr_max / 0.1
    return lr_min + 0.5 * (lr_max - lr_min) * (1 + math.cos((cur_iter - warm_up_iter) / (T_max - warm_up_iter) * math.pi)) / 0.1


class LambdaSchedule:
    """torch.optim.lr_scheduler.LambdaLR as the reference drives it: lr = base_lr * factor(k), k = 0 at construction and
    +1 per ``step()`` (one per epoch, main.py:837-854)."""

    def __init__(self, base_lr, factor):
        self.base_lr, self.factor, self.k = float(base_lr), factor, 0

    @property
    def lr(self):
        return self.base_lr * self.factor(self.k)

    def step(self, metric=None):
        self.k += 1
        return self.lr


class PlateauSchedule:
    """torch.optim.lr_scheduler.ReduceLROnPlateau(mode="min", threshold=1e-4 rel, cooldown=0) with the reference's factor /
    patience / min_lr (main.py:251-259)."""

    def __init__(self, base_lr, factor=0.9, patience=10, min_lr=1e-4, threshold=1e-4, eps=1e-8):
        self.lr, self.factor, self.patience, self.min_lr = float(base_lr), float(factor), int(patience), float(min_lr)
        self.threshold, self.eps, self.best, self.bad = float(threshold), float(eps), math.inf, 0

    def step(self, metric):
        metric = float(metric)
        if metric < self.best * (1.0 - self.threshold):
            self.best, self.bad = metric, 0
        else:
            self.bad += 1
        if self.bad > self.patience:
            new_lr = max(self.lr * self.factor, self.min_lr)
            if self.lr - new_lr > self.eps:
                self.lr = new_lr
            self.bad = 0
        return self.lr


def lr_schedule(schedule_name, lr, warm_up_iter, epochs, lr_min=1e-4, factor=0.9, patience=10):
    """``lr_schedule`` (main.py:233-277)."""
    if schedule_name == "ReduceLROnPlateau":
        return PlateauSchedule(lr, factor=factor, patience=patience, min_lr=lr_min)
    if schedule_name == "WarmUpCosineAnneal":
        return LambdaSchedule(lr, lambda k: warmup_cosine_factor(k, warm_up_iter, epochs, lr, lr_min))
    if schedule_name == "WarmUpCosineAnneal_v2":
        return LambdaSchedule(lr, lambda k: warmup_cosine_v2_factor(k, warm_up_iter, epochs, lr, lr_min))
    raise ValueError(f"unknown schedule {schedule_name!r}")


# ---- sample order (torch DistributedSampler(shuffle=True), main.py:958-961 + set_epoch main.py:885-887) ---------------------
def epoch_order(n, epoch, rank=0, world_size=1, seed=0, drop_last_batch=1):
    """Indices of the samples this rank trains on in ``epoch``: the permutation torch's DistributedSampler draws
    (generator seeded with seed + epoch, padded by wrap-around to a multiple of world_size, strided by rank).  With
    ``batch_size`` > 1 the loader drops the last incomplete batch (drop_last=True, main.py:977): pass it as drop_last_batch."""
    g = torch.Generator()
    g.manual_seed(int(seed) + int(epoch))
    idx = torch.randperm(n, generator=g).tolist()
    total = -(-n // world_size) * world_size
    pad = total - len(idx)
    if pad > 0:
        idx += (idx * (-(-pad // len(idx))))[:pad]
    idx = idx[rank:total:world_size]
    usable = len(idx) // drop_last_batch * drop_last_batch
    return idx[:usable]


# ---- checkpoints (earlystopping.py:5-74, main.py:280-389) -----------------------------------------------------------------
class BestCheckpoint:
    """``SaveBestModel``: writes ``checkpoint_{epoch}_{loss:.9f}.pth.tar`` = {"epoch", "state_dict", "optimizer"} whenever the
    epoch loss is not worse than the best so far.  The optimizer entry is a torch.optim.Adam state dict, so the reference can
    resume from our files and we from its."""

    def __init__(self, patience=7, verbose=False):
        self.patience, self.verbose = patience, verbose
        self.counter, self.best_score, self.early_stop, self.val_loss_min = 0, None, False, np.inf

    def __call__(self, val_loss, trainer, epoch, save_path):
        score = -val_loss
        if self.best_score is None or score >= self.best_score:
            self.best_score = score
            self.counter = 0
            return self.save_checkpoint(val_loss, trainer, epoch, save_path)
        self.counter += 1
        if self.counter >= self.patience:
            self.early_stop = True
        return None

    def save_checkpoint(self, val_loss, trainer, epoch, save_path):
        if self.verbose:
            print(f"Validation loss decreased ({self.val_loss_min:.9f} --> {val_loss:.9f}).  Saving model ...")
        os.makedirs(save_path, exist_ok=True)
        path = os.path.join(save_path, "checkpoint_{}_{:.9f}.pth.tar".format(epoch, val_loss))
        torch.save({"epoch": epoch, "state_dict": trainer.state_dict(), "optimizer": trainer.optimizer_state_dict()}, path)
        self.val_loss_min = val_loss
        return path


def latest_checkpoint(save_dir):
    """The file ``load_model`` resumes from (main.py:335-341): highest epoch number in ``checkpoint_{epoch}_{loss}.pth.tar``."""
    names = [n for n in os.listdir(save_dir) if n.startswith("checkpoint_")]
    if not names:
        return None
    return os.path.join(save_dir, sorted(names, key=lambda x: int(x.replace("checkpoint_", "").split("_")[0]))[-1])


def resume(trainer, save_dir):
    """Load weights + Adam state of the newest checkpoint (``module.`` prefixes stripped); returns the epoch to continue at."""
    path = latest_checkpoint(save_dir)
    if path is None:
        return 0
    info = torch.load(path, map_location="cpu", weights_only=False)
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in info["state_dict"].items()}
    trainer.load_state_dict(sd)
    if "optimizer" in info:
        trainer.load_optimizer_state_dict(info["optimizer"])
    return int(info.get("epoch", -1)) + 1


# ---- epoch loop (main.py:771-918) ------------------------------------------------------------------------------------
LOSS_KEYS = ("loss", "loss_reg", "loss_reg_label", "loss_reg_pred", "loss_cls")


def train_sample(trainer, event, label, seq_num, window_size, prewarming=False, all_seq_train=False, train_event=True,
                 full_window_size=False, wind_random=True):
    """``model_forward`` (main.py:695-768) for one sample: plan the windows, train them in (shuffled) order, carrying the
    states from window to window in fast mode.  label (B,T,H,W) already normalised.  Returns ({loss key: per-window list},
    corrected seq_num, corrected window_size)."""
    rain_len = int(event["T"] if "T" in event else event["rainfall"].shape[1])
    event_len = int(label.shape[1])
    _, seq_num, window_size, starts = plan_windows(rain_len, event_len, seq_num, window_size, all_seq_train, train_event,
                                                   full_window_size, wind_random)
    losses, _ = trainer.train_event(event, label, seq_num, prewarming=prewarming, starts=starts)
    comps = torch.stack(losses).cpu().numpy()          # one host copy per sample
    return {k: comps[:, i].tolist() for i, k in enumerate(LOSS_KEYS)}, seq_num, window_size


def fit(trainer, dataset, epochs, lr, flood_max, seq_num=28, window_size=360, schedule_name="WarmUpCosineAnneal", warm_up_iter=10,
        lr_min=1e-4, factor=0.9, patience=10, prewarming=False, all_seq_train=False, train_event=True, full_window_size=False,
        wind_random=True, save_dir=None, start_epoch=0, rank=0, world_size=1, seed=0, log=print):
    """``train`` (main.py:857-918): per epoch the rank's share of the samples in DistributedSampler order, one
    ``train_sample`` each; the epoch's record is the average over the LAST sample's windows (main.py:805-806 averages the
    ``iter_loss`` the last ``model_forward`` returned); the schedule steps once per epoch; rank 0 keeps the best
    checkpoints.  As in the reference a resumed run (start_epoch > 0) restarts the schedule at iteration 0.
    ``dataset[i]`` -> (event dict, label (T,H,W) or (B,T,H,W) in mm, name).  Returns the per-epoch records."""
    sched = lr_schedule(schedule_name, lr, warm_up_iter, epochs, lr_min=lr_min, factor=factor, patience=patience)
    best = BestCheckpoint(verbose=True) if rank == 0 and save_dir else None
    history = []
    for epoch in range(start_epoch, epochs):
        trainer.set_lr(sched.lr)
        rec_losses = None
        for i in epoch_order(len(dataset), epoch, rank, world_size, seed):
            event, label, _ = dataset.batched(i) if hasattr(dataset, "batched") else dataset[i]
            label = torch.as_tensor(np.asarray(label), dtype=torch.float32)
            label = label.reshape((-1,) + tuple(label.shape[-3:])) if label.dim() != 3 else label[None]   # (B,T,H,W)
            label = (label - 0.0) / (float(flood_max) - 0.0)                     # MinMaxScaler(labels, flood_max, 0), main.py:797
            rec_losses, seq_num, window_size = train_sample(trainer, event, label, seq_num, window_size, prewarming, all_seq_train,
                                                            train_event, full_window_size, wind_random)
        info = {"lr": trainer.lr}
        if rec_losses is not None:
            info.update({k: float(np.average(v)) for k, v in rec_losses.items()})
        sched.step(info.get("loss"))
        history.append(info)
        if log is not None and rank == 0:
            log(f"[{epoch + 1}/{epochs}] " + " | ".join(f"{k}:{v:.9f}" for k, v in info.items()))
        if best is not None and "loss" in info:
            best(info["loss"], trainer, epoch, save_dir)
    return history
