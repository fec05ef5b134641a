#!/usr/bin/env python
"""Goldens for the host side of the training loop, produced by calling the REFERENCE's own functions in this
container: learning-rate schedules driven the way main.py drives them (one scheduler.step() per epoch), window planning
(`get_window` with seeded numpy / random generators), and SaveBestModel's save decisions and file names on a seeded loss series.

Needs /root/reference; never shipped to the GPU box.  Usage: python tests/golden/make_fit_golden.py
"""
import os
import random
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402,F401  (puts the reference on sys.path, applies the .cuda shim)

for missing in ("wandb", "git", "openpyxl", "cv2", "seaborn"):      # imported at module level by main.py / its helpers, unused here
    try:
        __import__(missing)
    except ImportError:
        sys.modules[missing] = types.ModuleType(missing)
import main as ref_main  # noqa: E402
from src.lib.model.earlystopping import SaveBestModel  # noqa: E402


def lr_series(schedule_name, lr, warm_up_iter, epochs, lr_min, factor, patience, losses):
    lin = torch.nn.Linear(2, 2)
    opt = torch.optim.Adam(lin.parameters(), lr=lr)
    if schedule_name == "ReduceLROnPlateau":
        # main.py:249-256 passes verbose=True, which this container's torch (2.10) no longer accepts: same call without it
        sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, factor=factor, patience=patience, cooldown=0, min_lr=lr_min)
    else:
        sched = ref_main.lr_schedule(opt, schedule_name, lr, warm_up_iter, epochs, lr_min=lr_min, factor=factor, patience=patience)
    args = types.SimpleNamespace(schedule_name=schedule_name)
    out = []
    for loss in losses:
        out.append(opt.param_groups[0]["lr"])                 # the lr the epoch trains with
        opt.step()
        ref_main.scheduler_update(args, sched, {"loss": float(loss)})
    return np.asarray(out, np.float64)


def main():
    out = {}
    rs = np.random.RandomState(11)
    losses = np.abs(np.cumsum(rs.normal(-0.01, 0.03, 80)) + 1.0)          # drifts down with plateaus and bumps
    out["losses"] = losses
    out["lr_cosine"] = lr_series("WarmUpCosineAnneal", 0.01, 10, 60, 1e-4, 0.9, 10, losses)
    out["lr_cosine_short"] = lr_series("WarmUpCosineAnneal", 0.003, 0, 5, 1e-5, 0.9, 10, losses[:12])
    out["lr_plateau"] = lr_series("ReduceLROnPlateau", 0.01, 10, 60, 1e-4, 0.9, 3, losses)
    out["lr_plateau_floor"] = lr_series("ReduceLROnPlateau", 2e-4, 10, 60, 1e-4, 0.5, 0, losses[:20])

    cases = [(0, 28, 360), (0, 4, 10), (5, 7, 30), (3, 12, 36), (0, 36, 36), (2, 5, 5)]
    out["split_cases"] = np.asarray(cases)
    for i, (loc, seq, win) in enumerate(cases):
        out[f"split_{i}"] = np.asarray(ref_main.split_iter_index(loc, seq, win))

    # get_window with seeded generators: (rain_len, event_len, seq_num, window_size, all_seq_train, train_event, full_window, wind_random)
    plans = [(360, 360, 28, 360, 0, 1, 0, 1), (360, 200, 28, 120, 0, 1, 0, 1), (360, 200, 28, 120, 0, 0, 0, 0),
             (72, 72, 100, 360, 1, 1, 0, 1), (36, 36, 12, 36, 0, 1, 1, 0), (360, 300, 7, 50, 0, 1, 0, 1)]
    out["plan_cases"] = np.asarray(plans)
    for i, (rain_len, event_len, seq, win, allseq, tev, full, wrand) in enumerate(plans):
        np.random.seed(100 + i)
        random.seed(200 + i)
        args = types.SimpleNamespace(window_size=win, seq_num=seq, all_seq_train=bool(allseq), train_event=bool(tev),
                                     full_window_size=bool(full), wind_random=bool(wrand))
        inputs = {"rainfall": torch.zeros(1, rain_len, 1, 1, 1)}
        label = torch.zeros(1, event_len, 2, 2)
        loc, _, idx = ref_main.get_window(args, inputs, label)
        out[f"plan_{i}"] = np.asarray([loc, args.seq_num, args.window_size] + list(idx))

    # SaveBestModel on the loss series: which epochs write a checkpoint, and under which name
    lin = torch.nn.Linear(2, 2)
    opt = torch.optim.Adam(lin.parameters(), lr=0.01)
    saver = SaveBestModel(verbose=False)
    with tempfile.TemporaryDirectory() as d:
        saved = []
        for epoch, loss in enumerate(losses[:40]):
            before = set(os.listdir(d))
            saver(float(loss), lin, opt, epoch, d)
            new = sorted(set(os.listdir(d)) - before)
            saved.append(new[0] if new else "")
        info = torch.load(os.path.join(d, [s for s in saved if s][-1]), map_location="cpu", weights_only=False)
        out["ckpt_keys"] = np.asarray(sorted(info.keys()))
    out["saved_names"] = np.asarray(saved)
    path = os.path.join(HERE, "fit_host.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", sum(1 for s in saved if s), "checkpoints in 40 epochs")


if __name__ == "__main__":
    main()
