"""SWP window planning (urnn_amd.training.plan_windows / window_starts) against goldens produced by the
reference's own ``get_window`` / ``split_iter_index`` (tests/golden/make_fit_golden.py -> fit_host.npz).  CPU only."""
import os
import random

import numpy as np

import urnn_amd.training as tr

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "fit_host.npz"))


def test_window_split_and_plans_match_reference():
    for i, (loc, seq, win) in enumerate(G["split_cases"]):
        assert tr.window_starts(loc, seq, win) == G[f"split_{i}"].tolist()
    for i, (rain_len, event_len, seq, win, allseq, tev, full, wrand) in enumerate(G["plan_cases"]):
        np.random.seed(100 + i)
        random.seed(200 + i)
        loc, seq2, win2, starts = tr.plan_windows(int(rain_len), int(event_len), int(seq), int(win), bool(allseq), bool(tev), bool(full),
                                                  bool(wrand))
        assert [loc, seq2, win2] + starts == G[f"plan_{i}"].tolist(), i
