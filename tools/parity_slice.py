"""A/B of arithmetic variants on the mid-event slice (tests/slice_parity.py): run once per library build, e.g.
    URNN_LIB=u-rnn_amd/liburnn_hip_act0.so python tools/parity_slice.py --cache /tmp/slice.npz
The oracle / torch-fp32 outputs are cached when the hand-over states are bit-identical (same frames 0..t0-1 arithmetic)."""
import argparse
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import slice_parity  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--t0", type=int, default=60)
ap.add_argument("--n", type=int, default=120)
ap.add_argument("--cache", default=None)
ap.add_argument("--overlap", type=int, default=1)
ap.add_argument("--event-seed", type=int, default=42)
ap.add_argument("--weights-seed", type=int, default=0)
ap.add_argument("--inject", type=int, default=0, help="1: start from the cached hand-over states (same trajectory for every variant)")
a = ap.parse_args()
t = time.time()
print(f"# lib = {os.environ.get('URNN_LIB', 'default')}  env: " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("URNN_TUNE")))
res = slice_parity.run_slice(torch.device("cuda:0"), a.t0, a.n, weights_seed=a.weights_seed, event_seed=a.event_seed, cache=a.cache,
                             overlap=bool(a.overlap), inject=bool(a.inject))
slice_parity.report(res)
print(f"# {time.time() - t:.0f} s")
