"""Sparse ORACLE trace of a whole event at full size (test infrastructure).

The headline parity claim -- the benchmarked schedule stays within max(1e-4, 3 x plain-fp32-torch) of the fp64-accumulating CPU
oracle over all 360 frames of BASELINE configs[1] (500x500, C = 63; reference loop: test.py:326-377) -- used to need 7.5 minutes
of oracle time per run and was therefore opt-in.  This script runs the C oracle (oracle/urnn_oracle.c, pinned to the
reference-generated goldens by tests/test_oracle.py) ONCE over the whole event and keeps, for every ``--stride``-th frame, the
pre-mask regression and the class map on a fixed random subset of ``--pixels`` pixels, plus a subset of every final state: a few
MB that `tests/test_hip_rollout.py::test_whole_event_vs_committed_oracle_trace` compares all sampled frames against in seconds.
Next to the oracle it records what the reference's OWN arithmetic does on the same frames (tests/torch_ref.py in float32, on the
GPU when there is one): the per-frame yardstick of the 3x rule.

Inputs are the seeded synthetic ones the GPU tests use (urnn_amd.weights: make_state_dict(seed 0), make_event(seed 42)), so the
trace is reproducible anywhere; it was generated on the GPU box's 128 host threads:

    python tests/golden/make_whole_event_trace.py            # -> tests/golden/whole_event_500x500_T360.npz (~9 min)
    python tests/golden/make_whole_event_trace.py --H 128 --W 128 --nums 3 --rain-max 60 --out tests/golden/whole_event_128x128_T360.npz
"""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
for p in (REPO, os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def rel_err(a, b, plane_max):
    """conftest.rel_err on a SUBSET of a tensor: the floor is 0.1 x the max |.| of the WHOLE reference tensor."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    floor = 0.1 * max(float(plane_max), 1e-30)
    return float((np.abs(a - b) / np.maximum(np.abs(b), floor)).max())


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--H", type=int, default=500)
    ap.add_argument("--W", type=int, default=500)
    ap.add_argument("--nums", type=int, default=30)
    ap.add_argument("--T", type=int, default=360)
    ap.add_argument("--rain-max", type=float, default=6.0)
    ap.add_argument("--cumsum-max", type=float, default=250.0)
    ap.add_argument("--weights-seed", type=int, default=0)
    ap.add_argument("--event-seed", type=int, default=42)
    ap.add_argument("--stride", type=int, default=4)
    ap.add_argument("--pixels", type=int, default=4096)
    ap.add_argument("--out", default=None)
    a = ap.parse_args(argv)
    import torch
    import torch_ref
    import urnn_amd.weights as uw
    from oracle import oracle as orc
    H, W, nums, T = a.H, a.W, a.nums, a.T
    out = a.out or os.path.join(HERE, f"whole_event_{H}x{W}_T{T}.npz")
    sd = uw.make_state_dict(H, W, 2 * nums + 3, seed=a.weights_seed)
    ev = uw.make_event(T, H, W, a.rain_max, seed=a.event_seed)
    rs = np.random.RandomState(7)
    pix = np.sort(rs.choice(H * W, size=min(a.pixels, H * W), replace=False))
    frames = np.arange(0, T, a.stride)
    if frames[-1] != T - 1:
        frames = np.append(frames, T - 1)
    dev = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
    onet = orc.OracleNet(sd)
    pt = {k: torch.from_numpy(v).to(dev) for k, v in sd.items()}
    ost = orc.zero_states(1, H, W)
    tst = [torch.zeros(s.shape, device=dev) for s in ost]
    raw_o, cls_o, t_reg, t_cls, raw_max, cls_max = [], [], [], [], [], []
    t0 = time.time()
    for t in range(T):
        xo = orc.preprocess_inputs(t, ev, nums, a.rain_max, a.cumsum_max)[:, 0]
        _, ost, aux = onet.step(xo, ost, True)
        with torch.no_grad():
            _, tcls, traw, tst = torch_ref.step(pt, torch.from_numpy(np.ascontiguousarray(xo)).to(dev), tst, H, W)
        if t in frames:
            ro, co = aux["reg_raw"].reshape(-1), aux["cls"].reshape(-1)
            raw_o.append(ro[pix].astype(np.float32))
            cls_o.append(co[pix].astype(np.float32))
            raw_max.append(float(np.abs(ro).max()))
            # the yardstick on the SAME subset and floor the test uses (floor = 0.1 x the whole plane's max |reg|)
            cls_max.append(float(np.abs(co).max()))
            t_reg.append(rel_err(traw.cpu().numpy().reshape(-1)[pix], ro[pix], raw_max[-1]))
            t_cls.append(rel_err(tcls.cpu().numpy().reshape(-1)[pix], co[pix], cls_max[-1]))
        if t % 20 == 0 or t == T - 1:
            print(f"frame {t:4d}  {time.time() - t0:6.0f} s", flush=True)
    st_idx, st_o, st_t, st_max = [], [], [], []
    for k, (so, stt) in enumerate(zip(ost, tst)):
        flat = np.asarray(so).reshape(-1)
        idx = np.sort(rs.choice(flat.size, size=min(a.pixels, flat.size), replace=False))
        st_idx.append(idx)
        st_o.append(flat[idx].astype(np.float32))
        st_max.append(float(np.abs(flat).max()))
        st_t.append(rel_err(stt.cpu().numpy().reshape(-1)[idx], flat[idx], st_max[-1]))
    np.savez_compressed(
        out, H=H, W=W, nums=nums, T=T, rain_max=a.rain_max, cumsum_max=a.cumsum_max, weights_seed=a.weights_seed, event_seed=a.event_seed,
        frames=frames.astype(np.int32), pixels=pix.astype(np.int32), oracle_raw=np.stack(raw_o), oracle_cls=np.stack(cls_o),
        oracle_raw_plane_max=np.asarray(raw_max, np.float64), oracle_cls_plane_max=np.asarray(cls_max, np.float64), torch32_reg_err=np.asarray(t_reg), torch32_cls_err=np.asarray(t_cls),
        torch32_device=str(dev), **{f"state{k}_idx": st_idx[k].astype(np.int64) for k in range(6)},
        **{f"state{k}_oracle": st_o[k] for k in range(6)}, state_plane_max=np.asarray(st_max), torch32_state_err=np.asarray(st_t))
    print(f"wrote {out}: {len(frames)} frames x {len(pix)} pixels; torch-fp32 worst reg {max(t_reg):.2e} cls {max(t_cls):.2e} "
          f"states {['%.1e' % v for v in st_t]}; {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
