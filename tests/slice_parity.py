"""Mid-event slice of BASELINE configs[1] (location1: 500x500, C = 63, T = 360, seed 42) against the CPU oracle -- test infrastructure
shared by tests/test_hip_rollout.py::test_mid_event_slice_vs_oracle (driver-run, -m gpu) and tools/parity_slice.py (A/B of
arithmetic variants on one box).

The HIP engine rolls frames 0 .. T0-1 on the benchmarked schedule (captured graph, two kernel chains); its six states after frame
T0-1 are handed to the oracle and to the plain-float32 torch restatement (tests/torch_ref.py: the reference's own arithmetic),
and all three then run frames T0 .. T0+N-1 of the same event from the same states.  Per frame: cls and the pre-mask regression
vs the oracle under the relaxed floor (conftest.rel_err: 0.1 * max|b|, the 1e-4 bar) AND under SURVEY 8(c)'s strict floor
(1e-3 * max|b|), each next to the torch-fp32 error on the same frame (test.py:352-371 is the loop being restated)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)
from conftest import rel_err  # noqa: E402

H = W = 500
NUMS, RAIN_MAX, CUM_MAX, T_EVENT = 30, 6.0, 250.0, 360


def run_slice(dev, t0=60, n=120, weights_seed=0, event_seed=42, cache=None, overlap=True, log=print, inject=False):
    """Returns dict(rows=[(t, hip_reg, torch_reg, hip_cls, torch_cls, hip_reg_strict, torch_reg_strict, hip_cls_strict,
    torch_cls_strict)], states=[(k, hip, torch, hip_strict, torch_strict)]).  ``cache``: optional .npz path holding the oracle's
    and the torch-fp32 outputs for this (t0, n) and initial states (A/B runs of arithmetic variants that start from
    bit-identical states reuse it; the file is keyed by a checksum of the hand-over states).  ``inject``: when the cache exists,
    start the HIP engine (one kernel chain) from the cached hand-over states instead of its own frames 0 .. t0-1, so that every
    variant is compared on the same trajectory."""
    import torch_ref
    import urnn_amd.weights as uw
    from oracle import oracle as orc
    from urnn_amd.dataset import preprocess_inputs
    from urnn_amd.net_config import load_net_config
    from urnn_amd.networks import ED, get_network_params
    from urnn_amd.rollout import RolloutEngine

    C = 2 * NUMS + 3
    sd = uw.make_state_dict(H, W, C, seed=weights_seed)
    ep, dp = get_network_params(False, H, W, C, load_net_config())
    net = ED(False, ep, dp, 0.5, False, H, W)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net = net.to(dev).eval()
    ev = uw.make_event(T_EVENT, H, W, RAIN_MAX, seed=event_seed)
    injected = None
    if inject and cache and os.path.isfile(cache):
        z = np.load(cache)
        if int(z["t0"]) == t0 and int(z["n"]) == n:
            injected = [torch.from_numpy(z[f"start{k}"]).to(dev) for k in range(6)]
    eng = RolloutEngine(net, H, W, NUMS, RAIN_MAX, CUM_MAX, max_frames=T_EVENT, keep_raw=True,
                        overlap=overlap and injected is None, use_graph=True)
    eng.load_event(ev)
    eng.reset()
    if injected is None:
        if t0 > 0:
            eng.run(t0)
        torch.cuda.synchronize()
        start = [s.clone() for s in eng.final_states()]
    else:
        for s, v in zip(eng.states, injected):
            s.copy_(v)
        eng.t_dev.fill_(t0)
        eng._frames_done = t0
        start = injected
        log("HIP engine started from the cached hand-over states (one kernel chain)")
    eng.run(n)
    torch.cuda.synchronize()
    hip_raw = eng.out_raw[t0:t0 + n].cpu().numpy()
    hip_cls = eng.out_cls[t0:t0 + n].cpu().numpy()
    hip_states = [s.cpu().numpy() for s in eng.final_states()]

    start_np = [s.cpu().numpy() for s in start]
    key = float(sum(float(np.abs(s, dtype=np.float64).sum()) for s in start_np))
    ref = None
    if cache and os.path.isfile(cache):
        z = np.load(cache)
        if float(z["key"]) == key and int(z["t0"]) == t0 and int(z["n"]) == n:
            ref = {k: z[k] for k in z.files}
            log(f"oracle / torch-fp32 outputs from {cache}")
    if ref is None:
        onet = orc.OracleNet(sd)
        pt = {k: torch.from_numpy(v).to(dev) for k, v in sd.items()}
        ost = [s.copy() for s in start_np]
        tst = [s.clone() for s in start]
        o_raw, o_cls, t_raw, t_cls = [], [], [], []
        for t in range(t0, t0 + n):
            _, ost, aux = onet.step(orc.preprocess_inputs(t, ev, NUMS, RAIN_MAX, CUM_MAX)[:, 0], ost, True)
            with torch.no_grad():
                x = preprocess_inputs(t, ev, dev, nums=NUMS, rain_max=RAIN_MAX, cumsum_rain_max=CUM_MAX)[:, 0]
                _, tc, tr, tst = torch_ref.step(pt, x, tst, H, W)
            o_raw.append(aux["reg_raw"].reshape(1, H, W))
            o_cls.append(aux["cls"].reshape(1, H, W))
            t_raw.append(tr.cpu().numpy().reshape(1, H, W))
            t_cls.append(tc.cpu().numpy().reshape(1, H, W))
        ref = {"o_raw": np.stack(o_raw), "o_cls": np.stack(o_cls), "t_raw": np.stack(t_raw), "t_cls": np.stack(t_cls)}
        for k, s in enumerate(ost):
            ref[f"o_state{k}"] = s
        for k, s in enumerate(tst):
            ref[f"t_state{k}"] = s.cpu().numpy()
        if cache:
            np.savez(cache, key=key, t0=t0, n=n, **{f"start{k}": s for k, s in enumerate(start_np)}, **ref)
    rows = []
    for i in range(n):
        a, b = ref["o_raw"][i], ref["o_cls"][i]
        rows.append((t0 + i,
                     rel_err(hip_raw[i].reshape(a.shape), a), rel_err(ref["t_raw"][i], a),
                     rel_err(hip_cls[i].reshape(b.shape), b), rel_err(ref["t_cls"][i], b),
                     rel_err(hip_raw[i].reshape(a.shape), a, 1e-3), rel_err(ref["t_raw"][i], a, 1e-3),
                     rel_err(hip_cls[i].reshape(b.shape), b, 1e-3), rel_err(ref["t_cls"][i], b, 1e-3)))
    states = []
    for k in range(6):
        want = ref[f"o_state{k}"]
        states.append((k, rel_err(hip_states[k], want), rel_err(ref[f"t_state{k}"], want),
                       rel_err(hip_states[k], want, 1e-3), rel_err(ref[f"t_state{k}"], want, 1e-3)))
    return {"rows": rows, "states": states}


def report(res, log=print, every=10):
    rows = res["rows"]
    log("frame | pre-mask reg: HIP / torch-fp32 (floor 0.1 max) | cls: HIP / torch-fp32 | strict floor (1e-3 max) reg HIP / torch | cls HIP / torch")
    for r in rows:
        if (r[0] - rows[0][0]) % every == 0 or r is rows[-1]:
            log(f"{r[0]:5d} | {r[1]:.2e} / {r[2]:.2e} | {r[3]:.2e} / {r[4]:.2e} | {r[5]:.2e} / {r[6]:.2e} | {r[7]:.2e} / {r[8]:.2e}")
    mx = lambda j: max(r[j] for r in rows)
    wins_reg = sum(1 for r in rows if r[1] <= r[2])
    wins_cls = sum(1 for r in rows if r[3] <= r[4])
    log(f"max over {len(rows)} frames: reg HIP {mx(1):.2e} torch {mx(2):.2e} | cls HIP {mx(3):.2e} torch {mx(4):.2e} | "
        f"strict: reg HIP {mx(5):.2e} torch {mx(6):.2e} | cls HIP {mx(7):.2e} torch {mx(8):.2e}")
    log(f"frames where HIP <= torch-fp32: reg {wins_reg}/{len(rows)}, cls {wins_cls}/{len(rows)}; "
        f"mean error ratio HIP/torch: reg {np.mean([r[1] / r[2] for r in rows]):.2f}, cls {np.mean([r[3] / r[4] for r in rows]):.2f}")
    log("final states (state, HIP, torch-fp32 | strict floor HIP, torch): " +
        ", ".join(f"({k}, {a:.2e}, {b:.2e} | {c:.2e}, {d:.2e})" for k, a, b, c, d in res["states"]))
