"""Sparse trace of the REFERENCE ITSELF over a whole event at full size (test infrastructure).

Runs only in the build container: imports /root/reference/code with the two shims of make_golden.py, loads the
build's seeded weights into the reference ``ED`` (model.py:52-121) and rolls the event exactly as ``test.Inference`` does
(test.py:326-377: zero states, per-frame ``preprocess_inputs``, ``net(x, *states)``, state carry) -- once with the reference's
modules as shipped (float32, "ref32") and once with a deep copy of the same modules in float64 ("ref64", the exact result of the
reference's own graph).  Nothing of the build's arithmetic is involved: this pins BOTH the HIP path and the C oracle to the
reference at the headline size and horizon (before round 6 the longest reference-generated fixture was 64x64 x T = 30).

Per sampled frame (every ``--stride``-th, every ``--peak-stride``-th through ``--peak``, plus the last) the trace keeps the
class map and the pre-mask regression of ref32 and ref64 on
  * ``--pixels`` fixed random pixels (RandomState(7): the same subset make_whole_event_trace.py uses, so the committed ORACLE
    trace and this one can be compared pixel by pixel without running either),
  * the ``--adversarial`` pixels where ref32 and ref64 differ most on that frame (where float32 roundoff is amplified most),
  * the ``--adversarial`` pixels whose ref64 class score is closest to the wet/dry threshold outside |cls - 0.5| <= 1e-5,
and, when the committed oracle trace of the same event exists, ref64 on that trace's own adversarial pixels; plus the full-plane
maxima (the floors of the parity metric), ref32's full-plane distance from ref64 per frame under both floors (the yardstick of
the bars), and a subset of every final state.  Data only: seeds, indices and the reference's outputs.

    python tests/golden/make_reference_trace.py                       # location1 500x500 C=63 T=360, ~25 min on 8 threads
    python tests/golden/make_reference_trace.py --H 400 --W 560 --nums 6 --T 72 --rain-max 5 --cumsum-max 100 --spatial \
        --weights-seed 17 --event-seed 23 --stride 2 --peak 0 0      # Futian (futian_scratch.yaml:41-51,66-68)
"""
import argparse
import copy
import os
import sys
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/code"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

torch.Tensor.cuda = lambda self, *a, **k: self           # shim 1: ConvRNN.py:136,146 / decoder.py:132 hard-code .cuda()
sys.modules.setdefault("wandb", types.ModuleType("wandb"))  # shim 2: test.py:5


def rel(a, b, plane_max, floor_frac):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float((np.abs(a - b) / np.maximum(np.abs(b), floor_frac * max(float(plane_max), 1e-30))).max())


class Taps:
    """Forward hooks on the reference head: the class map (YOLOXHead output channel 1, flood_head.py:166-177) and the pre-mask
    regression (reg_preds output, flood_head.py:160-164) of the current call, flattened to H*W."""

    def __init__(self, net):
        self.cls = self.raw = None
        net.head.register_forward_hook(lambda m, i, o: setattr(self, "cls", o.detach().reshape((-1,) + tuple(o.shape[-3:]))[0, 1].reshape(-1).numpy().copy()))
        net.head.reg_preds.register_forward_hook(lambda m, i, o: setattr(self, "raw", o.detach()[0, 0].reshape(-1).numpy().copy()))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--H", type=int, default=500)
    ap.add_argument("--W", type=int, default=500)
    ap.add_argument("--nums", type=int, default=30)
    ap.add_argument("--T", type=int, default=360)
    ap.add_argument("--rain-max", type=float, default=6.0)
    ap.add_argument("--cumsum-max", type=float, default=250.0)
    ap.add_argument("--spatial", action="store_true")
    ap.add_argument("--weights-seed", type=int, default=0)
    ap.add_argument("--event-seed", type=int, default=42)
    ap.add_argument("--stride", type=int, default=4)
    ap.add_argument("--pixels", type=int, default=4096)
    ap.add_argument("--peak", type=int, nargs=2, default=[60, 180])
    ap.add_argument("--peak-stride", type=int, default=2)
    ap.add_argument("--adversarial", type=int, default=1024)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--out", default=None)
    a = ap.parse_args(argv)
    torch.set_num_threads(a.threads)
    from src.lib.model.networks.net_params import get_network_params
    from src.lib.model.networks.model import ED
    from src.lib.utils.net_config import load_net_config
    from src.lib.utils.general import initialize_states
    from src.lib.dataset.Dynamic2DFlood import preprocess_inputs
    import urnn_amd.weights as uw
    H, W, nums, T = a.H, a.W, a.nums, a.T
    C = 2 * nums + 3
    out = a.out or os.path.join(HERE, f"reference_trace_{H}x{W}_T{T}.npz")
    cfg = load_net_config()
    ep, dp = get_network_params(False, H, W, C, cfg)
    net = ED(False, ep, dp, 0.5, False, H, W)
    sd = uw.make_state_dict(H, W, C, seed=a.weights_seed)
    full = {}
    for key in net.state_dict().keys():   # alias keys of the checkpoint wrappers (ConvRNN.py:108-109, encoder.py:106-117)
        canon = key.replace("_wrapper.module.", ".").replace(".conv1_module.", ".conv1.").replace(".conv2_module.", ".conv2.")
        full[key] = torch.from_numpy(sd[canon])
    net.load_state_dict(full, strict=True)
    net.eval()
    net64 = copy.deepcopy(net).double()
    tap32, tap64 = Taps(net), Taps(net64)
    ev = uw.make_event(T, H, W, a.rain_max, seed=a.event_seed, spatial_rain=a.spatial, batch=1)
    tev = {k: torch.from_numpy(np.asarray(v)) for k, v in ev.items()}
    rs = np.random.RandomState(7)
    pix = np.sort(rs.choice(H * W, size=min(a.pixels, H * W), replace=False))
    peak = np.arange(a.peak[0], min(a.peak[1], T - 1) + 1, a.peak_stride) if a.peak[1] > a.peak[0] else np.zeros(0, np.int64)
    frames = np.unique(np.concatenate([np.arange(0, T, a.stride), peak, [T - 1]])).astype(np.int64)
    # the committed oracle trace of the same event, if there is one: ref64 on ITS adversarial pixels too
    otrace = os.path.join(HERE, f"whole_event_{H}x{W}_T{T}.npz")
    og = np.load(otrace) if os.path.isfile(otrace) else None
    if og is not None and not (np.array_equal(og["frames"], frames) and np.array_equal(og["pixels"], pix) and int(og["weights_seed"]) == a.weights_seed
                               and int(og["event_seed"]) == a.event_seed and int(og["nums"]) == nums):
        print("oracle trace", otrace, "samples another event / other frames: its adversarial pixels are not recorded")
        og = None
    nadv = min(a.adversarial, H * W // 4)
    dev = torch.device("cpu")
    K = {k: [] for k in ("r32_raw", "r32_cls", "r64_raw", "r64_cls", "adv_idx", "a32_raw", "a32_cls", "a64_raw", "a64_cls", "oadv64_raw", "oadv64_cls",
                         "raw_max", "cls_max", "e_raw_full", "e_cls_full", "e_raw_full_strict", "e_cls_full_strict", "flips")}
    t0 = time.time()
    with torch.no_grad():
        st = initialize_states(dev, input_height=H, input_width=W, net_cfg=cfg)
        st64 = tuple(s.double() for s in st)
        fi = 0
        for t in range(T):
            x = preprocess_inputs(t, tev, dev, nums=nums, rain_max=a.rain_max, cumsum_rain_max=a.cumsum_max)
            res = net(x, *st)                       # ED.forward, model.py:65-121
            st = tuple(res[1:])
            res64 = net64(x.double(), *st64)
            st64 = tuple(res64[1:])
            if t in frames:
                r32, c32, r64, c64 = tap32.raw, tap32.cls, tap64.raw, tap64.cls
                rmax, cmax = float(np.abs(r64).max()), float(np.abs(c64).max())
                K["r32_raw"].append(r32[pix]); K["r32_cls"].append(c32[pix])
                K["r64_raw"].append(r64[pix].astype(np.float32)); K["r64_cls"].append(c64[pix].astype(np.float32))
                K["raw_max"].append(rmax); K["cls_max"].append(cmax)
                e = np.abs(r32.astype(np.float64) - r64) / np.maximum(np.abs(r64), 1e-3 * rmax)
                worst = np.argpartition(e, -nadv)[-nadv:]
                dist = np.abs(c64 - 0.5)
                dist[dist <= 1e-5] = np.inf
                near = np.argpartition(dist, nadv)[:nadv]
                ai = np.unique(np.concatenate([worst, near]))
                ai = np.pad(ai, (0, 2 * nadv - ai.size), mode="edge")
                K["adv_idx"].append(ai.astype(np.int32))
                K["a32_raw"].append(r32[ai]); K["a32_cls"].append(c32[ai])
                K["a64_raw"].append(r64[ai].astype(np.float32)); K["a64_cls"].append(c64[ai].astype(np.float32))
                if og is not None:
                    oi = og["adv_idx"][fi].astype(np.int64)
                    K["oadv64_raw"].append(r64[oi].astype(np.float32)); K["oadv64_cls"].append(c64[oi].astype(np.float32))
                K["e_raw_full"].append(rel(r32, r64, rmax, 0.1)); K["e_cls_full"].append(rel(c32, c64, cmax, 0.1))
                K["e_raw_full_strict"].append(rel(r32, r64, rmax, 1e-3)); K["e_cls_full_strict"].append(rel(c32, c64, cmax, 1e-3))
                K["flips"].append(int(((c32 >= 0.5) != (c64 >= 0.5)).sum()))
                fi += 1
            if t % 10 == 0 or t == T - 1:
                print(f"frame {t:4d}  {time.time() - t0:6.0f} s", flush=True)
    so = {}
    st_err, st_err_strict, st_max = [], [], []
    for k in range(6):
        f32, f64 = st[k].numpy().reshape(-1), st64[k].numpy().reshape(-1)
        idx = np.sort(rs.choice(f64.size, size=min(a.pixels, f64.size), replace=False))
        so[f"state{k}_idx"] = idx.astype(np.int64)
        so[f"state{k}_ref32"] = f32[idx]
        so[f"state{k}_ref64"] = f64[idx].astype(np.float32)
        st_max.append(float(np.abs(f64).max()))
        st_err.append(rel(f32, f64, st_max[-1], 0.1))
        st_err_strict.append(rel(f32, f64, st_max[-1], 1e-3))
    arr = {k: np.stack(v) for k, v in K.items() if k not in ("raw_max", "cls_max", "e_raw_full", "e_cls_full", "e_raw_full_strict", "e_cls_full_strict", "flips", "oadv64_raw", "oadv64_cls") and v}
    if og is not None:
        arr["oracle_adv_ref64_raw"], arr["oracle_adv_ref64_cls"] = np.stack(K["oadv64_raw"]), np.stack(K["oadv64_cls"])
    np.savez_compressed(
        out, H=H, W=W, nums=nums, T=T, rain_max=a.rain_max, cumsum_max=a.cumsum_max, spatial=int(a.spatial), weights_seed=a.weights_seed, event_seed=a.event_seed,
        frames=frames.astype(np.int32), pixels=pix.astype(np.int32), ref64_raw_plane_max=np.asarray(K["raw_max"]), ref64_cls_plane_max=np.asarray(K["cls_max"]),
        ref32_reg_err_full=np.asarray(K["e_raw_full"]), ref32_cls_err_full=np.asarray(K["e_cls_full"]), ref32_reg_err_full_strict=np.asarray(K["e_raw_full_strict"]),
        ref32_cls_err_full_strict=np.asarray(K["e_cls_full_strict"]), ref32_flips=np.asarray(K["flips"], np.int32), state_plane_max=np.asarray(st_max),
        ref32_state_err=np.asarray(st_err), ref32_state_err_strict=np.asarray(st_err_strict), torch_version=str(torch.__version__), threads=a.threads, **arr, **so)
    print(f"wrote {out} ({os.path.getsize(out) // 1024} KiB): {len(frames)} frames x ({len(pix)} random + {2 * nadv} adversarial) pixels; reference fp32 vs fp64, full plane, "
          f"worst frame: reg {max(K['e_raw_full']):.2e} (strict floor {max(K['e_raw_full_strict']):.2e}) cls {max(K['e_cls_full']):.2e} (strict {max(K['e_cls_full_strict']):.2e}); "
          f"final states {['%.1e' % v for v in st_err]}; {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
