"""Generate the golden fixtures under tests/golden/ by running the REFERENCE implementation.

Runs only in the build container (needs /root/reference; the reference never travels to the GPU
box).  The reference is imported, its weights are overwritten with the build's seeded generator
(urnn_amd.weights.make_state_dict, non-trivial GroupNorm/LayerNorm affines) and its own modules
produce every expected output stored here.  Fixtures are DATA ONLY: inputs (or the seeds that
regenerate them) and the reference's outputs.

Two oracle-side shims, applied here and nowhere else:
  1. ``torch.Tensor.cuda = identity``   -- ConvRNN.py:136,146 / decoder.py:132 hard-code .cuda()
  2. ``sys.modules['wandb'] = stub``    -- test.py:5 imports wandb (not installed)

Usage:  python tests/golden/make_golden.py
"""
import copy
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/code"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

torch.Tensor.cuda = lambda self, *a, **k: self           # shim 1
sys.modules.setdefault("wandb", types.ModuleType("wandb"))  # shim 2

from src.lib.model.networks.net_params import get_network_params  # noqa: E402
from src.lib.model.networks.model import ED  # noqa: E402
from src.lib.model.networks.ConvRNN import CGRU_cell  # noqa: E402
from src.lib.model.networks.utils import make_layers  # noqa: E402
from src.lib.model.networks.head.flood_head import YOLOXHead  # noqa: E402
from src.lib.utils.net_config import load_net_config  # noqa: E402
from src.lib.utils.general import initialize_states  # noqa: E402
from src.lib.dataset.Dynamic2DFlood import preprocess_inputs  # noqa: E402

import urnn_amd.weights as uw  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)
CFG = load_net_config()


def ref_net(H, W, C, seed):
    """Reference ED with the build's seeded weights loaded into every alias key (strict load)."""
    ep, dp = get_network_params(False, H, W, C, CFG)
    net = ED(False, ep, dp, 0.5, False, H, W)
    sd = uw.make_state_dict(H, W, C, seed=seed)
    full = {}
    for key in net.state_dict().keys():
        canon = key
        # alias keys registered by the checkpoint wrappers (ConvRNN.py:108-109, encoder.py:106-117, ...)
        canon = canon.replace("_wrapper.module.", ".")
        canon = canon.replace(".conv1_module.", ".conv1.").replace(".conv2_module.", ".conv2.")
        assert canon in sd, (key, canon)
        full[key] = torch.from_numpy(sd[canon])
    net.load_state_dict(full, strict=True)
    net.eval()
    return net, sd


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def rnd(rs, *shape, scale=1.0):
    return (scale * rs.standard_normal(shape)).astype(np.float32)


class head_taps:
    """Forward hooks on the reference head while a whole ED.forward / test.Inference runs: per call the class map
    (YOLOXHead output channel 1, flood_head.py:166-177) and the pre-mask regression (reg_preds output, flood_head.py:160-164),
    each as (B, H, W) -- the head flattens its two leading dims (flood_head.py:145-150), whichever order the caller used."""

    def __init__(self, net):
        self.net, self.cls, self.raw, self._h = net, [], [], []

    def __enter__(self):
        self._h.append(self.net.head.register_forward_hook(lambda m, i, o: self.cls.append(o.detach().reshape((-1,) + tuple(o.shape[-3:]))[:, 1].numpy().copy())))
        self._h.append(self.net.head.reg_preds.register_forward_hook(lambda m, i, o: self.raw.append(o.detach()[:, 0].numpy().copy())))
        return self

    def __exit__(self, *a):
        for h in self._h:
            h.remove()


def gen_kernels(path):
    """Per-kernel vectors at 16x16 (and 8x8 / 4x4 deep stages) from the reference sub-modules."""
    rs = np.random.RandomState(1234)
    H = W = 16
    C = 9
    net, sd = ref_net(H, W, C, seed=7)
    out = {"H": H, "W": W, "C": C, "weights_seed": 7}
    with torch.no_grad():
        for B in (1, 2):
            tag = f"B{B}"
            # encoder stage convs (encoder.py:140-151)
            x = rnd(rs, B, C, H, W)
            out[f"s1_in_{tag}"] = x
            out[f"s1_out_{tag}"] = net.encoder.stage1(T(x)).numpy()
            x = rnd(rs, B, 64, H, W)
            out[f"s2_in_{tag}"] = x
            out[f"s2_out_{tag}"] = net.encoder.stage2(T(x)).numpy()
            x = rnd(rs, B, 96, H // 2, W // 2)
            out[f"s3_in_{tag}"] = x
            out[f"s3_out_{tag}"] = net.encoder.stage3(T(x)).numpy()
            # encoder cells (ConvRNN.py:111-194): inputs (S=1,B,I,H,W), hidden (B,F,H,W)
            for i, (I, F, h, w) in enumerate([(16, 64, H, W), (64, 96, H // 2, W // 2), (96, 96, H // 4, W // 4)], 1):
                x = rnd(rs, B, I, h, w)
                hh = rnd(rs, B, F, h, w, scale=0.5)
                y = getattr(net.encoder, f"rnn{i}")(T(x)[None], T(hh))[0].numpy()
                out[f"enc{i}_x_{tag}"], out[f"enc{i}_h_{tag}"], out[f"enc{i}_out_{tag}"] = x, hh, y
            # decoder cells: hidden = cat(e, d) (decoder.py:130-135); stage 3 has x == None
            for i, (I, F, h, w) in {3: (96, 96, H // 4, W // 4), 2: (96, 96, H // 2, W // 2), 1: (96, 64, H, W)}.items():
                e = rnd(rs, B, F, h, w, scale=0.5)
                d = rnd(rs, B, F, h, w, scale=0.5)
                out[f"dec{i}_e_{tag}"], out[f"dec{i}_d_{tag}"] = e, d
                st = torch.cat((T(e), T(d)), 1)
                if i == 3:
                    y = net.decoder.rnn3(None, st)[0].numpy()
                else:
                    x = rnd(rs, B, I, h, w)
                    out[f"dec{i}_x_{tag}"] = x
                    y = getattr(net.decoder, f"rnn{i}")(T(x)[None], st)[0].numpy()
                out[f"dec{i}_out_{tag}"] = y
            # deconvs + final decoder conv (decoder.py:150-164)
            x = rnd(rs, B, 96, H // 4, W // 4)
            out[f"dc3_in_{tag}"], out[f"dc3_out_{tag}"] = x, net.decoder.stage3(T(x)).numpy()
            x = rnd(rs, B, 96, H // 2, W // 2)
            out[f"dc2_in_{tag}"], out[f"dc2_out_{tag}"] = x, net.decoder.stage2(T(x)).numpy()
            x = rnd(rs, B, 64, H, W)
            out[f"dc1_in_{tag}"], out[f"dc1_out_{tag}"] = x, net.decoder.stage1(T(x)).numpy()
            # head (flood_head.py:131-177): input (S,B,16,H,W) -> (S,B,2,H,W); raw reg via sub-modules
            f = rnd(rs, B, 16, H, W)
            o = net.head(T(f)[None])[0].numpy()
            t = net.head.stems(T(f))
            raw = net.head.reg_preds(net.head.reg_convs(t)).numpy()
            out[f"head_in_{tag}"] = f
            out[f"head_masked_{tag}"], out[f"head_cls_{tag}"], out[f"head_raw_{tag}"] = o[:, 0], o[:, 1], raw[:, 0]
            # one full ED.forward step from non-zero states (model.py:65-121)
            x = rnd(rs, B, 1, C, H, W, scale=0.5)
            shapes = [(B, 64, H, W), (B, 96, H // 2, W // 2), (B, 96, H // 4, W // 4),
                      (B, 96, H // 4, W // 4), (B, 96, H // 2, W // 2), (B, 64, H, W)]
            st = [rnd(rs, *s, scale=0.5) for s in shapes]
            with head_taps(net) as tap:
                res = net(T(x), *[T(s) for s in st])
            out[f"step_x_{tag}"] = x
            for k, s in enumerate(st):
                out[f"step_state{k}_{tag}"] = s
            out[f"step_reg_{tag}"] = res[0].numpy()
            # the class map and the pre-mask regression of the same call (flood_head.py:166-177), for the flip-tolerant comparison
            out[f"step_cls_{tag}"], out[f"step_raw_{tag}"] = tap.cls[0], tap.raw[0]
            for k in range(6):
                out[f"step_newstate{k}_{tag}"] = res[1 + k].numpy()
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def event_to_torch(ev):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in ev.items()}


def gen_preprocess(path):
    """preprocess_inputs (Dynamic2DFlood.py:265-320) at t in {0, n-1, n, T-1}, scalar and spatial rain."""
    out = {}
    H, W, nums, Tn = 12, 20, 5, 14
    out.update({"H": H, "W": W, "nums": nums, "T": Tn, "rain_max": 6.0, "cumsum_max": 250.0, "event_seed": 3})
    for spatial in (False, True):
        for B in (1, 2):
            ev = uw.make_event(Tn, H, W, 6.0, seed=3, spatial_rain=spatial, batch=B)
            tev = event_to_torch(ev)
            for t in (0, nums - 1, nums, Tn - 1):
                y = preprocess_inputs(t, tev, torch.device("cpu"), nums=nums, rain_max=6.0, cumsum_rain_max=250.0)
                out[f"pre_sp{int(spatial)}_B{B}_t{t}"] = y.numpy()
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def gen_rollout(path, H, W, nums, Tn, rain_max, cumsum_max, wseed, eseed, spatial=False, every=1):
    """T-step rollout exactly as test.Inference does it (test.py:326-377): zero states, per-frame
    preprocess_inputs, state carry; additionally records cls and pre-mask reg via the head sub-modules."""
    C = 2 * nums + 3
    net, sd = ref_net(H, W, C, seed=wseed)
    ev = uw.make_event(Tn, H, W, rain_max, seed=eseed, spatial_rain=spatial, batch=1)
    tev = event_to_torch(ev)
    dev = torch.device("cpu")
    regs, clss, raws = [], [], []
    # the same modules in float64: the "exact" rollout, used to measure the fp32 reference's own roundoff
    net64 = copy.deepcopy(net).double()
    raws64, clss64 = [], []
    with torch.no_grad():
        st = initialize_states(dev, input_height=H, input_width=W, net_cfg=CFG)
        st64 = tuple(s.double() for s in st)
        for t in range(Tn):
            x = preprocess_inputs(t, tev, dev, nums=nums, rain_max=rain_max, cumsum_rain_max=cumsum_max)
            enc64 = net64.encoder(x.double().permute(1, 0, 2, 3, 4), list(st64[:3]))
            feat64, dec64 = net64.decoder(enc64, list(st64[3:]))
            o64 = net64.head(feat64)
            raw64 = net64.head.reg_preds(net64.head.reg_convs(net64.head.stems(feat64.reshape(-1, 16, H, W))))
            st64 = tuple(enc64) + tuple(dec64)
            raws64.append(raw64[0, 0].float().numpy())
            clss64.append(o64[0, 0, 1].float().numpy())
            # replicate ED.forward but keep the head's intermediate outputs
            enc = net.encoder(x.permute(1, 0, 2, 3, 4), list(st[:3]))
            feat, dec = net.decoder(enc, list(st[3:]))
            o = net.head(feat)                      # (B,S,2,H,W)
            tt = net.head.stems(feat.reshape(-1, 16, H, W))
            raw = net.head.reg_preds(net.head.reg_convs(tt))
            # cross-check against the real ED.forward on the same inputs
            full = net(x, *st)
            assert torch.equal(full[0], o[:, :, 0])
            st = tuple(enc) + tuple(dec)
            regs.append(o[0, 0, 0].numpy())
            clss.append(o[0, 0, 1].numpy())
            raws.append(raw[0, 0].numpy())
    out = {"H": H, "W": W, "nums": nums, "T": Tn, "rain_max": rain_max, "cumsum_max": cumsum_max,
           "weights_seed": wseed, "event_seed": eseed, "spatial": int(spatial), "every": every,
           "reg": np.stack(regs)[::every], "cls": np.stack(clss)[::every], "raw": np.stack(raws)[::every]}
    out["raw64"] = np.stack(raws64)[::every]
    out["cls64"] = np.stack(clss64)[::every]
    for k in range(6):
        out[f"final_state{k}"] = st[k].numpy()
        out[f"final_state64_{k}"] = st64[k].float().numpy()
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def gen_inference_entry(path):
    """The reference's own entry point test.Inference on a synthetic event (tiny), to pin the build's
    Inference() mirror including its (T,H,W) output contract."""
    import test as ref_test  # noqa: F401  (reference test.py; needs the wandb shim)
    H, W, nums, Tn = 16, 16, 3, 6
    C = 2 * nums + 3
    net, sd = ref_net(H, W, C, seed=11)
    ev = uw.make_event(Tn, H, W, 60.0, seed=5, batch=1)
    with head_taps(net) as tap:
        y = ref_test.Inference(net, event_to_torch(ev), torch.device("cpu"), historical_nums=nums, rain_max=60.0,
                               cumsum_rain_max=250.0, input_height=H, input_width=W, net_cfg=CFG)
    np.savez_compressed(path, H=H, W=W, nums=nums, T=Tn, rain_max=60.0, cumsum_max=250.0, weights_seed=11,
                        event_seed=5, out=y, cls=np.concatenate(tap.cls), raw=np.concatenate(tap.raw))
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def gen_events_metrics(path):
    """Reference Dynamic2DFlood over the synthetic folder tree of tests/synth_dataset.py, and reference compute_metrics on
    seeded arrays (needs the wandb shim to import test.py)."""
    import tempfile
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import synth_dataset as sd
    from src.lib.dataset.Dynamic2DFlood import Dynamic2DFlood
    out = {}
    with tempfile.TemporaryDirectory() as root:
        lst = sd.write_tree(root)
        ds = Dynamic2DFlood(root, "test", event_list_file=lst, duration=sd.DURATION)
        out["len"] = len(ds)
        out["locations"] = np.array(ds.locations)
        out["event_names"] = np.array(ds.event_names)
        for i in range(len(ds)):
            inp, tgt, event_dir = ds[i]
            out[f"item{i}_dir"] = np.array(os.path.relpath(event_dir, root))
            out[f"item{i}_target"] = tgt.numpy()
            for k, v in inp.items():
                out[f"item{i}_{k}"] = v.numpy()
        one = Dynamic2DFlood(root, "test", event_list_file=lst, duration=sd.DURATION, location="location16")
        out["one_len"] = len(one)
    sys.modules.setdefault("wandb", types.ModuleType("wandb"))
    import test as ref_test
    rs = np.random.RandomState(77)
    gt = (rs.uniform(0, 1, (6, 9, 11)) ** 3 * 900).astype(np.float32)
    pred = (gt * rs.uniform(0.7, 1.2, gt.shape) + rs.normal(0, 20, gt.shape)).astype(np.float32)
    out["m_pred"], out["m_gt"] = pred, gt
    for thr in (150.0, 600.0):
        m = ref_test.compute_metrics(pred, gt, flood_thres=thr)
        for k, v in m.items():
            out[f"m_{int(thr)}_{k}"] = np.float64(v)
    np.savez_compressed(path, **out)


if __name__ == "__main__":
    gen_kernels(os.path.join(HERE, "kernels_16x16.npz"))
    gen_preprocess(os.path.join(HERE, "preprocess.npz"))
    # BASELINE config 1: lite hyper-parameters at 64x64, T=30
    gen_rollout(os.path.join(HERE, "rollout_64x64_T30.npz"), 64, 64, 3, 30, 60.0, 250.0, wseed=0, eseed=42, every=3)
    # non-square, spatial rain (Futian/UKEA style: nums=6 -> C=15), short
    gen_rollout(os.path.join(HERE, "rollout_24x40_T8_spatial.npz"), 24, 40, 6, 8, 5.0, 100.0, wseed=2, eseed=9,
                spatial=True)
    try:
        gen_inference_entry(os.path.join(HERE, "inference_entry_16x16_T6.npz"))
    except Exception as exc:  # pragma: no cover - depends on optional reference imports
        print("test.Inference import failed, skipped:", repr(exc))
    try:
        gen_events_metrics(os.path.join(HERE, "events_metrics.npz"))
    except Exception as exc:  # noqa: BLE001
        print("events/metrics goldens failed:", repr(exc))
