"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE -- never imported by the product path).

Loads ``oracle/liburnn_oracle.so`` (built from ``urnn_oracle.c`` by ``oracle/build.py``) and composes
its primitives into the reference's one-timestep dataflow (``ED.forward``, model.py:65-121) and the
``Inference`` rollout (test.py:326-377).  All arrays are numpy float32, NCHW contiguous.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this module.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f = ctypes.POINTER(ctypes.c_float)


def _ptr(a):
    if a is None:
        return ctypes.cast(None, _f)
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"], "oracle wants contiguous float32"
    return a.ctypes.data_as(_f)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liburnn_oracle.so")
        if not os.path.isfile(path):
            from . import build as _b
            _b.build_oracle()
        _LIB = ctypes.CDLL(path)
        _LIB.orc_num_threads.restype = ctypes.c_int
    return _LIB


def num_threads():
    return int(lib().orc_num_threads())


def set_num_threads(n):
    lib().orc_set_num_threads(ctypes.c_int(int(n)))


def c32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---------------------------------------------------------------------------------------------
# primitives
# ---------------------------------------------------------------------------------------------
def stage_conv(x, w, b, pool, slope=0.2):
    """[AvgPool2](LeakyReLU(conv1x1(x))) -- encoder.py:140-151 / utils.py:92-121."""
    x = c32(x)
    B, Cin, H, W = x.shape
    w2 = c32(w).reshape(w.shape[0], -1)
    Cout = w2.shape[0]
    out = np.empty((B, Cout, H // 2, W // 2) if pool else (B, Cout, H, W), np.float32)
    scratch = np.empty((B, Cout, H, W), np.float32) if pool else None
    lib().orc_stage_conv(_ptr(x), _ptr(w2), _ptr(c32(b)), _ptr(out), _ptr(scratch), B, Cin, Cout, H, W,
                         1 if pool else 0, ctypes.c_float(slope))
    return out


def deconv2x2(x, w, b, slope=0.2):
    """LeakyReLU(ConvTranspose2d(k=2,s=2)(x)) -- utils.py:95-107; w is (Cin, Cout, 2, 2)."""
    x = c32(x)
    B, Cin, H, W = x.shape
    w = c32(w)
    Cout = w.shape[1]
    out = np.empty((B, Cout, 2 * H, 2 * W), np.float32)
    lib().orc_deconv2x2(_ptr(x), _ptr(w), _ptr(c32(b)), _ptr(out), B, Cin, Cout, H, W, ctypes.c_float(slope))
    return out


def gru_cell(x, e, h, p, eps=1e-5):
    """ConvGRU (e is None) / Skip-ConvGRU cell -- ConvRNN.py:111-194.  ``p`` holds the eight parameter
    arrays W1,b1,g1,be1,W2,b2,g2,be2; x may be None (decoder stage 3, x == 0 with I channels)."""
    h = c32(h)
    B, F, H, W = h.shape
    P = H * W
    W1 = c32(p["W1"]).reshape(2 * F, -1)
    W2 = c32(p["W2"]).reshape(F, -1)
    I = W1.shape[1] - (2 * F if e is not None else F)
    if x is not None:
        x = c32(x)
        assert x.shape[1] == I
    if e is not None:
        e = c32(e)
    out = np.empty_like(h)
    scratch = np.empty(B * 4 * F * P, np.float32)
    lib().orc_gru_cell(_ptr(x), _ptr(e), _ptr(h), _ptr(W1), _ptr(c32(p["b1"])), _ptr(c32(p["g1"])),
                       _ptr(c32(p["be1"])), _ptr(W2), _ptr(c32(p["b2"])), _ptr(c32(p["g2"])), _ptr(c32(p["be2"])),
                       _ptr(out), _ptr(scratch), B, I, F, ctypes.c_long(P), ctypes.c_float(eps))
    return out


def head(f, hp, cls_thred=0.5, eps=1e-5, slope=0.2):
    """YOLOXHead.forward + correction_depth -- flood_head.py:131-202.
    Returns (masked_reg, cls, raw_reg), each (B, H, W)."""
    f = c32(f)
    B, C, H, W = f.shape
    P = H * W
    masked = np.empty((B, H, W), np.float32)
    cls = np.empty((B, H, W), np.float32)
    raw = np.empty((B, H, W), np.float32)
    scratch = np.empty(3 * B * C * P, np.float32)
    lib().orc_head(_ptr(f), _ptr(hp["conv_w"]), _ptr(hp["ln_w"]), _ptr(hp["ln_b"]), _ptr(hp["cls_w"]), _ptr(hp["cls_b"]),
                   _ptr(hp["reg_w"]), _ptr(hp["reg_b"]), _ptr(masked), _ptr(cls), _ptr(raw), _ptr(scratch), B, C,
                   ctypes.c_long(P), ctypes.c_float(cls_thred), ctypes.c_float(eps), ctypes.c_float(slope))
    return masked, cls, raw


def preprocess_inputs(t, event, nums, rain_max, cumsum_rain_max):
    """Dynamic2DFlood.preprocess_inputs (:265-320): returns (B, 1, C, H, W) float32."""
    dem = c32(event["absolute_DEM"])
    B, H, W = dem.shape[0], dem.shape[-2], dem.shape[-1]
    rain = c32(event["rainfall"])
    cums = c32(event["cumsum_rainfall"])
    T = rain.shape[1]
    spatial = 0 if rain.shape[-1] == 1 and rain.shape[-2] == 1 else 1
    C = 2 * nums + 3
    out = np.empty((B, 1, C, H, W), np.float32)
    lib().orc_preprocess(int(t), _ptr(rain), _ptr(cums), T, spatial, _ptr(dem), _ptr(c32(event["impervious"])),
                         _ptr(c32(event["manhole"])), ctypes.c_float(float(np.asarray(event["min_DEM"]).ravel()[0])),
                         ctypes.c_float(float(np.asarray(event["max_DEM"]).ravel()[0])), _ptr(out), B, nums, H, W,
                         ctypes.c_float(rain_max), ctypes.c_float(cumsum_rain_max))
    return out


# ---------------------------------------------------------------------------------------------
# whole network
# ---------------------------------------------------------------------------------------------
def _gru_params(sd, prefix):
    return {"W1": sd[f"{prefix}.conv1.0.weight"], "b1": sd[f"{prefix}.conv1.0.bias"],
            "g1": sd[f"{prefix}.conv1.1.weight"], "be1": sd[f"{prefix}.conv1.1.bias"],
            "W2": sd[f"{prefix}.conv2.0.weight"], "b2": sd[f"{prefix}.conv2.0.bias"],
            "g2": sd[f"{prefix}.conv2.1.weight"], "be2": sd[f"{prefix}.conv2.1.bias"]}


def head_params(sd):
    blocks = ("stems", "cls_convs.0", "cls_convs.1", "reg_convs.0", "reg_convs.1")
    return {
        "conv_w": c32(np.stack([np.asarray(sd[f"head.{b}.conv.weight"]).reshape(16, 16) for b in blocks])),
        "ln_w": c32(np.stack([np.asarray(sd[f"head.{b}.ln.weight"]) for b in blocks])),
        "ln_b": c32(np.stack([np.asarray(sd[f"head.{b}.ln.bias"]) for b in blocks])),
        "cls_w": c32(np.asarray(sd["head.cls_preds.conv.weight"]).reshape(1, 16)),
        "cls_b": c32(sd["head.cls_preds.conv.bias"]),
        "reg_w": c32(np.asarray(sd["head.reg_preds.conv.weight"]).reshape(1, 16)),
        "reg_b": c32(sd["head.reg_preds.conv.bias"]),
    }


class OracleNet:
    """One-timestep U-RNN forward on the CPU oracle, from a reference-named state dict of numpy arrays.

    ``step`` follows ED.forward (model.py:65-121): encoder (encoder.py:187-215), decoder
    (decoder.py:173-217; state order d1 = deepest), head (flood_head.py:131-177)."""

    def __init__(self, sd, cls_thred=0.5):
        self.sd = {k: np.asarray(v, dtype=np.float32) for k, v in sd.items()}
        self.cls_thred = float(cls_thred)
        self.enc = [_gru_params(self.sd, f"encoder.rnn{i}") for i in (1, 2, 3)]
        self.dec = {i: _gru_params(self.sd, f"decoder.rnn{i}") for i in (1, 2, 3)}
        self.hp = head_params(self.sd)

    def step(self, x, states, want_aux=False):
        """x: (B, C, H, W); states: [e1, e2, e3, d1(deepest), d2, d3].  Returns (masked_reg (B,H,W),
        new_states, aux) where aux = {"cls", "reg_raw", "feat"} when want_aux."""
        sd = self.sd
        e1p, e2p, e3p, d1p, d2p, d3p = states
        a = stage_conv(x, sd["encoder.stage1.conv1_leaky_1.weight"], sd["encoder.stage1.conv1_leaky_1.bias"], False)
        e1 = gru_cell(a, None, e1p, self.enc[0])
        a = stage_conv(e1, sd["encoder.stage2.conv2_leaky_1.weight"], sd["encoder.stage2.conv2_leaky_1.bias"], True)
        e2 = gru_cell(a, None, e2p, self.enc[1])
        a = stage_conv(e2, sd["encoder.stage3.conv3_leaky_1.weight"], sd["encoder.stage3.conv3_leaky_1.bias"], True)
        e3 = gru_cell(a, None, e3p, self.enc[2])
        # decoder, deepest first; x == 0 at stage 3
        d1 = gru_cell(None, e3, d1p, self.dec[3])
        u = deconv2x2(d1, sd["decoder.stage3.deconv1_leaky_1.weight"], sd["decoder.stage3.deconv1_leaky_1.bias"])
        d2 = gru_cell(u, e2, d2p, self.dec[2])
        u = deconv2x2(d2, sd["decoder.stage2.deconv2_leaky_1.weight"], sd["decoder.stage2.deconv2_leaky_1.bias"])
        d3 = gru_cell(u, e1, d3p, self.dec[1])
        feat = stage_conv(d3, sd["decoder.stage1.conv3_leaky_1.weight"], sd["decoder.stage1.conv3_leaky_1.bias"], False)
        masked, cls, raw = head(feat, self.hp, self.cls_thred)
        aux = {"cls": cls, "reg_raw": raw, "feat": feat} if want_aux else None
        return masked, [e1, e2, e3, d1, d2, d3], aux


def zero_states(batch, H, W):
    """initialize_states (general.py:50-95) with the published channel table."""
    return [np.zeros(s, np.float32) for s in
            [(batch, 64, H, W), (batch, 96, H // 2, W // 2), (batch, 96, H // 4, W // 4),
             (batch, 96, H // 4, W // 4), (batch, 96, H // 2, W // 2), (batch, 64, H, W)]]


def rollout(net, event, T, nums, rain_max, cumsum_rain_max, want_aux=False):
    """test.Inference (test.py:326-377): T single-step forwards with state carry, input assembly inside
    the loop.  Returns (frames (T,B,H,W), final_states, aux_list)."""
    dem = event["absolute_DEM"]
    B, H, W = dem.shape[0], dem.shape[-2], dem.shape[-1]
    states = zero_states(B, H, W)
    frames, auxs = [], []
    for t in range(T):
        x = preprocess_inputs(t, event, nums, rain_max, cumsum_rain_max)[:, 0]
        out, states, aux = net.step(x, states, want_aux)
        frames.append(out)
        auxs.append(aux)
    return np.stack(frames), states, auxs
