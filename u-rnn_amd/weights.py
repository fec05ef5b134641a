"""Deterministic synthetic weights and synthetic flood events (seeded numpy).

There are no published checkpoints or datasets offline, so parity tests, goldens and the
bench all use this generator.  Weights carry the reference's 79 unique ``state_dict`` names
and shapes (SURVEY 8a "Unique parameter tensors"); GroupNorm/LayerNorm affines are
deliberately non-trivial (gamma ~ U(0.5, 1.5), beta ~ N(0, 0.1)) so that affine bugs cannot
hide behind the default gamma=1, beta=0.

``np.random.RandomState`` (legacy MT19937 stream) is used because its output is frozen across
numpy versions: the goldens under tests/golden/ were produced with these exact numbers.
"""
from collections import OrderedDict

import numpy as np

from .net_config import load_net_config


def _conv(rs, cout, cin, kh=1, kw=1):
    return (rs.standard_normal((cout, cin, kh, kw)) / np.sqrt(cin)).astype(np.float32)


def _bias(rs, n):
    return (0.1 * rs.standard_normal(n)).astype(np.float32)


def _gamma(rs, shape):
    return rs.uniform(0.5, 1.5, size=shape).astype(np.float32)


def _beta(rs, shape):
    return (0.1 * rs.standard_normal(shape)).astype(np.float32)


def make_state_dict(input_height, input_width, input_channels, seed=0, net_cfg=None):
    """Return ``OrderedDict[name -> np.ndarray(float32)]`` with the reference's parameter names.

    Names/shapes follow the modules built by ``get_network_params`` (net_params.py:5-141),
    ``CGRU_cell`` (ConvRNN.py:73-109) and ``YOLOXHead`` (flood_head.py:62-129).
    """
    cfg = net_cfg if net_cfg is not None else load_net_config()
    rs = np.random.RandomState(seed)
    H, W = int(input_height), int(input_width)
    enc_conv = [int(c) for c in cfg["encoder"]["conv_out_channels"]]
    enc_gru = [int(c) for c in cfg["encoder"]["gru_channels"]]
    dec_gru = [int(c) for c in cfg["decoder"]["gru_channels"]]
    dec_conv = [int(c) for c in cfg["decoder"]["conv_out_channels"]]
    up = [int(c) for c in cfg["decoder"]["upsample_factors"]]
    n = len(enc_gru)
    sd = OrderedDict()

    # encoder stage convs: conv{k}_leaky_1 (net_params.py:80-88)
    enc_in = [int(input_channels)] + enc_gru[:-1]
    for k in range(n):
        sd[f"encoder.stage{k+1}.conv{k+1}_leaky_1.weight"] = _conv(rs, enc_conv[k], enc_in[k])
        sd[f"encoder.stage{k+1}.conv{k+1}_leaky_1.bias"] = _bias(rs, enc_conv[k])

    def gru(prefix, I, F, skip):
        K = I + (2 * F if skip else F)
        sd[f"{prefix}.conv1.0.weight"] = _conv(rs, 2 * F, K)
        sd[f"{prefix}.conv1.0.bias"] = _bias(rs, 2 * F)
        sd[f"{prefix}.conv1.1.weight"] = _gamma(rs, 2 * F)
        sd[f"{prefix}.conv1.1.bias"] = _beta(rs, 2 * F)
        sd[f"{prefix}.conv2.0.weight"] = _conv(rs, F, K)
        sd[f"{prefix}.conv2.0.bias"] = _bias(rs, F)
        sd[f"{prefix}.conv2.1.weight"] = _gamma(rs, F)
        sd[f"{prefix}.conv2.1.bias"] = _beta(rs, F)

    for k in range(n):
        gru(f"encoder.rnn{k+1}", enc_conv[k], enc_gru[k], skip=False)

    # decoder: index k=0 is the deepest stage == module "stage{n}" / "rnn{n}" (decoder.py:83-88)
    dec_in = [enc_gru[n - 1 - k] for k in range(n)]
    dec_gru_in = [dec_conv[0]] + [dec_conv[k - 1] for k in range(1, n)]
    for k in range(n):
        stage = n - k
        if up[k] > 1:
            name = f"decoder.stage{stage}.deconv{k+1}_leaky_1"
            # ConvTranspose2d weight layout (Cin, Cout, kh, kw)
            w = (rs.standard_normal((dec_in[k], dec_conv[k], up[k], up[k])) / np.sqrt(dec_in[k])).astype(np.float32)
            sd[f"{name}.weight"] = w
        else:
            name = f"decoder.stage{stage}.conv{k+1}_leaky_1"
            sd[f"{name}.weight"] = _conv(rs, dec_conv[k], dec_in[k])
        sd[f"{name}.bias"] = _bias(rs, dec_conv[k])
    for k in range(n):
        gru(f"decoder.rnn{n-k}", dec_gru_in[k], dec_gru[k], skip=True)

    # head (flood_head.py:78-118): 5 BaseConv blocks + 2 prediction convs
    ch = int(int(cfg["head"]["in_channels"]) * float(cfg["head"]["width"]))
    for blk in ("stems", "cls_convs.0", "cls_convs.1", "reg_convs.0", "reg_convs.1"):
        sd[f"head.{blk}.conv.weight"] = _conv(rs, ch, ch)
        sd[f"head.{blk}.ln.weight"] = _gamma(rs, (ch, H, W))
        sd[f"head.{blk}.ln.bias"] = _beta(rs, (ch, H, W))
    for blk in ("cls_preds", "reg_preds"):
        sd[f"head.{blk}.conv.weight"] = _conv(rs, 1, ch)
        sd[f"head.{blk}.conv.bias"] = _bias(rs, 1)
    return sd


def make_event(T, input_height, input_width, rain_max, seed=42, spatial_rain=False, batch=1):
    """Synthetic flood event in the layout ``Dynamic2DFlood.__getitem__`` + DataLoader produce
    (Dynamic2DFlood.py:181-240): numpy float32 arrays

      absolute_DEM (B,1,1,H,W) in mm, max_DEM / min_DEM (B,), impervious, manhole (B,1,1,H,W),
      rainfall / cumsum_rainfall (B,T,1,1,1) scalar or (B,T,1,H,W) spatial.

    Recipe follows the reference notebook (quickstart.ipynb cells 6/13): DEM ~ U(0,10) m,
    impervious ~ U(0,1), manhole = [U(0,1) > 0.95], rainfall ~ U(0, rain_max/2) per step.
    """
    rs = np.random.RandomState(seed)
    B, H, W = int(batch), int(input_height), int(input_width)
    dem = (rs.uniform(0.0, 10.0, size=(B, 1, 1, H, W)) * 1000.0).astype(np.float32)
    imp = rs.uniform(0.0, 1.0, size=(B, 1, 1, H, W)).astype(np.float32)
    man = (rs.uniform(0.0, 1.0, size=(B, 1, 1, H, W)) > 0.95).astype(np.float32)
    if spatial_rain:
        rain = rs.uniform(0.0, rain_max / 2.0, size=(B, T, 1, H, W)).astype(np.float32)
    else:
        rain = rs.uniform(0.0, rain_max / 2.0, size=(B, T, 1, 1, 1)).astype(np.float32)
    cums = np.cumsum(rain, axis=1, dtype=np.float32)
    return {
        "absolute_DEM": dem,
        "max_DEM": dem.reshape(B, -1).max(axis=1),
        "min_DEM": dem.reshape(B, -1).min(axis=1),
        "impervious": imp,
        "manhole": man,
        "rainfall": rain,
        "cumsum_rainfall": cums,
    }
