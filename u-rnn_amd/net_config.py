"""Architecture config loading and state-shape arithmetic (host glue).

Mirrors the reference's ``src/lib/utils/net_config.py`` interface
(``load_net_config`` :29-56, ``get_state_shapes`` :59-116, ``get_input_channels`` :119-142)
and reads the same YAML keys, so reference config files work unchanged.

One build-side extension: ``get_state_shapes`` takes ``batch`` (default 1).  The reference
hard-codes a batch dimension of 1 (net_config.py:104-114) although the model itself is
per-sample; event batching needs B-sized states.
"""
import os

import yaml

_DEFAULT_CFG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs", "network.yaml")


def load_net_config(cfg_path=None):
    """Return the ``model`` section of a network YAML (default: the packaged table)."""
    path = _DEFAULT_CFG if cfg_path is None else cfg_path
    if not os.path.isfile(path):
        raise FileNotFoundError(
            f"Network config not found: {path}\n"
            f"Make sure configs/network.yaml exists or pass --net_config.")
    with open(path, "r", encoding="utf-8") as fh:
        return yaml.safe_load(fh)["model"]


def _cumulative_scales(factors):
    out, s = [], 1
    for f in factors:
        s *= int(f)
        out.append(s)
    return out


def get_state_shapes(net_cfg, input_height, input_width, batch=1):
    """Shapes of the six hidden states, order [e1, e2, e3, d1(deepest), d2, d3(full res)]."""
    enc = [int(c) for c in net_cfg["encoder"]["gru_channels"]]
    dec = [int(c) for c in net_cfg["decoder"]["gru_channels"]]
    scales = _cumulative_scales(net_cfg["encoder"]["downsample_factors"])
    n = len(enc)
    if len(dec) != n or len(scales) != n:
        raise ValueError("encoder/decoder stage counts must match")
    enc_shapes = [(batch, enc[k], input_height // scales[k], input_width // scales[k]) for k in range(n)]
    dec_shapes = [(batch, dec[k], input_height // scales[n - 1 - k], input_width // scales[n - 1 - k])
                  for k in range(n)]
    return enc_shapes + dec_shapes


def get_input_channels(net_cfg, historical_nums):
    """rain history + cumulative-rain history + DEM + impervious + manhole."""
    return int(historical_nums) * 2 + 3
