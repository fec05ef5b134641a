"""Host glue mirrored from the reference's ``src/lib/utils/general.py``: zero-state construction and
dict-to-device transfer (general.py:36-95)."""
import torch

from .net_config import get_state_shapes, load_net_config


def to_device(inputs, device):
    return {key: value.to(device) for key, value in inputs.items()}


def initialize_states(device, input_height=500, input_width=500, net_cfg=None, batch=1):
    """Six zero states (e1, e2, e3, d1 deepest, d2, d3 full-res).  ``batch`` is a build-side extension; the
    reference hard-codes 1 (general.py:82-89)."""
    cfg = net_cfg if net_cfg is not None else load_net_config()
    shapes = get_state_shapes(cfg, input_height, input_width, batch=batch)
    return tuple(torch.zeros(s, device=device, dtype=torch.float32) for s in shapes)
