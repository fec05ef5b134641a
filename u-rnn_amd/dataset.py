"""Feature half of the reference's ``Dynamic2DFlood.py`` (:265-385): per-frame input assembly on the GPU and
the Min-Max helpers.  The event dict layout is the one ``Dynamic2DFlood.__getitem__`` + DataLoader produce
(Dynamic2DFlood.py:181-240); ``urnn_amd.weights.make_event`` generates synthetic events in that layout."""
import numpy as np
import torch

from . import ops


def MinMaxScaler(data, max, min):  # noqa: A002 - reference signature (Dynamic2DFlood.py:369-376)
    return (data - min) / (max - min)


def r_MinMaxScaler(data, max, min):  # noqa: A002 - reference signature (Dynamic2DFlood.py:379-385)
    return data * (max - min) + min


def event_to_device(inputs, device):
    """Reference-layout event dict (numpy or torch, any device) -> the flat device tensors the kernels read:
    rain/cumsum (B,T) or (B,T,H,W), dem/imperv/manhole (B,H,W), dem_min/dem_max python floats (sample 0's,
    as the reference indexes ``inputs["max_DEM"][0]``, Dynamic2DFlood.py:305-306)."""
    def t32(v):
        v = torch.as_tensor(np.asarray(v) if not torch.is_tensor(v) else v)
        return v.to(device=device, dtype=torch.float32)

    dem = t32(inputs["absolute_DEM"])
    B, H, W = dem.shape[0], dem.shape[-2], dem.shape[-1]
    rain, cums = t32(inputs["rainfall"]), t32(inputs["cumsum_rainfall"])
    T = rain.shape[1]
    spatial = not (rain.shape[-1] == 1 and rain.shape[-2] == 1)   # (B,T,1,1,1) scalar vs (B,T,1,H,W) spatial
    if spatial:
        rain, cums = rain.reshape(B, T, H, W), cums.reshape(B, T, H, W)
    else:
        rain, cums = rain.reshape(B, T), cums.reshape(B, T)
    return {
        "rain": rain.contiguous(), "cumsum": cums.contiguous(),
        "dem": dem.reshape(B, H, W).contiguous(),
        "imperv": t32(inputs["impervious"]).reshape(B, H, W).contiguous(),
        "manhole": t32(inputs["manhole"]).reshape(B, H, W).contiguous(),
        "dem_min": float(torch.as_tensor(inputs["min_DEM"]).reshape(-1)[0]),
        "dem_max": float(torch.as_tensor(inputs["max_DEM"]).reshape(-1)[0]),
        "B": B, "T": T, "H": H, "W": W,
    }


def preprocess_inputs(t, inputs, device, nums=30, rain_max=6.0, cumsum_rain_max=250.0):
    """Reference contract (Dynamic2DFlood.py:265-320): returns (B, 1, 2*nums+3, H, W) float32 on ``device``,
    assembled by the ``urnn_preprocess_f32`` kernel."""
    ev = inputs if "rain" in inputs else event_to_device(inputs, device)
    x = ops.preprocess(ev["rain"], ev["cumsum"], ev["dem"], ev["imperv"], ev["manhole"], ev["dem_min"], ev["dem_max"],
                       int(t), int(nums), rain_max, cumsum_rain_max)
    return x.unsqueeze(1)
