"""Event loop of the reference's ``test()`` (test.py:411-520, without plots / xlsx / TensorRT): events from a
``Dynamic2DFlood`` loader -> ``Inference`` on this rank's GPU -> de-normalised depths (mm) -> ``compute_metrics`` per event.
Multi-GPU is the reference's scheme: every rank takes the ``DistributedSampler(shuffle=False)`` share of the events
(``urnn_amd.distributed.shard_events``); nothing is exchanged on the data path."""
import os

import numpy as np

from .dataset import r_MinMaxScaler
from .inference import Inference
from .metrics import compute_metrics, summarize


def evaluate_events(net, dataset, device, historical_nums=30, rain_max=6.0, cumsum_rain_max=250.0, flood_max=5000.0,
                    flood_thres=150.0, rank=0, world_size=1, net_cfg=None, keep_outputs=False, use_graph=True):
    """Returns ``(metrics, summary, outputs)``: ``metrics[event_name]`` the six scores of test.py:607-675 for every event of
    this rank, ``summary`` their mean / std, ``outputs[event_name]`` the (T,H,W) depth maps in mm when ``keep_outputs``.
    Event names are ``<location>/<event>`` (the reference keys on the event folder name, which collides across
    locations)."""
    metrics, outputs = {}, {}
    for index in dataset.shard(rank, world_size):   # mixed grid shapes: Inference keeps one engine + hipGraph per shape
        inputs, target, event_dir = dataset.batched(int(index))
        H, W = inputs["absolute_DEM"].shape[-2], inputs["absolute_DEM"].shape[-1]
        frames = Inference(net, inputs, device, historical_nums=historical_nums, rain_max=rain_max,
                           cumsum_rain_max=cumsum_rain_max, input_height=H, input_width=W, net_cfg=net_cfg,
                           use_graph=use_graph)
        out_mm = r_MinMaxScaler(frames, max=flood_max, min=0)
        gt = target[0].numpy()
        T = min(out_mm.shape[0], gt.shape[0])
        name = os.path.join(os.path.basename(os.path.dirname(event_dir)), os.path.basename(event_dir))
        metrics[name] = compute_metrics(out_mm[:T], gt[:T], flood_thres=flood_thres)
        if keep_outputs:
            outputs[name] = np.asarray(out_mm)
    return metrics, (summarize(metrics) if metrics else None), outputs
