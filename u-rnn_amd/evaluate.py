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


def main(argv=None):
    """``python -m urnn_amd.evaluate --exp_config <reference yaml> --device 0 [--test_list_file ...] [--checkpoint ...]``: the
    evaluation half of the reference's test.py entry (test.py:797-808, config.py:55-213) on the HIP path -- hyper-parameters from
    the experiment YAML's own keys, events from its data_root / test_list_file, metrics printed per event and as mean / std."""
    import argparse
    import json

    import torch

    from .events import Dynamic2DFlood
    from .exp_config import load_exp_config
    from .net_config import load_net_config
    from .networks import ED, get_network_params
    ap = argparse.ArgumentParser(prog="urnn_amd.evaluate")
    ap.add_argument("--exp_config", required=True)
    ap.add_argument("--device", default="0")
    ap.add_argument("--test_list_file", default=None)
    ap.add_argument("--data_root", default=None)
    ap.add_argument("--location", default=None, help="evaluate one catchment only (config.py:121; overrides the YAML's `location`)")
    ap.add_argument("--timestamp", default=None, help="experiment folder name of the reference (only echoed)")
    ap.add_argument("--checkpoint", default=None, help="reference checkpoint (.pth.tar with 'state_dict'); seeded weights if omitted")
    a = ap.parse_args(argv)
    cfg = load_exp_config(a.exp_config, test_list_file=a.test_list_file, data_root=a.data_root, location=a.location)
    dev = torch.device("cuda", int(str(a.device).split(",")[0]))
    H, W, C = cfg["input_height"], cfg["input_width"], 2 * cfg["historical_nums"] + 3
    ep, dp = get_network_params(False, H, W, C, load_net_config())
    net = ED(False, ep, dp, cfg["cls_thred"], False, H, W)
    if a.checkpoint:
        ck = torch.load(a.checkpoint, map_location="cpu")
        net.load_state_dict(ck.get("state_dict", ck))
    else:
        from . import weights as uw
        net.load_state_dict({k: torch.from_numpy(v) for k, v in uw.make_state_dict(H, W, C, seed=0).items()})
    net = net.to(dev).eval()
    ds = Dynamic2DFlood(cfg["data_root"], "test", event_list_file=cfg["test_list_file"] or None, duration=cfg["duration"],
                        location=cfg["location"])
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    metrics, summary, _ = evaluate_events(net, ds, dev, historical_nums=cfg["historical_nums"], rain_max=cfg["rain_max"],
                                          cumsum_rain_max=cfg["cumsum_rain_max"], flood_max=cfg["flood_max"],
                                          flood_thres=cfg["flood_thres"], rank=rank, world_size=world)
    print(json.dumps({"exp_config": a.exp_config, "timestamp": a.timestamp, "rank": rank, "events": metrics, "summary": summary},
                     default=float))


if __name__ == "__main__":
    main()
