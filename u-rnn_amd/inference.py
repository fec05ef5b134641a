"""Inference entry points mirrored from the reference's ``test.py``: ``Inference`` (test.py:326-377) and
``load_net`` (test.py:380-408).  Same arguments, same (T,H,W) float32 numpy result and timing print; the T-step
loop runs on the device through ``RolloutEngine`` (one captured hipGraph per timestep)."""
import os
import time

import numpy as np
import torch

from .networks.model import ED
from .rollout import RolloutEngine

# One RolloutEngine (resident states, frame buffers, captured hipGraphs) per (net, grid, history, batch, rain kind): mixed-
# resolution event streams (Futian / UKEA) switch between them without re-capturing.  Least recently used
# engines are dropped beyond MAX_ENGINES (their frame buffers are sized for whole events: ~1.1 GB at 500x500, T=360).
_ENGINES = {}
MAX_ENGINES = 4


def Inference(net, inputs, device, historical_nums=30, rain_max=6.0, cumsum_rain_max=250.0,
              input_height=500, input_width=500, net_cfg=None, use_graph=True):
    with torch.no_grad():
        net.eval()
        rain = inputs["rainfall"]
        Frames = rain.shape[1]
        B = rain.shape[0]
        spatial = not (rain.shape[-1] == 1 and rain.shape[-2] == 1)
        key = (id(net), input_height, input_width, historical_nums, float(rain_max), float(cumsum_rain_max), B, spatial,
               bool(use_graph))
        eng = _ENGINES.pop(key, None)
        if eng is None or eng.Tcap < Frames:
            eng = RolloutEngine(net, input_height, input_width, historical_nums, rain_max, cumsum_rain_max, batch=B,
                                max_frames=Frames, spatial_rain=spatial, net_cfg=net_cfg, use_graph=use_graph,
                                device=device, overlap=True)
        _ENGINES[key] = eng                       # most recently used last
        while len(_ENGINES) > MAX_ENGINES:
            _ENGINES.pop(next(iter(_ENGINES)))
        eng.load_event(inputs)
        eng.reset()
        torch.cuda.synchronize(eng.device)
        test_start_time = time.time()
        eng.run(Frames)
        torch.cuda.synchronize(eng.device)
        test_duration = time.time() - test_start_time
        eng.check_status()                         # non-finite norm statistics (operand range): raise, never return NaN maps
        print(f"Test completed in {test_duration:.2f} sec for {Frames} steps, {test_duration / Frames:.2f} sec/step")
        output_data = eng.out_masked[:Frames, 0].cpu().numpy()
    return np.array(output_data)


def load_net(args, device):
    """Build ``ED`` from ``args`` and load the newest ``checkpoint_{epoch}_{loss}.pth.tar`` of
    ``args.save_model_dir`` (reference checkpoints load unchanged: alias keys and ``module.`` prefixes are
    handled by ``ED.load_state_dict``)."""
    net = ED(args.clstm, args.model_params["encoder_params"], args.model_params["decoder_params"], args.cls_thred,
             args.use_checkpoint, input_height=args.input_height, input_width=args.input_width)
    print("loading model...")
    names = sorted(os.listdir(args.save_model_dir), key=lambda x: int(x.replace("checkpoint_", "").split("_")[0]))
    model_path = os.path.join(args.save_model_dir, names[-1])
    info = torch.load(model_path, map_location=torch.device("cpu"))
    print("loaded model:%s" % model_path)
    net.load_state_dict(info["state_dict"])
    return net.to(device)
