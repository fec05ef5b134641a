"""Development aid: roll out T frames at a bench config under the current URNN_TUNE_* environment and save the frames / final states,
or compare two saved runs.   python tools/debug_variants.py run out.npz [--overlap 1 --graph 1 --T 8]   |   ... cmp a.npz b.npz"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cmd")
    ap.add_argument("a")
    ap.add_argument("b", nargs="?")
    ap.add_argument("--config", default="location1")
    ap.add_argument("--overlap", type=int, default=1)
    ap.add_argument("--graph", type=int, default=1)
    ap.add_argument("--T", type=int, default=8)
    args = ap.parse_args()
    if args.cmd == "cmp":
        x, y = np.load(args.a), np.load(args.b)
        for k in x.files:
            d = np.abs(x[k].astype(np.float64) - y[k])
            print(f"{k:8s} max|a-b| {d.max():.3e}  rel-to-max {d.max() / max(np.abs(y[k]).max(), 1e-30):.3e}  mismatching {(d > 1e-4).mean():.3e}")
        return
    import urnn_amd.weights as uw
    from urnn_amd.rollout import RolloutEngine
    H, W, nums, T, rain_max, cum_max, spatial = bench.CONFIGS[args.config]
    dev = torch.device("cuda:0")
    net, sd, cfg = bench.build_net(H, W, 2 * nums + 3, dev)
    eng = RolloutEngine(net, H, W, nums, rain_max, cum_max, max_frames=args.T, spatial_rain=spatial, net_cfg=cfg, use_graph=bool(args.graph),
                        device=dev, overlap=bool(args.overlap), keep_raw=True)
    eng.rollout(uw.make_event(args.T, H, W, rain_max, seed=42, spatial_rain=spatial))
    torch.cuda.synchronize()
    out = {"raw": eng.out_raw[:args.T].cpu().numpy(), "cls": eng.out_cls[:args.T].cpu().numpy()}
    for k, s in enumerate(eng.final_states()):
        out[f"state{k}"] = s.cpu().numpy()
    np.savez(args.a, **out)


if __name__ == "__main__":
    main()
