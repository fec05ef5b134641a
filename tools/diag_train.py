"""Development aid: run SWP training windows at a bench config and report, per window, the loss, the kernel's clip output and
torch's own view of the flat gradient buffer (which parameter tensors hold non-finite or huge values)."""
import argparse
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="location1")
    ap.add_argument("--windows", type=int, default=12)
    ap.add_argument("--seq-num", type=int, default=4)
    ap.add_argument("--graph", type=int, default=0)
    ap.add_argument("--batch", type=int, default=1)
    args = ap.parse_args()
    import urnn_amd.weights as uw
    from urnn_amd.training import Trainer
    dev = torch.device("cuda:0")
    H, W, nums, T, rain_max, cum_max, spatial = bench.CONFIGS[args.config]
    S = args.seq_num
    net, sd, cfg = bench.build_net(H, W, 2 * nums + 3, dev)
    tr = Trainer(net, H, W, nums, rain_max, cum_max, lr=1e-4, grad_clip=1.0, use_graph=bool(args.graph))
    frames = S * args.windows
    ev = uw.make_event(frames, H, W, rain_max, seed=42, spatial_rain=spatial, batch=args.batch)
    g = torch.Generator(device=dev).manual_seed(7)
    label = torch.rand(args.batch, frames, H, W, device=dev, generator=g) ** 3
    label[label < 0.1] = 0
    states = None
    for w in range(args.windows):
        loss, states = tr.train_window(ev, label[:, w * S:(w + 1) * S], w * S, S, states)
        torch.cuda.synchronize()
        clip = tr.last["clip"].cpu().tolist()
        gf = tr.gflat
        tnorm = float(gf.double().norm())
        bad = []
        for n, (off, k, shape) in tr.views.items():
            v = gf[off:off + k]
            fin = bool(torch.isfinite(v).all())
            mx = float(v.abs().max()) if fin else float("nan")
            if not fin or mx > 1e3:
                bad.append((n, "nonfinite" if not fin else f"max {mx:.3e}", int((~torch.isfinite(v)).sum())))
        pfin = bool(torch.isfinite(tr.flat).all())
        sfin = all(bool(torch.isfinite(s).all()) for s in states)
        print(f"window {w}: loss {float(loss[0]):.6f} clip_out {clip} torch_norm {tnorm:.6e} params_finite {pfin} states_finite {sfin} bad {bad[:8]}",
              flush=True)


if __name__ == "__main__":
    main()
