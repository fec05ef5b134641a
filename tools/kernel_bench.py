#!/usr/bin/env python
"""Per-kernel timing table at a real workload size (default 500x500, C=63): every launch of the timestep timed
alone with events on the launch stream, next to its fp32-MFMA and HBM floors.  Development tool (GPU box)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from urnn_amd import ops
from urnn_amd.rollout import RolloutEngine
import urnn_amd.weights as uw


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) * 1e3 / iters   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="location1")
    ap.add_argument("--batch", type=int, default=1)
    args = ap.parse_args()
    H, W, nums, T, rain_max, cum_max, spatial = bench.CONFIGS[args.config]
    C, B = 2 * nums + 3, args.batch
    dev = torch.device("cuda:0")
    net, sd, cfg = bench.build_net(H, W, C, dev)
    eng = RolloutEngine(net, H, W, nums, rain_max, cum_max, batch=B, max_frames=8, spatial_rain=spatial, net_cfg=cfg,
                        use_graph=False, device=dev)
    eng.load_event(uw.make_event(8, H, W, rain_max, seed=42, spatial_rain=spatial, batch=B))
    eng.reset()
    eng.run(2)   # realistic (non-zero) state contents
    torch.cuda.synchronize()
    enc, dec = net.encoder, net.decoder
    e1, e2, e3, d1, d2, d3 = eng.states
    P1, P2, P4 = B * H * W, B * (H // 2) * (W // 2), B * (H // 4) * (W // 4)
    MB = 1e6 / 4  # floats per MB
    rows = []

    def add(name, fn, mac, mbytes):
        us = timeit(fn)
        rows.append((name, us, 2 * mac / 419.5e12 * 1e6, mbytes * 1e6 / 8e12 * 1e6))

    add("preprocess", lambda: ops.preprocess(eng.rain, eng.cumsum, eng.dem, eng.imperv, eng.manhole, 0., 1., 3, nums, rain_max,
                                             cum_max, out=eng.x_in), 0, P1 * C * 4 / 1e6)
    add("enc stage1 conv", lambda: enc.stage1(eng.x_in, out=eng.a1), P1 * C * 16, P1 * (C + 16) * 4 / 1e6)
    add("enc stage2 conv+pool", lambda: enc.stage2(e1, out=eng.a2), P1 * 64 * 64, (P1 * 64 + P2 * 64) * 4 / 1e6)
    add("enc stage3 conv+pool", lambda: enc.stage3(e2, out=eng.a3), P2 * 96 * 96, (P2 * 96 + P4 * 96) * 4 / 1e6)
    add("deconv3 (P4->P2)", lambda: dec.stage3(d1, out=eng.u3), P4 * 96 * 384, (P4 * 96 + P2 * 96) * 4 / 1e6)
    add("deconv2 (P2->P1)", lambda: dec.stage2(d2, out=eng.u2), P2 * 96 * 384, (P2 * 96 + P1 * 96) * 4 / 1e6)
    add("dec stage1 conv", lambda: dec.stage1(d3, out=eng.feat), P1 * 64 * 16, P1 * 80 * 4 / 1e6)
    cells = [("enc1", enc.rnn1, eng.a1, None, e1, P1, 16, 64, 0), ("enc2", enc.rnn2, eng.a2, None, e2, P2, 64, 96, 0),
             ("enc3", enc.rnn3, eng.a3, None, e3, P4, 96, 96, 0), ("dec3", dec.rnn3, None, e3, d1, P4, 0, 96, 1),
             ("dec2", dec.rnn2, eng.u3, e2, d2, P2, 96, 96, 1), ("dec1", dec.rnn1, eng.u2, e1, d3, P1, 96, 64, 1)]
    for name, cell, x, e, h, P, I, F, skip in cells:
        Kx = I + (F if skip else 0)
        tmp = h.clone()
        ws = ops.workspace(ops.gru_cell_workspace_bytes(h.shape[0], F, h.shape[2], h.shape[3]), h.device)   # ONE scratch per cell: the later phases read what the gate phase wrote
        add(f"{name} gates GEMM", lambda: cell.step(x, e, h, out=tmp, phases=ops.PHASE_GATES, ws=ws),
            P * 2 * F * (Kx + F), P * (Kx + F + 2 * F) * 4 / 1e6)
        add(f"{name} cand GEMM (+GN1)", lambda: cell.step(x, e, h, out=tmp, phases=ops.PHASE_CAND, ws=ws), P * F * (Kx + F),
            P * (Kx + 2 * F + F) * 4 / 1e6)
        add(f"{name} GN2 finalize", lambda: cell.step(x, e, h, out=tmp, phases=ops.PHASE_GN2, ws=ws), 0, 0)
        add(f"{name} blend", lambda: cell.step(x, e, h, out=tmp, phases=ops.PHASE_BLEND, ws=ws), 0, P * 4 * F * 4 / 1e6)
        add(f"{name} whole cell", lambda: cell.step(x, e, h, out=tmp, ws=ws), P * 3 * F * (Kx + F), P * (Kx + 2 * F) * 4 / 1e6)
    add("head (7 launches)", lambda: net.head.run(eng.feat, out_masked=eng.out_masked, out_cls=eng.out_cls),
        P1 * (5 * 256 + 32), (P1 * 16 * 11 + 2 * P1) * 4 / 1e6)
    def whole():
        eng.t_dev.zero_()
        eng._step()
    add("whole step (eager)", whole, bench.algorithmic_work(H, W, C) * 1e9 / 2 * B, 1225.7 * B if H == 500 else 0)
    eng.t_dev.zero_()
    tot = 0.0
    print(f"{'kernel':28s} {'us':>9s} {'mfma_floor':>11s} {'hbm_floor':>10s} {'frac_of_binding_floor':>22s}")
    for name, us, fm, fh in rows:
        floor = max(fm, fh)
        print(f"{name:28s} {us:9.1f} {fm:11.1f} {fh:10.1f} {floor / us if us > 0 else 0:22.2f}")
    print("sum of parts (us):", sum(r[1] for r in rows if 'whole' not in r[0]))


if __name__ == "__main__":
    main()
