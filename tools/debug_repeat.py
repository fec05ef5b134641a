"""Development aid: run every launch of the timestep repeatedly on fixed inputs and report launches whose output is not
bit-identical from run to run (races)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from urnn_amd import ops
from urnn_amd.rollout import RolloutEngine
import urnn_amd.weights as uw

H, W, nums, T, rain_max, cum_max, spatial = bench.CONFIGS["location1"]
dev = torch.device("cuda:0")
net, sd, cfg = bench.build_net(H, W, 63, dev)
eng = RolloutEngine(net, H, W, nums, rain_max, cum_max, max_frames=8, net_cfg=cfg, use_graph=False, device=dev)
eng.load_event(uw.make_event(8, H, W, rain_max, seed=42)); eng.reset(); eng.run(2)
torch.cuda.synchronize()
enc, dec = net.encoder, net.decoder
e1, e2, e3, d1, d2, d3 = [s.clone() for s in eng.states]
a1, a2, a3, u3, u2 = eng.a1.clone(), eng.a2.clone(), eng.a3.clone(), eng.u3.clone(), eng.u2.clone()
cases = {
    "enc1 cell": lambda: enc.rnn1.step(a1, None, e1), "enc2 cell": lambda: enc.rnn2.step(a2, None, e2), "enc3 cell": lambda: enc.rnn3.step(a3, None, e3),
    "dec3 cell": lambda: dec.rnn3.step(None, e3, d1), "dec2 cell": lambda: dec.rnn2.step(u3, e2, d2), "dec1 cell": lambda: dec.rnn1.step(u2, e1, d3),
    "stage2": lambda: enc.stage2(e1), "stage3": lambda: enc.stage3(e2), "deconv3": lambda: dec.stage3(d1), "deconv2": lambda: dec.stage2(d2),
    "dec stage1": lambda: dec.stage1(d3),
}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for name, fn in cases.items():
    ref = fn().clone()
    bad = 0
    worst = 0.0
    for _ in range(n):
        out = fn()
        if not torch.equal(out, ref):
            bad += 1
            worst = max(worst, float((out - ref).abs().max()))
    print(f"{name:12s} non-identical runs {bad}/{n}  worst |diff| {worst:.3e}")
