"""Development aid: soak the benchmarked schedule -- N whole events (360 frames at 500x500, graph + two chains) must give
bit-identical frames and final states every time (rare ordering hazards between the two kernel chains would show up here)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, urnn_amd.weights as uw
from urnn_amd.rollout import RolloutEngine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
H, W, nums, T, rain_max, cum_max, spatial = bench.CONFIGS["location1"]
dev = torch.device("cuda:0")
net, sd, cfg = bench.build_net(H, W, 63, dev)
eng = RolloutEngine(net, H, W, nums, rain_max, cum_max, max_frames=T, net_cfg=cfg, use_graph=True, device=dev, overlap=True, keep_raw=True)
ev = uw.make_event(T, H, W, rain_max, seed=42)
ref = eng.rollout(ev).clone(); ref_cls = eng.out_cls[:T].clone(); ref_st = [s.clone() for s in eng.final_states()]
bad = 0
for i in range(N):
    out = eng.rollout(ev)
    same = torch.equal(out, ref) and torch.equal(eng.out_cls[:T], ref_cls) and all(torch.equal(a, b) for a, b in zip(eng.final_states(), ref_st))
    bad += 0 if same else 1
    if not same:
        print("event", i, "differs: frames", int((out != ref).sum()), "elements")
print(f"{N} events x {T} frames: {bad} non-identical")
