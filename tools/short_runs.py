import sys, time, torch
sys.path.insert(0, '.')
import bench, urnn_amd.weights as uw
from urnn_amd.rollout import RolloutEngine
H, W, nums, T, rain_max, cum_max, spatial = bench.CONFIGS["location1"]
dev = torch.device("cuda:0")
net, sd, cfg = bench.build_net(H, W, 63, dev)
eng = RolloutEngine(net, H, W, nums, rain_max, cum_max, max_frames=T, net_cfg=cfg, use_graph=True, device=dev, overlap=True)
eng.load_event(uw.make_event(T, H, W, rain_max, seed=42)); eng.reset()
eng.run(5); torch.cuda.synchronize()
for k in (1, 2, 5, 10, 20, 40, 80, 20, 20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.run(k)
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"k={k:3d} total {1e3*(t2-t0):7.3f} ms  per frame {1e3*(t2-t0)/k:6.3f}  host enqueue {1e3*(t1-t0):7.3f} ms")
