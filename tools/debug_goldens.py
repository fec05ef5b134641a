"""Ad-hoc GPU debugging: per-kernel error table against the goldens (not part of the test-suite)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import rel_err
import urnn_amd.weights as uw
from urnn_amd.net_config import load_net_config
from urnn_amd.networks import ED, get_network_params

dev = torch.device("cuda:0")
g = np.load("tests/golden/kernels_16x16.npz")
H, W, C = int(g["H"]), int(g["W"]), int(g["C"])
sd = uw.make_state_dict(H, W, C, seed=int(g["weights_seed"]))
ep, dp = get_network_params(False, H, W, C, load_net_config())
net = ED(False, ep, dp, 0.5, False, H, W)
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net = net.to(dev).eval()
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
def rep(name, got, ref):
    got = got.cpu().numpy()
    print(f"{name:28s} rel {rel_err(got, ref):.3e}  tensor-rel {rel_err(got, ref, 1.0):.3e}  nan {np.isnan(got).sum()}")
for t in ("B1", "B2"):
    for i in (1, 2, 3):
        rep(f"enc stage{i} {t}", getattr(net.encoder, f"stage{i}")(T(g[f"s{i}_in_{t}"])), g[f"s{i}_out_{t}"])
    rep(f"dec stage1 {t}", net.decoder.stage1(T(g[f"dc1_in_{t}"])), g[f"dc1_out_{t}"])
    rep(f"deconv3 {t}", net.decoder.stage3(T(g[f"dc3_in_{t}"])), g[f"dc3_out_{t}"])
    rep(f"deconv2 {t}", net.decoder.stage2(T(g[f"dc2_in_{t}"])), g[f"dc2_out_{t}"])
    for i in (1, 2, 3):
        rnn = getattr(net.encoder, f"rnn{i}")
        rep(f"enc cell{i} {t}", rnn(T(g[f"enc{i}_x_{t}"])[None], T(g[f"enc{i}_h_{t}"]))[0], g[f"enc{i}_out_{t}"])
    for i in (3, 2, 1):
        rnn = getattr(net.decoder, f"rnn{i}")
        st = torch.cat((T(g[f"dec{i}_e_{t}"]), T(g[f"dec{i}_d_{t}"])), 1)
        x = None if i == 3 else T(g[f"dec{i}_x_{t}"])[None]
        rep(f"dec cell{i} {t}", rnn(x, st)[0], g[f"dec{i}_out_{t}"])
    masked, cls, raw = net.head.run(T(g[f"head_in_{t}"]), want_raw=True)
    rep(f"head cls {t}", cls, g[f"head_cls_{t}"]); rep(f"head raw {t}", raw, g[f"head_raw_{t}"])
    res = net(T(g[f"step_x_{t}"]), *[T(g[f"step_state{k}_{t}"]) for k in range(6)])
    for k in range(6):
        rep(f"step state{k} {t}", res[1 + k], g[f"step_newstate{k}_{t}"])
torch.cuda.synchronize()
print("done")
