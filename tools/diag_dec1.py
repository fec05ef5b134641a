"""Which kernel of the dec1 cell (500x500: K = 224, F = 64) is not bit-reproducible?  Each phase (gate GEMM | candidate GEMM |
finalize + blend) is launched DIAG_N times on fixed inputs with its predecessors' outputs frozen in the workspace; after every
launch the bytes the phase wrote are compared with the first launch's (development aid)."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import urnn_amd.weights as uw  # noqa: E402
from urnn_amd import ops  # noqa: E402
from urnn_amd.net_config import load_net_config  # noqa: E402
from urnn_amd.networks import ED, get_network_params  # noqa: E402
from urnn_amd.rollout import RolloutEngine  # noqa: E402

H = W = 500
N = int(os.environ.get("DIAG_N", "1500"))
which = os.environ.get("DIAG_CELL", "dec1")
dev = torch.device("cuda:0")
sd = uw.make_state_dict(H, W, 63, seed=0)
ep, dp = get_network_params(False, H, W, 63, load_net_config())
net = ED(False, ep, dp, 0.5, False, H, W)
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net = net.to(dev).eval()
eng = RolloutEngine(net, H, W, 30, 6.0, 250.0, max_frames=4, use_graph=False)
eng.load_event(uw.make_event(4, H, W, 6.0, seed=42))
eng.reset()
eng.run(2)
torch.cuda.synchronize()
e1, e2, e3, d1, d2, d3 = [s.clone() for s in eng.states]
u2, u3, a1, a2 = eng.u2.clone(), eng.u3.clone(), eng.a1.clone(), eng.a2.clone()
cell, x, e, h = {"dec1": (net.decoder.rnn1, u2, e1, d3), "enc1": (net.encoder.rnn1, a1, None, e1), "dec2": (net.decoder.rnn2, u3, e2, d2),
                 "enc2": (net.encoder.rnn2, a2, None, e2)}[which]
F = h.shape[1]
P = h.shape[2] * h.shape[3]
ws = ops.workspace(ops.gru_cell_workspace_bytes(1, F, h.shape[2], h.shape[3]), dev)
out = torch.empty_like(h)
g1_bytes = 2 * F * P * 4
cx_off = (g1_bytes + 255) // 256 * 256
cx_bytes = F * P * 4


def region(name):
    if name == "gates":
        return [("g1", ws[:g1_bytes]), ("rest", ws[cx_off + cx_bytes:])]
    if name == "cand":
        return [("cx", ws[cx_off:cx_off + cx_bytes]), ("rest", ws[cx_off + cx_bytes:])]
    return [("out", out.view(torch.uint8).reshape(-1)), ("rest", ws[cx_off + cx_bytes:])]


for name, mask in (("gates", ops.PHASE_GATES), ("cand", ops.PHASE_CAND), ("blend", ops.PHASE_GN2 | ops.PHASE_BLEND)):
    cell.step(x, e, h, out=out, phases=mask, ws=ws)
    torch.cuda.synchronize()
    ref = [(n, r.clone()) for n, r in region(name)]
    bad = 0
    for i in range(N):
        cell.step(x, e, h, out=out, phases=mask, ws=ws)
        torch.cuda.synchronize()
        for (n, r0), (_, r1) in zip(ref, region(name)):
            if not torch.equal(r0, r1):
                bad += 1
                if bad <= 5:
                    a, b = r0.view(torch.float32), r1.view(torch.float32)
                    idx = torch.nonzero(a != b, as_tuple=True)[0].cpu().numpy()
                    ch, px = idx // P, idx % P
                    d = (a.double() - b.double()).abs()
                    print(f"  {which} {name} launch {i + 1}: region {n}: {len(idx)} floats differ, max |d| {float(d.max()):.3e} (max |v| {float(a.abs().max()):.2e}); "
                          f"channels {sorted(set(ch.tolist()))[:16]}, pixels {px.min()}..{px.max()} ({len(set(px.tolist()))} distinct, {len(set((px // 64).tolist()))} 64-px tiles), first px {sorted(set(px.tolist()))[:8]}")
    print(f"{which} {name}: {bad} of {N} launches differ from the first")
    # leave the first launch's outputs in place for the next phase
    cell.step(x, e, h, out=out, phases=mask, ws=ws)
    torch.cuda.synchronize()

# ---- the whole cell back to back (no host synchronisation between its kernels), fixed workspace: which buffer departs first? ----
M = int(os.environ.get("DIAG_M", "3000"))
part_off = cx_off + cx_bytes
cell.step(x, e, h, out=out, ws=ws)
torch.cuda.synchronize()
ref = {"g1": ws[:g1_bytes].clone(), "cx": ws[cx_off:cx_off + cx_bytes].clone(), "rest": ws[part_off:].clone(), "out": out.clone()}
bad = 0
for i in range(M):
    cell.step(x, e, h, out=out, ws=ws)
    torch.cuda.synchronize()
    cur = {"g1": ws[:g1_bytes], "cx": ws[cx_off:cx_off + cx_bytes], "rest": ws[part_off:], "out": out}
    nan_cx = int(torch.isnan(cur["cx"].view(torch.float32)).sum())
    if nan_cx:
        a = cur["cx"].view(torch.float32)
        idx = torch.nonzero(torch.isnan(a), as_tuple=True)[0].cpu().numpy()
        ch, px = idx // P, idx % P
        print(f"  launch {i + 1}: {nan_cx} NaN in cx: channels {len(set(ch.tolist()))}, px {px.min()}..{px.max()}, distinct px {len(set(px.tolist()))}, px % 128 values {sorted(set((px % 128).tolist()))[:20]}")
    diff = [k for k in ref if not torch.equal(ref[k], cur[k])]
    if diff:
        bad += 1
        if bad <= 6:
            msg = []
            for k in diff:
                a, b = ref[k].reshape(-1).view(torch.float32) if ref[k].dtype == torch.uint8 else ref[k].reshape(-1), None
                b = cur[k].reshape(-1).view(torch.float32) if cur[k].dtype == torch.uint8 else cur[k].reshape(-1)
                idx = torch.nonzero(a != b, as_tuple=True)[0].cpu().numpy()
                if k in ("g1", "cx", "out"):
                    ch, px = idx // P, idx % P
                    msg.append(f"{k}: {len(idx)} floats, max|d| {float((a.double() - b.double()).abs().max()):.2e}, channels {sorted(set(ch.tolist()))[:10]} ({len(set(ch.tolist()))}), "
                               f"px {px.min()}..{px.max()} ({len(set((px // 64).tolist()))} 64-px tiles; first {sorted(set(px.tolist()))[:6]})")
                else:
                    msg.append(f"{k}: {len(idx)} floats differ, offsets {idx[:8].tolist()}")
            print(f"  whole {which} cell, launch {i + 1}: " + " | ".join(msg))
print(f"whole {which} cell back to back: {bad} of {M} differ from the first")
