#!/usr/bin/env python
"""HIP rollout against the committed REFERENCE trace (tests/golden/reference_trace_*.npz: the reference's own modules in float32 and
float64 over the whole event), frame by frame, under both floors, for one or more matrix modes:
    python tools/parity_reference_trace.py [trace.npz] [fp32 fp32_mfma fp32_cand ...]
Evidence tool for profiles/r06_parity_reference_trace.txt (tests/test_hip_rollout.py::test_whole_event_vs_reference_trace asserts)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch


def sub_err(got, want, plane_max, floor):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return float((np.abs(got - want) / np.maximum(np.abs(want), floor * max(float(plane_max), 1e-30))).max())


def main():
    import urnn_amd.weights as uw
    from urnn_amd import ops
    from urnn_amd.rollout import RolloutEngine
    from test_hip_rollout import make_net
    args = sys.argv[1:]
    trace = args[0] if args and args[0].endswith(".npz") else "reference_trace_500x500_T360.npz"
    modes = [a for a in args if not a.endswith(".npz")] or ["fp32", "fp32_mfma"]
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", trace))
    dev = torch.device("cuda:0")
    H, W, nums, T = int(g["H"]), int(g["W"]), int(g["nums"]), int(g["T"])
    rain_max, cum_max, spatial = float(g["rain_max"]), float(g["cumsum_max"]), bool(int(g["spatial"]))
    net, sd = make_net(H, W, 2 * nums + 3, int(g["weights_seed"]), dev)
    ev = uw.make_event(T, H, W, rain_max, seed=int(g["event_seed"]), spatial_rain=spatial)
    eng = RolloutEngine(net, H, W, nums, rain_max, cum_max, max_frames=T, spatial_rain=spatial, keep_raw=True, overlap=True, use_graph=True)
    fr = g["frames"]
    fr_d = torch.from_numpy(fr.astype(np.int64)).to(dev)
    idx = np.concatenate([np.broadcast_to(g["pixels"].astype(np.int64), (len(fr), g["pixels"].size)), g["adv_idx"].astype(np.int64)], axis=1)
    idx_d = torch.from_numpy(np.ascontiguousarray(idx)).to(dev)
    r64 = np.concatenate([g["r64_raw"], g["a64_raw"]], axis=1)
    c64 = np.concatenate([g["r64_cls"], g["a64_cls"]], axis=1)
    r32 = np.concatenate([g["r32_raw"], g["a32_raw"]], axis=1)
    c32 = np.concatenate([g["r32_cls"], g["a32_cls"]], axis=1)
    for mode in modes:
        with ops.matrix_mode(mode):
            eng.rollout(ev)
        raw = torch.gather(eng.out_raw[:T, 0].reshape(T, -1)[fr_d], 1, idx_d).cpu().numpy()
        cls = torch.gather(eng.out_cls[:T, 0].reshape(T, -1)[fr_d], 1, idx_d).cpu().numpy()
        print(f"\n=== {trace}  matrix mode {mode}  ({len(fr)} of {T} frames x {idx.shape[1]} pixels: 4096 random + 1024 where ref32 is furthest from ref64 + 1024 nearest the threshold)")
        print("frame |  floor 0.1 x max: reg HIP-ref64  ref32-ref64  HIP-ref32 | cls HIP-ref64 ref32-ref64 | STRICT 1e-3 x max: reg HIP-ref64 ref32-ref64 | ratio(0.1) ratio(strict)")
        rows = []
        for i, t in enumerate(fr):
            rm, cm = g["ref64_raw_plane_max"][i], g["ref64_cls_plane_max"][i]
            a = [sub_err(raw[i], r64[i], rm, 0.1), sub_err(r32[i], r64[i], rm, 0.1), sub_err(raw[i], r32[i], rm, 0.1), sub_err(cls[i], c64[i], cm, 0.1), sub_err(c32[i], c64[i], cm, 0.1),
                 sub_err(raw[i], r64[i], rm, 1e-3), sub_err(r32[i], r64[i], rm, 1e-3)]
            rows.append([int(t)] + a + [a[0] / max(1e-4, 1.5 * a[1]), a[5] / max(1e-4, 1.5 * a[6])])
        rows = np.asarray(rows)
        if os.environ.get("PARITY_DUMP"):
            np.save(os.path.join(os.environ["PARITY_DUMP"], f"rows_{os.path.splitext(trace)[0]}_{mode}.npy"), rows)
        for r in rows:
            if int(r[0]) % 12 == 0 or r[8] > 1.0 or r[9] > 1.0:
                print("%5d | %26.2e %12.2e %10.2e | %13.2e %11.2e | %30.2e %11.2e | %9.2f %13.2f" % tuple(r))
        print(f"worst frame: reg HIP-ref64 {rows[:, 1].max():.2e} (frame {int(rows[rows[:, 1].argmax(), 0])}), ref32-ref64 {rows[:, 2].max():.2e} (frame {int(rows[rows[:, 2].argmax(), 0])}); "
              f"strict: HIP {rows[:, 6].max():.2e} (frame {int(rows[rows[:, 6].argmax(), 0])}), ref32 {rows[:, 7].max():.2e} (frame {int(rows[rows[:, 7].argmax(), 0])})")
        print(f"mean over frames: reg HIP-ref64 {rows[:, 1].mean():.2e}, ref32-ref64 {rows[:, 2].mean():.2e}; strict {rows[:, 6].mean():.2e} / {rows[:, 7].mean():.2e}; "
              f"median {np.median(rows[:, 1]):.2e} / {np.median(rows[:, 2]):.2e}; frames where HIP is closer to ref64 than ref32 is: {int((rows[:, 1] < rows[:, 2]).sum())} of {len(rows)} "
              f"(strict: {int((rows[:, 6] < rows[:, 7]).sum())})")
        print(f"per-frame bar max(1e-4, 1.5 x ref32-ref64): frames over it: {int((rows[:, 8] > 1).sum())} (floor 0.1), {int((rows[:, 9] > 1).sum())} (strict); worst ratio {rows[:, 8].max():.2f} / {rows[:, 9].max():.2f}")


if __name__ == "__main__":
    main()
