#!/usr/bin/env python
"""bench.py -- flood-map frames/s of the U-RNN rollout hot path on MI355X.

A "step" is one timestep of the rollout = one H x W water-depth frame per event in the batch: per-frame input
assembly + encoder + decoder + head (`ED.forward`, reference model.py:65-121, inside the `Inference` loop of
test.py:326-377), replayed as a captured hipGraph with the six hidden states resident in HBM.

Workload at N=1: BASELINE.json configs[1] -- location1 inference, 500x500 @ 2 m, historical_nums=30 (C=63),
T=360, one event per GPU, fp32, synthetic event + seeded random weights (no datasets/checkpoints offline).
N>1: one process per GPU (torchrun), one independent event per rank, no data-path collective (events are
independent: test.py:741-746) -> weak scaling; value = total frames of all ranks / max-over-ranks time.

Prints ONE JSON line (rank 0).  Extra objects: "roofline" -- the kernel FAMILY with the most time per frame among the cells'
three (candidate GEMMs / gate GEMMs / blend; the candidate GEMMs since round 3), each launch timed with HIP events on its launch
stream: `avg_launch_us` on ONE kernel chain, where an event pair spans the kernel alone (the figure profiles/r05_kernel_stats_
overlap0.txt -- rocprofv3 --kernel-trace --stats of `bench.py --overlap 0` -- must agree with), `live_overlapped` in the
benchmarked schedule of concurrent chains (kernel + what it queued behind: profiles/r05_kernel_stats.txt holds the kernels' own
durations there); every family under "roofline.kernels" -- and "cpu_baseline" (float32 torch ops on CPU tensors at the best of a few thread counts, the C oracle nested under it; bounded sample,
rank 0 at N=1 only).  `python bench.py --gpus N` without a launcher re-runs itself as N ranks (torch.distributed.run).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

CONFIGS = {
    # name: (H, W, historical_nums, T, rain_max, cumsum_max, spatial_rain)
    "location1": (500, 500, 30, 360, 6.0, 250.0, False),   # BASELINE configs[1]
    "lite64": (64, 64, 3, 30, 60.0, 250.0, False),         # BASELINE configs[0] (plumbing size)
    "lite128": (128, 128, 3, 36, 60.0, 250.0, False),
    "futian": (400, 560, 6, 72, 5.0, 100.0, True),
    "ukea": (52, 120, 6, 36, 10.0, 150.0, True),
}

MIXED = ("futian", "ukea")     # BASELINE configs[4]: mixed-resolution events on every rank

PEAK_MFMA_F32_TFLOPS = 157.3   # MI355X fp32 matrix peak (MI355X_MICROARCH.md)
PEAK_MFMA_BF16_TFLOPS = 2517.0  # dense bf16 matrix peak (MI355X_MICROARCH.md: ~2.5 PF; 16x the fp32 matrix rate)
SPLIT_MFMAS = 3                 # v_mfma_f32_32x32x16_f16 per fp32-equivalent 16-k step of the forward k-loop (f16 hi/lo pieces, urnn_gemm.hip SPLIT = 3)
PEAK_HBM_TBS = 8.0


def algorithmic_work(H, W, C):
    """GFLOP and A_stage MB per frame (SURVEY 8d formulas: GEMM MAC = P*K*N per row of 8a; bytes = 4*P*channels)."""
    P1, P2, P4 = H * W, (H // 2) * (W // 2), (H // 4) * (W // 4)
    mac = 0
    mac += P1 * C * 16 + P1 * 64 * 64 + P2 * 96 * 96                          # encoder stage convs
    mac += P1 * 80 * 192 + P2 * 160 * 288 + P4 * 192 * 288                    # encoder cells
    mac += P4 * 288 * 288 + P2 * 288 * 288 + P1 * 224 * 192                   # decoder cells
    mac += P4 * 96 * 96 * 4 + P2 * 96 * 96 * 4 + P1 * 64 * 16                 # deconvs + final conv
    mac += P1 * (5 * 16 * 16 + 2 * 16)                                        # head
    return 2.0 * mac / 1e9


def a_stage_bytes(H, W, C):
    """SURVEY 8d's A_stage: every one of the 13 stages reads its inputs once, writes its outputs once and reads its
    parameters once (bytes per frame, B=1).  1225.7 MB at 500x500, C=63."""
    P1, P2, P4 = H * W, (H // 2) * (W // 2), (H // 4) * (W // 4)
    rd = P1 * (C + 80 + 64 + 224 + 64 + 16) + P2 * (160 + 96 + 288 + 96) + P4 * (192 + 192 + 96)
    wr = P1 * (16 + 64 + 96 + 64 + 16 + 1) + P2 * (64 + 96 + 96 + 96) + P4 * (96 + 96 + 96)
    par = 5 * 2 * 16 * P1                                                      # LayerNorm affines of the head
    par += (C * 16 + 64 * 64 + 96 * 96) + 2 * 96 * 96 * 4 + 64 * 16 + 5 * 256 + 32            # stage convs, deconvs, head convs
    par += 3 * (80 * 64 + 160 * 96 + 192 * 96 + 288 * 96 + 288 * 96 + 224 * 64)             # cells: conv1 (2F) + conv2 (F)
    return 4.0 * (rd + wr + par)


CELLS = {   # the four cells whose kernels are timed: (I, F, skip, plane divisor)
    "enc1": (16, 64, 0, 1), "dec1": (96, 64, 1, 1), "enc2": (64, 96, 0, 2), "dec2": (96, 96, 1, 2),
}


def fused_reset_gate_cells(H, W, B=1):
    """Which of the four timed cells run with the reset gate recomputed inside the candidate kernel (URNN_PHASE_FUSED_R: the gate
    GEMM then writes the F update-gate planes only) -- asked of the library, which decides by the same rule at launch."""
    from urnn_amd._lib import lib
    L = lib()
    return {name: bool(L.urnn_gru_cell_fused_reset_gate_applies(B, I, F, H // div, W // div, skip)) for name, (I, F, skip, div) in CELLS.items()}


def cell_kernel_work(H, W, B=1, fused=None):
    """Algorithmic HBM bytes and FLOP per launch of a cell's three kernels (SURVEY 8d's per-stage accounting: every input plane
    read once, every output plane written once, fp32; weights < 0.3 MB not counted), K = I + [F] + F input channels:
      gates      reads K planes, writes the 2F raw gate planes -- F where the cell runs fused (the reset gate leaves only its statistics)
      candidate  fused (cand_fused_kernel): reads K, writes F (h's second pass comes from L2 / MALL: not counted; FLOP include the
                 recomputed reset gate);  three-pass (conv_gemm_kernel EPI_CAND): reads K - F plain planes + F raw r + F h, writes F
      blend      reads z, c, h (3F), writes h' (F)
    -> {family: {cell: (bytes, flop)}}"""
    fused = fused or {}
    out = {"gates": {}, "candidate": {}, "blend": {}}
    for name, (I, F, skip, div) in CELLS.items():
        P = B * (H // div) * (W // div)
        K = I + (F if skip else 0) + F
        fu = bool(fused.get(name))
        out["gates"][name] = (4.0 * P * (K + (F if fu else 2 * F)), 2.0 * P * 2 * F * K)
        out["candidate"][name] = (4.0 * P * (K + F if fu else K + 2 * F), 2.0 * P * F * K * (2 if fu else 1))
        out["blend"][name] = (4.0 * P * 4 * F, 0.0)
    return out


FAMILY_KERNELS = {
    "candidate": "cand_fused_kernel<2,4,8> (full-resolution cells: W2.[x;e] on f16 pieces + the reset gate recomputed from the same input stream + "
                 "W2[:,h].(r*h) on v_mfma_f32_32x32x2_f32) and cand_gated_kernel<3,4> (half resolution: 64-pixel tiles, one wave per SIMD, hidden rows gated "
                 "on the fly from the stored raw r, group-wise ring); 4 launches per frame: enc1, dec1, enc2, dec2",
    "gates": "conv_gemm_kernel<NB,2,MAP_QUAD16,EPI_GRU1,D,8,SPLIT=3> (gate GEMM z|r: all 2F columns of a 64-pixel tile per wave -- NB = 4 at F = 64, "
             "the z / r halves with NB = 3 at F = 96; two scaled f16 pieces per fp32 operand, 3 x v_mfma_f32_32x32x16_f16 per 16 k); 4 launches per frame",
    "blend": "gru_blend_kernel<V,FIN=true,ITER> (candidate GroupNorm finalize in the prologue, h' = (1 - z) h + z tanh(GN(c))); the same 4 cells",
}


def build_net(H, W, C, dev, seed=0):
    import urnn_amd.weights as uw
    from urnn_amd.net_config import load_net_config
    from urnn_amd.networks import ED, get_network_params
    cfg = load_net_config()
    sd = uw.make_state_dict(H, W, C, seed=seed)
    ep, dp = get_network_params(False, H, W, C, cfg)
    net = ED(False, ep, dp, 0.5, False, H, W)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return net.to(dev).eval(), sd, cfg


def cpu_model():
    """Host CPU model string (BASELINE.md section 4 asks for it beside the core count)."""
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def c_oracle_baseline(sd, cfgname, budget_s=8.0, max_frames=6):
    """The C oracle (oracle/urnn_oracle.c, OpenMP) on the host cores: same synthetic event, first frames of the
    rollout, input assembly included -- frames/s like the reference's Inference timer."""
    from oracle import oracle as orc
    import urnn_amd.weights as uw
    H, W, nums, T, rain_max, cum_max, spatial = CONFIGS[cfgname]
    net = orc.OracleNet(sd)
    ev = uw.make_event(min(T, max_frames + 1), H, W, rain_max, seed=42, spatial_rain=spatial)
    states = orc.zero_states(1, H, W)
    # one untimed warm-up frame (page faults, thread pool)
    x = orc.preprocess_inputs(0, ev, nums, rain_max, cum_max)[:, 0]
    _, states, _ = net.step(x, states)
    n, t0 = 0, time.time()
    while n < max_frames and (time.time() - t0) < budget_s:
        x = orc.preprocess_inputs(n + 1, ev, nums, rain_max, cum_max)[:, 0]
        _, states, _ = net.step(x, states)
        n += 1
    dt = time.time() - t0
    return {"value": n / dt, "unit": "frames/s", "cores": orc.num_threads(), "kind": "port",
            "arith": "plain-C restatement of the reference's ops (oracle/urnn_oracle.c): fp32 storage, fp64 accumulation, OpenMP; "
                     "the reference itself is Python/PyTorch and cannot travel to the GPU box",
            "sample": f"{n} frames of the {H}x{W} C={2*nums+3} rollout after 1 warm-up frame ({dt:.1f} s), "
                      f"C oracle with OpenMP on {orc.num_threads()} threads of {os.cpu_count()} logical CPUs ({cpu_model()})",
            "cpu_model": cpu_model()}


def torch_cpu_baseline(sd, cfgname, threads, budget_s=10.0, max_frames=4):
    """SURVEY 8(d)'s CPU path: the same step in plain float32 torch ops on the host cores (tests/torch_ref.py, the restatement
    that is pinned to the reference's goldens next to the C oracle), input assembly by the oracle's numpy preprocess_inputs."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import torch_ref
    from oracle import oracle as orc
    import urnn_amd.weights as uw
    H, W, nums, T, rain_max, cum_max, spatial = CONFIGS[cfgname]
    prev = torch.get_num_threads()
    torch.set_num_threads(int(threads))
    try:
        p = {k: torch.from_numpy(v) for k, v in sd.items()}
        ev = uw.make_event(min(T, max_frames + 1), H, W, rain_max, seed=42, spatial_rain=spatial)
        st = [torch.from_numpy(s) for s in orc.zero_states(1, H, W)]
        with torch.no_grad():
            x = torch.from_numpy(orc.preprocess_inputs(0, ev, nums, rain_max, cum_max)[:, 0])
            _, _, _, st = torch_ref.step(p, x, st, H, W)            # untimed warm-up frame
            n, t0 = 0, time.time()
            while n < max_frames and (time.time() - t0) < budget_s:
                x = torch.from_numpy(orc.preprocess_inputs(n + 1, ev, nums, rain_max, cum_max)[:, 0])
                _, _, _, st = torch_ref.step(p, x, st, H, W)
                n += 1
            dt = time.time() - t0
    finally:
        torch.set_num_threads(prev)
    return {"value": n / dt, "unit": "frames/s", "cores": int(threads),
            "arith": "plain float32 torch ops on CPU tensors (tests/torch_ref.py: conv2d / group_norm / layer_norm ...), the reference's own arithmetic",
            "sample": f"{n} frames of the {H}x{W} C={2*nums+3} rollout after 1 warm-up frame ({dt:.1f} s), torch {torch.__version__} with {int(threads)} threads"}


def cpu_baseline(sd, cfgname):
    """the CPU path beside the GPU number is the reference's arithmetic -- plain float32 torch ops on CPU tensors
    (tests/torch_ref.py, pinned to the reference's goldens) -- on the box's host cores, at the BEST of a few thread counts (a 128-thread
    EPYC runs this memory-bound step fastest well below its thread count: VERDICT r5 weak item 10); the C oracle's figure is nested
    under it.  About 25 s of CPU work in total."""
    ncpu = os.cpu_count() or 1
    counts = [t for t in (16, 32, 64, 128) if t <= ncpu] or [ncpu]
    runs = []
    for t in counts:
        try:
            runs.append(torch_cpu_baseline(sd, cfgname, t, budget_s=2.5, max_frames=2))
        except Exception as exc:
            runs.append({"value": 0.0, "cores": t, "error": repr(exc)})
    best = max(runs, key=lambda r: r["value"])
    out = dict(best, kind="port", cpu_model=cpu_model(), logical_cpus=ncpu,
               thread_sweep={str(r["cores"]): round(r["value"], 3) for r in runs})
    try:
        out["c_oracle"] = c_oracle_baseline(sd, cfgname)
    except Exception as exc:
        out["c_oracle"] = {"error": repr(exc)}
    return out


def train_roofline(dev, H, W, B, dtype):
    """The dominant kernel of the training step -- the weight-gradient GEMM (wgrad_kernel, urnn_train.hip; 12 launches per timestep
    for the six cells' gate / candidate convolutions) -- against the HBM roof: algorithmic bytes per launch = the dY planes + the X
    planes it contracts over the pixels (fp32), timed live with events on the launch stream, one launch per cell shape."""
    from urnn_amd import ops, train_ops
    try:
        shapes = []                                            # (name, N, segment channels, plane divisor)
        for name, I, F, skip, div in (("enc1", 16, 64, 0, 1), ("enc2", 64, 96, 0, 2), ("enc3", 96, 96, 0, 4), ("dec3", 0, 96, 1, 4),
                                      ("dec2", 96, 96, 1, 2), ("dec1", 96, 64, 1, 1)):
            segs = ([I] if I else []) + ([F] if skip else []) + [F]
            shapes += [(name + " gates", 2 * F, segs, div), (name + " candidate", F, segs, div)]
        gen = torch.Generator(device=dev).manual_seed(11)
        tot_b = tot_t = 0.0
        per = {}
        with ops.matrix_mode(dtype):
            for name, N, segs, div in shapes:
                h, w = H // div, W // div
                dy = torch.randn(B, N, h, w, device=dev, generator=gen)
                xs = [torch.randn(B, c, h, w, device=dev, generator=gen) for c in segs]
                ws = ops.workspace(ops.lib().urnn_weight_gradient_workspace_bytes(B, N, sum(segs), h, w), dev)
                dW, db = train_ops.weight_gradient(dy, xs, scratch=ws)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 10
                torch.cuda.synchronize(dev)
                a.record()
                for _ in range(reps):
                    train_ops.weight_gradient(dy, xs, dW=dW, db=db, scratch=ws)
                b.record()
                torch.cuda.synchronize(dev)
                t = a.elapsed_time(b) / reps / 1e3
                nbytes = 4.0 * B * h * w * (N + sum(segs))
                per[name] = {"us": t * 1e6, "GB/s": nbytes / t / 1e9}
                tot_b += nbytes
                tot_t += t
        achieved = tot_b / tot_t / 1e9
        return {"bound": "hbm", "kernel": "wgrad_kernel<NP> + wgrad_finalize (1x1-conv weight gradient over the pixels: dW = dY . X^T; fp32 operands as three "
                                          "bf16 pieces on v_mfma_f32_32x32x16_bf16 in fp32 mode, one rounded piece in bf16 mode), the 12 cell launches of a timestep",
                "achieved": achieved, "peak": PEAK_HBM_TBS * 1e3, "unit": "GB/s", "frac": achieved / (PEAK_HBM_TBS * 1e3),
                **train_traffic(H, W, B, dtype),
                "bytes_per_timestep": tot_b, "us_per_timestep": tot_t * 1e6, "launches": per}
    except Exception as exc:  # keep the headline number
        return {"error": repr(exc)}


def train_traffic(H, W, B, dtype):
    """Counter traffic per launch of the weight-gradient GEMM (FETCH_SIZE doubled + WRITE_SIZE passes of `bench.py --mode train`,
    tools/collect_profiles.sh -> profiles/pmc_train.json), only when the record belongs to the kernel sources loaded here and to
    this workload."""
    rec, why = load_pmc_record("pmc_train.json")
    if rec is None:
        return {"traffic": None, "traffic_source": why}
    if (H, W, B, dtype) != (500, 500, 1, "fp32") or "wgrad" not in rec.get("families", {}):
        return {"traffic": None, "traffic_source": "profiles/pmc_train.json holds the 500x500, one event, fp32 training step"}
    f = rec["families"]["wgrad"]
    return {"traffic": f["hbm_bytes_per_launch"], "traffic_source": rec.get("source"),
            "traffic_note": f"dispatch-weighted average over the {f['dispatches']} wgrad_kernel launches of the profiled windows (all layers, not only "
                            "the 12 cell launches timed here)",
            "step_hbm_bytes_per_timestep_counters": rec.get("whole_step", {}).get("hbm_bytes_per_frame")}


def bench_train(args, dev, dist, world, rank):
    """SWP training throughput (BASELINE configs 3-4 shape of work, fp32): a step = one training timestep of one event per GPU
    (forward with kept activations, backward through the window, loss; per window one gradient mean over the ranks and one
    clipped Adam step).  See DESIGN.md section 6a."""
    import urnn_amd.weights as uw
    from urnn_amd.training import Trainer
    name = args.config if args.config != "mixed" else "futian"
    H, W, nums, T, rain_max, cum_max, spatial = CONFIGS[name]
    S = args.seq_num
    net, sd, cfg = build_net(H, W, 2 * nums + 3, dev)
    tr = Trainer(net, H, W, nums, rain_max, cum_max, lr=1e-4, grad_clip=1.0, distributed=world > 1,
                 use_graph=not args.no_graph, matrix_mode=args.dtype)      # N > 1: three graphs per window with the gradient mean between them
    nwin_w, nwin = max(1, (args.warmup + S - 1) // S), max(1, (args.steps + S - 1) // S)
    frames = S * (nwin_w + nwin)
    B = args.batch
    ev = uw.make_event(frames, H, W, rain_max, seed=42 + rank, spatial_rain=spatial, batch=B)
    g = torch.Generator(device=dev).manual_seed(7 + rank)
    label = torch.rand(B, frames, H, W, device=dev, generator=g) ** 3
    label[label < 0.1] = 0

    def run(w0, n, states):
        for w in range(w0, w0 + n):
            loss, states = tr.train_window(ev, label[:, w * S:(w + 1) * S], w * S, S, states)
        return loss, states
    loss, states = run(0, nwin_w, None)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    loss, states = run(nwin_w, nwin, states)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    from urnn_amd.distributed import max_over_ranks
    elapsed = max_over_ranks(elapsed, device=dev if args.dist_backend == "nccl" else None)
    steps = nwin * S * B                                            # one step = one training timestep of one event
    gnorm = float(tr.last["clip"][1])
    if not (np.isfinite(gnorm) and np.isfinite(float(loss[0]))):
        sys.exit(f"bench.py --mode train: non-finite loss / gradient norm ({float(loss[0])}, {gnorm})")
    if rank == 0:
        gflop = algorithmic_work(H, W, 2 * nums + 3) * 3.0          # forward + dX + dW
        print(json.dumps({
            "metric": "SWP training timesteps/s (forward + backward + clipped Adam), whole job", "value": steps * world / elapsed,
            "unit": "steps/s", "n_gpus": world, "steps": steps, "warmup": nwin_w * S, "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.dtype == "fp32" else "bf16 GEMM operands (forward, input- and weight-gradient GEMMs), fp32 accumulate / norms / stored gradients / Adam",
            "data": "synthetic",
            "config": {"workload": f"train {name}: {H}x{W} grid, historical_nums={nums}, SWP windows of seq_num={S} (fast mode), "
                                   f"{B} event(s) per GPU, Adam lr 1e-4, grad clip 1.0", "parallelism": f"DDP x{world} (flat-buffer mean all-reduce)"
                       if world > 1 else "single GPU"},
            "gflop_per_step": gflop, "step_mfma_frac": steps / elapsed * gflop / 1e3 / PEAK_MFMA_F32_TFLOPS,
            "loss": float(loss[0]), "grad_norm": gnorm,
            "roofline": train_roofline(dev, H, W, B, args.dtype), "cpu_baseline": None,
            "note": "training path (DESIGN.md 6a); the BASELINE metric is the default --mode infer"}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def bench_strips(args, dev, dist, world, rank):
    """Single-event latency mode: ONE event's grid split over the ranks in horizontal strips; per timestep
    15 all-reduces of a few doubles (GroupNorm / LayerNorm statistics) are the only exchange.  Strong scaling: the total work is
    fixed.  Eager launches (the exchanges sit inside the cells)."""
    import urnn_amd.weights as uw
    from urnn_amd.strips import StripRollout, strip_rows
    name = args.config if args.config != "mixed" else "futian"
    H, W, nums, T, rain_max, cum_max, spatial = CONFIGS[name]
    net, sd, cfg = build_net(H, W, 2 * nums + 3, dev)
    frames = args.warmup + args.steps
    ev = uw.make_event(frames, H, W, rain_max, seed=42, spatial_rain=spatial, batch=args.batch)     # the same event on every rank
    sr = StripRollout(net, H, W, nums, rain_max, cum_max, rank=rank, world=world)
    sr.load_event(ev)
    sr.run(args.warmup)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    strip = sr.run(args.steps, t0=args.warmup)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    from urnn_amd.distributed import max_over_ranks
    elapsed = max_over_ranks(elapsed, device=dev if args.dist_backend == "nccl" else None)
    if rank == 0:
        print(json.dumps({
            "metric": "flood-map frames/s of ONE event split into spatial strips", "value": args.steps * args.batch / elapsed,
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{name}: {H}x{W} grid split into {world} strip(s) of rows {[strip_rows(H, r, world) for r in range(world)]}, "
                                   f"historical_nums={nums}, {args.batch} event(s)", "parallelism": f"spatial strips x{world}, "
                       f"{sr.exchanges // max(1, frames)} statistics all-reduces per timestep", "graph": False},
            "roofline": None, "cpu_baseline": None,
            "note": "latency mode, eager; the BASELINE metric is the default --mode infer (event-parallel)"}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-run this command as N ranks, one per GPU, the way the reference is
    started (`torchrun --nproc_per_node=N`, README.md:415-422).  Fails loudly when the node has fewer than N GPUs."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    share = "--share-gpu" in sys.argv
    if have < n and not share:
        sys.exit(f"bench.py: --gpus {n} needs {n} GPUs, this node shows {have}")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: launching " + " ".join(cmd), file=sys.stderr)
    sys.exit(subprocess.call(cmd, env=env))


def bind_rank_to_cores(local_rank, world, mode):
    """CPU placement of one rank: N graph-replaying processes on a 128-core two-socket host should not migrate across
    each other's cores or NUMA nodes.  ``--bind auto`` (default): bind when the node runs >= 4 ranks; ``on`` / ``off``: force.  Rank r gets
    the r-th of ``world`` equal, contiguous slices of the logical CPUs this process may use (Linux numbers a socket's cores contiguously
    and the GPUs of an MI355X node are split evenly over the sockets, so contiguous slices keep a rank on one node) and sizes torch's
    intra-op pool to it.  Returns the slice (first, last) or None."""
    if mode == "off" or (mode == "auto" and world < 4) or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        cpus = sorted(os.sched_getaffinity(0))
        per = max(1, len(cpus) // max(world, 1))
        mine = cpus[local_rank * per:(local_rank + 1) * per] or cpus
        os.sched_setaffinity(0, mine)
        torch.set_num_threads(max(1, min(len(mine), 16)))
        return (mine[0], mine[-1])
    except OSError:
        return None


def kernel_source_hash():
    """sha256 over the HIP sources + headers the library is built from (u-rnn_amd/build_ext.py): identifies the kernels whatever
    machine compiled them (the .so bytes may differ between two builds of the same sources; its own hash is recorded beside)."""
    import hashlib
    csrc = os.path.join(REPO, "u-rnn_amd", "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h"))) + [os.path.join(REPO, "include", "urnn_hip.h")]
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def load_pmc_record(name="pmc_kernels.json"):
    """profiles/pmc_kernels.json (tools/make_pmc_json.py) if it was measured with THIS build's kernels, else (None, why): the
    record carries the hash of the kernel sources the counters were collected with; a stale file must not pass as a measurement."""
    path = os.path.join(REPO, "profiles", name)
    if not os.path.isfile(path):
        return None, f"no profiles/{name}"
    with open(path) as fh:
        rec = json.load(fh)
    if os.environ.get("URNN_LIB"):
        return None, "URNN_LIB override (tuning variant): counter traffic of the product build not reported"
    now = kernel_source_hash()
    if rec.get("kernel_source_sha256") != now:
        return None, (f"profiles/{name} was collected with kernel sources {str(rec.get('kernel_source_sha256'))[:12]}, this tree is "
                      f"{now[:12]}: counter traffic not reported (re-run tools/collect_profiles.sh)")
    return rec, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=360)
    ap.add_argument("--warmup", type=int, default=36)
    ap.add_argument("--config", default="location1", choices=sorted(CONFIGS) + ["mixed"])
    ap.add_argument("--exp-config", default=None, help="a reference experiment YAML (config.py:55-213 keys: input_height/width, "
                    "historical_nums, rain_max, cumsum_rain_max, duration): the workload is sized from it instead of --config")
    ap.add_argument("--spatial-rain", action="store_true", help="with --exp-config: (T,H,W) rainfall (Futian / UKEA style)")
    ap.add_argument("--batch", type=int, default=1, help="events per GPU (the reference entry points use 1)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of the captured hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-long-run", action="store_true", help="skip the additional 360-step figure of short runs (counter-collection passes)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="control-plane backend for N>1 (nccl = RCCL over xGMI; gloo only for single-GPU dry runs of the N>1 path)")
    ap.add_argument("--bind", default="auto", choices=["auto", "on", "off"],
                    help="bind every rank to its own contiguous slice of the host's CPUs (auto: when >= 4 ranks run on the node)")
    ap.add_argument("--share-gpu", action="store_true", help="dry run: every rank uses cuda:0 (needs --dist-backend gloo)")
    ap.add_argument("--overlap", type=int, default=1,
                    help="1 (default): encoder(t+1) || decoder+head(t) as two concurrent kernel chains; 0: one chain")
    ap.add_argument("--fused-tails", action="store_true",
                    help="infer mode: RolloutEngine(fused_tails=True) -- the end of enc1 / enc2 / dec1 in one launch with the stage conv behind it "
                         "(190 MB per frame less through HBM, 3-4 %% fewer frames/s: off by default, DESIGN.md 4.10)")
    ap.add_argument("--mode", default="infer", choices=["infer", "train", "strips"],
                    help="infer (default, the BASELINE metric): rollout frames/s.  train: SWP training timesteps/s (forward + backward + "
                         "clipped Adam, windows of --seq-num steps; N>1: DDP mean all-reduce of the flat gradient buffer over RCCL).  "
                         "strips: ONE event split into horizontal strips over the ranks (single-event latency, strong scaling)")
    ap.add_argument("--seq-num", type=int, default=4, help="train mode: timesteps per SWP window")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"],
                    help="train mode: GEMM arithmetic (bf16 = BASELINE configs[3]'s variant; inference always runs the fp32-exact path)")
    ap.add_argument("--matrix-mode", default="fp32", choices=["fp32", "fp32_mfma", "fp32_cand"],
                    help="infer mode: fp32 = fp32 operands as f16 pieces on the 16-bit matrix pipe (the product default); fp32_mfma = the exact "
                         "fp32 matrix instructions everywhere (include/urnn_hip.h URNN_MATRIX_FP32_MFMA: slower, tightest long-rollout parity)")
    args = ap.parse_args()

    if args.exp_config:
        from urnn_amd.exp_config import load_exp_config, workload
        name = "yaml:" + os.path.splitext(os.path.basename(args.exp_config))[0]
        CONFIGS[name] = workload(load_exp_config(args.exp_config), args.spatial_rain)
        args.config = name
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus)                      # plain `python bench.py --gpus N`: become N ranks (one per GPU)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    if world > 1 and not args.share_gpu and torch.cuda.device_count() < world:
        sys.exit(f"bench.py: {world} ranks need {world} GPUs, this node shows {torch.cuda.device_count()} "
                 "(one process per GPU; --share-gpu --dist-backend gloo is the single-GPU dry run)")
    dev = torch.device("cuda", 0 if args.share_gpu else local_rank)
    torch.cuda.set_device(dev)
    bound = bind_rank_to_cores(local_rank, world, args.bind)
    dist = None
    if "WORLD_SIZE" in os.environ:                         # launched by torchrun (also with one rank: RCCL is exercised either way)
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI; data path has no collective: barrier + max-reduce only
        else:
            dist.init_process_group("gloo")
        # n_gpus in the result line is the size of a world the backend has CONFIRMED: every rank contributes a one
        ones = torch.ones(1, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(ones)
        if int(ones.item()) != world or dist.get_world_size() != world:
            sys.exit(f"bench.py: all_reduce over the {args.dist_backend} world returned {ones.item()} for WORLD_SIZE={world}")
    if args.gpus != world and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; reporting n_gpus={world}", file=sys.stderr)

    import urnn_amd.weights as uw
    from urnn_amd.rollout import RolloutEngine

    if args.mode == "train":
        return bench_train(args, dev, dist, world, rank)
    if args.mode == "strips":
        return bench_strips(args, dev, dist, world, rank)

    if args.matrix_mode != "fp32":                          # process-wide; the captured graphs keep it
        from urnn_amd import ops as _ops
        from urnn_amd._lib import check as _check, lib as _lib
        _check(_lib().urnn_set_matrix_mode(_ops.MATRIX_MODES[args.matrix_mode]), "urnn_set_matrix_mode")
    # "mixed" = BASELINE configs[4]: Futian + UKEA events alternating on every rank, one engine (and one captured hipGraph)
    # per grid shape; every other config is a single shape
    names = MIXED if args.config == "mixed" else (args.config,)
    B = args.batch
    engines = []
    for i, name in enumerate(names):
        H, W, nums, T, rain_max, cum_max, spatial = CONFIGS[name]
        net_i, sd_i, cfg = build_net(H, W, 2 * nums + 3, dev)
        eng_i = RolloutEngine(net_i, H, W, nums, rain_max, cum_max, batch=B, max_frames=T, spatial_rain=spatial, net_cfg=cfg,
                              use_graph=not args.no_graph, device=dev, overlap=bool(args.overlap), fused_tails=args.fused_tails)
        eng_i.load_event(uw.make_event(T, H, W, rain_max, seed=42 + rank + 100 * i, spatial_rain=spatial, batch=B))
        eng_i.reset()
        engines.append((eng_i, T))
    for e_i, T_i in engines:            # every engine captures its graphs (and packs its weights) here: with several shapes the warm-up steps
        for _ in range(2 if e_i.levels else 1):      # below may never reach the second engine, whose capture would then land in the timed region
            e_i.run(T_i if e_i.levels else min(T_i, 8))      # (small planes: two whole events -- the second captures the event-length graph)
            e_i.reset()
    torch.cuda.synchronize(dev)
    H, W, nums, T, rain_max, cum_max, spatial = CONFIGS[names[0]]
    C = 2 * nums + 3
    eng = engines[0][0]
    sd = sd_i   # state dict of the (single) shape: the CPU baseline runs the same weights

    def run_steps(k):
        """k timesteps; a new event starts (states zeroed, frame counter reset) whenever an event's T frames are done; with
        several shapes the events alternate between the engines."""
        done = 0
        while done < k:
            e, Te = engines[run_steps.which]
            n = min(k - done, Te - run_steps.t)
            e.run(n)
            run_steps.t += n
            done += n
            if run_steps.t == Te:
                e.reset()
                run_steps.t = 0
                run_steps.which = (run_steps.which + 1) % len(engines)
    run_steps.t = 0
    run_steps.which = 0

    # The engines capture a graph of its own for a run length they have seen before (an event length in production; here --warmup / --steps):
    # two untimed dry passes of the very call pattern below, so that the timed region replays what a steady stream of such calls replays
    for _ in range(2):
        run_steps(args.warmup)
        run_steps(args.steps)
        for e_i, _T in engines:
            e_i.reset()
        run_steps.t = 0
        run_steps.which = 0
    torch.cuda.synchronize(dev)

    run_steps(args.warmup)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    run_steps(args.steps)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    from urnn_amd.distributed import max_over_ranks
    elapsed = max_over_ranks(elapsed, device=dev if args.dist_backend == "nccl" else None)

    frames = args.steps * B * world
    fps = frames / elapsed
    long_run = None
    if args.steps < 360 and world == 1 and not args.no_long_run:           # a short timed region (the driver's --steps 20 is ~20 ms): also one full event
        run_steps(0)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        run_steps(360)
        torch.cuda.synchronize(dev)
        long_run = {"steps": 360, "value": 360 * B / (time.perf_counter() - t1), "unit": "frames/s"}
    if len(names) == 1:
        gflop = algorithmic_work(H, W, C)
    else:   # frame-weighted over one cycle of events
        tot = sum(CONFIGS[n][3] for n in names)
        gflop = sum(algorithmic_work(CONFIGS[n][0], CONFIGS[n][1], 2 * CONFIGS[n][2] + 3) * CONFIGS[n][3] for n in names) / tot

    result = {
        "metric": "flood-map frames/s (HxW water-depth grids), whole job",
        "value": fps,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": (f"{args.config}: {H}x{W} grid, historical_nums={nums} (C={C}), T={T}, "
                                f"{B} event(s) per GPU, inference rollout incl. per-frame input assembly") if len(names) == 1 else
                               ("mixed: " + " + ".join(f"{n} {CONFIGS[n][0]}x{CONFIGS[n][1]} T={CONFIGS[n][3]}" for n in names) +
                                f" events alternating, {B} event(s) per GPU, one hipGraph per shape"),
                   "events_per_gpu": B, "parallelism": f"event-parallel x{world} (no collective)", "cpu_binding": (f"cpus {bound[0]}-{bound[1]} per rank" if bound else "none"),
                   "graph": not args.no_graph, "run_length_graphs": (not args.no_graph) and bool(args.overlap),   # (two untimed dry passes of this call pattern ran first: DESIGN.md 4.13)
                   "overlap_chains": bool(args.overlap),
                   "kernel_chains": (4 if eng.levels else 3 if eng._head_own_chain else 2) if args.overlap else 1, "matrix_mode": args.matrix_mode,
                   "fused_tails": bool(args.fused_tails)},
        "long_run": long_run,
        "gflop_per_frame": gflop,
        "step_mfma_frac": fps / world * gflop / 1e3 / PEAK_MFMA_F32_TFLOPS,                          # of the fp32 matrix peak (round 1's pipe)
        "step_mfma_frac_split_pipe": fps / world * gflop / 1e3 / (PEAK_MFMA_BF16_TFLOPS / SPLIT_MFMAS),  # of 2517 / 3 TFLOP/s: the f16 x 3 pipe the GEMMs run on
        # frames/s x A_stage / 8 TB/s -- the unfused per-stage traffic model (algorithmic bytes)
        "step_hbm_frac_a_stage": (fps / world * a_stage_bytes(H, W, C) / (PEAK_HBM_TBS * 1e12)) if len(names) == 1 else None,
        # the same with the bytes the counters saw (FETCH_SIZE / WRITE_SIZE passes of this build, profiles/pmc_kernels.json)
        "step_hbm_frac_measured_traffic": None,
    }
    pmc_rec, pmc_why = load_pmc_record()       # counter figures of a profiled run: only if they belong to the library loaded here
    if pmc_rec is not None and args.config == "location1" and B == 1:
        ws = pmc_rec.get("whole_step")
        if ws:
            result["step_hbm_frac_measured_traffic"] = fps / world * ws["hbm_bytes_per_frame"] / (PEAK_HBM_TBS * 1e12)
            result["step_hbm_bytes_per_frame_counters"] = ws["hbm_bytes_per_frame"]

    if rank == 0:
        # The cells' three kernel families, each launch timed with events on its launch stream.  All are HBM-bound: their arithmetic
        # intensity (<= 40 FLOP/B) sits below the ridge of the pipe they run on (2517 / 3 = 839 TFLOP/s fp32-equivalent over
        # 8 TB/s = 105 FLOP/B).  `roofline` = the family with the most time per frame; the others under roofline.kernels.
        try:
            fused_cells = fused_reset_gate_cells(H, W, B)
            work = cell_kernel_work(H, W, B, fused_cells)
            live = eng.probe_cell_kernels()       # the timed region's scheduling mode (concurrent chains unless --overlap 0)
            if args.overlap:                      # one chain: an event pair spans the kernel alone -- what rocprofv3 calls its duration
                iso_eng = RolloutEngine(eng.net, H, W, nums, rain_max, cum_max, batch=B, max_frames=T, spatial_rain=spatial, net_cfg=cfg,
                                        use_graph=False, device=dev, overlap=False)
                iso_eng.load_event(uw.make_event(T, H, W, rain_max, seed=42 + rank, spatial_rain=spatial, batch=B))
                iso = iso_eng.probe_cell_kernels()
                del iso_eng
            else:
                iso = live
            split = os.environ.get("URNN_TUNE_SPLIT", "1") != "0"
            mfma_peak = PEAK_MFMA_BF16_TFLOPS / SPLIT_MFMAS if split else PEAK_MFMA_F32_TFLOPS
            fams = []
            for fam in ("candidate", "gates", "blend"):
                by = {c: work[fam][c][0] for c in CELLS}
                fl = {c: work[fam][c][1] for c in CELLS}
                us = {c: iso[c][fam] * 1e6 for c in CELLS}
                us_live = {c: live[c][fam] * 1e6 for c in CELLS}
                bpl, avg, avg_live = sum(by.values()) / 4, sum(us.values()) / 4, sum(us_live.values()) / 4
                traffic, tsrc = None, pmc_why
                if pmc_rec is not None and args.config == "location1" and B == 1 and fam in pmc_rec.get("families", {}):
                    traffic, tsrc = pmc_rec["families"][fam]["hbm_bytes_per_launch"], pmc_rec.get("source")
                entry = {"family": fam, "kernel": FAMILY_KERNELS[fam], "bound": "hbm", "launches_per_frame": 4,
                         "achieved": bpl / avg / 1e3, "peak": PEAK_HBM_TBS * 1e3, "unit": "GB/s", "frac": bpl / avg / 1e3 / (PEAK_HBM_TBS * 1e3),
                         "traffic": traffic, "traffic_source": tsrc, "bytes_per_launch": bpl, "avg_launch_us": avg,
                         "avg_launch_us_is": "one kernel chain (an event pair on the launch stream spans the kernel alone, plus ~2 us of record path / launch "
                                             "boundary: 4-6 % above rocprofv3's duration of the same launch): compare with rocprofv3 --kernel-trace --stats of "
                                             "`bench.py --overlap 0` (profiles/*_kernel_stats_overlap0.txt)",
                         "us_per_frame": sum(us.values()), "launch_us": us, "bytes_per_launch_by_cell": by,
                         "live_overlapped": {"chains": (4 if eng.levels else 3 if eng._head_own_chain else 2) if args.overlap else 1, "avg_launch_us": avg_live, "launch_us": us_live, "frac": bpl / avg_live / 1e3 / (PEAK_HBM_TBS * 1e3),
                                            "note": "the benchmarked schedule: kernel + what it queued behind on its stream while the other chain holds the CUs"}
                         if args.overlap else None}
                if sum(fl.values()) > 0:
                    entry["mfma"] = {"flops_per_launch": sum(fl.values()) / 4, "achieved_tflops": sum(fl.values()) / 4 / avg / 1e6,
                                     "peak_tflops_fp32_equivalent": mfma_peak, "frac": sum(fl.values()) / 4 / avg / 1e6 / mfma_peak}
                fams.append(entry)
            top = max(fams, key=lambda e: e["us_per_frame"])
            result["roofline"] = dict(top, reset_gate_recomputed_in_candidate_kernel=fused_cells,
                                      dominant_by="largest sum of launch durations per frame among the cells' kernel families (one chain)",
                                      kernels=fams)
        except Exception as exc:  # keep the headline number even if the side measurement fails
            result["roofline"] = {"error": repr(exc)}
        if world == 1 and not args.no_cpu_baseline and len(names) == 1:
            try:
                result["cpu_baseline"] = cpu_baseline(sd, args.config)
            except Exception as exc:
                result["cpu_baseline"] = {"error": repr(exc)}
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
