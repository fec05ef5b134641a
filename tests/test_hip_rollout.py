"""GPU rollout parity: the on-device engine (hipGraph-captured timestep, in-place states, device frame counter)
against the reference-generated rollout goldens, the float64 reference, and the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from conftest import assert_close, masked_parity, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def make_net(H, W, C, seed, dev):
    import urnn_amd.weights as uw
    from urnn_amd.net_config import load_net_config
    from urnn_amd.networks import ED, get_network_params
    sd = uw.make_state_dict(H, W, C, seed=seed)
    ep, dp = get_network_params(False, H, W, C, load_net_config())
    net = ED(False, ep, dp, 0.5, False, H, W)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return net.to(dev).eval(), sd


@pytest.mark.parametrize("name", ["rollout_64x64_T30.npz", "rollout_24x40_T8_spatial.npz"])
@pytest.mark.parametrize("use_graph", [True, False])
def test_rollout_vs_reference(golden, dev, name, use_graph):
    import urnn_amd.weights as uw
    from urnn_amd.rollout import RolloutEngine
    g = golden(name)
    H, W, nums, T = int(g["H"]), int(g["W"]), int(g["nums"]), int(g["T"])
    spatial = bool(int(g["spatial"]))
    net, _ = make_net(H, W, 2 * nums + 3, int(g["weights_seed"]), dev)
    ev = uw.make_event(T, H, W, float(g["rain_max"]), seed=int(g["event_seed"]), spatial_rain=spatial)
    eng = RolloutEngine(net, H, W, nums, float(g["rain_max"]), float(g["cumsum_max"]), max_frames=T, spatial_rain=spatial,
                        use_graph=use_graph, keep_raw=True)
    frames = eng.rollout(ev)
    torch.cuda.synchronize()
    every = int(g["every"])
    got = frames[::every, 0].cpu().numpy()
    cls = eng.out_cls[:T:every, 0].cpu().numpy()
    raw = eng.out_raw[:T:every, 0].cpu().numpy()
    tol = 1e-4
    print(f"{name} graph={use_graph}: cls {rel_err(cls, g['cls']):.2e} (floor 0.1 max) / {rel_err(cls, g['cls'], 1e-3):.2e} (strict 1e-3 max); "
          f"pre-mask reg {rel_err(raw, g['raw']):.2e} / {rel_err(raw, g['raw'], 1e-3):.2e}; states " +
          ", ".join(f"{rel_err(eng.states[k].cpu().numpy(), g[f'final_state{k}']):.1e}/{rel_err(eng.states[k].cpu().numpy(), g[f'final_state{k}'], 1e-3):.1e}" for k in range(6)))
    assert_close(cls, g["cls"], tol, "cls over rollout")
    assert_close(raw, g["raw"], tol, "pre-mask reg over rollout")
    for k in range(6):
        assert_close(eng.states[k].cpu().numpy(), g[f"final_state{k}"], tol, f"final state {k}")
    nflip = masked_parity(got, g["reg"], g["cls"], g["raw"], tol)
    assert nflip < 0.001 * got.size
    # no further from the exact (float64) rollout than the fp32 reference itself is (x3 slack), strict floor
    for k in range(6):
        truth = g[f"final_state64_{k}"]
        e_hip = rel_err(eng.states[k].cpu().numpy(), truth, 1e-3)
        e_ref = rel_err(g[f"final_state{k}"], truth, 1e-3)
        assert e_hip <= max(1e-4, 3.0 * e_ref), f"state {k}: HIP {e_hip:.2e} vs reference-fp32 {e_ref:.2e} (vs fp64 truth)"


def test_inference_entry_matches_reference(golden, dev):
    """Inference() mirror: same (T,H,W) float32 contract and values as the reference's test.Inference."""
    import urnn_amd.weights as uw
    from urnn_amd.inference import Inference
    g = golden("inference_entry_16x16_T6.npz")
    H, W, nums, T = int(g["H"]), int(g["W"]), int(g["nums"]), int(g["T"])
    net, _ = make_net(H, W, 2 * nums + 3, int(g["weights_seed"]), dev)
    ev = uw.make_event(T, H, W, float(g["rain_max"]), seed=int(g["event_seed"]))
    out = Inference(net, {k: torch.from_numpy(np.asarray(v)) for k, v in ev.items()}, dev, historical_nums=nums,
                    rain_max=float(g["rain_max"]), cumsum_rain_max=float(g["cumsum_max"]), input_height=H, input_width=W)
    assert out.shape == g["out"].shape and out.dtype == np.float32
    # every pixel whose reference class map is not within 1e-5 of the threshold (the goldens hold the reference's cls / pre-mask
    # regression of the same test.Inference call, make_golden.py head_taps)
    excluded = masked_parity(out, g["out"], g["cls"], g["raw"], 1e-4)
    assert excluded <= 2


def test_graph_replay_is_deterministic(dev):
    import urnn_amd.weights as uw
    from urnn_amd.rollout import RolloutEngine
    H, W, nums, T = 32, 48, 3, 10
    net, _ = make_net(H, W, 9, 3, dev)
    ev = uw.make_event(T, H, W, 60.0, seed=1)
    eng = RolloutEngine(net, H, W, nums, 60.0, 250.0, max_frames=T)
    a = eng.rollout(ev).clone()
    b = eng.rollout(ev).clone()
    assert torch.equal(a, b), "fixed-order GroupNorm/LayerNorm reductions must make rollouts bit-reproducible"
    eng2 = RolloutEngine(net, H, W, nums, 60.0, 250.0, max_frames=T, use_graph=False)
    c = eng2.rollout(ev)
    assert torch.equal(a, c), "graph replay and eager launches must agree bit for bit"


def test_overlapped_chains_match_sequential(dev):
    """Encoder(t+1) || decoder+head(t) on two streams (graph with forked branches) is a re-scheduling only: the
    frames and the final states must equal the sequential engine's bit for bit."""
    import urnn_amd.weights as uw
    from urnn_amd.rollout import RolloutEngine
    H, W, nums, T = 32, 48, 3, 9
    net, _ = make_net(H, W, 9, 3, dev)
    ev = uw.make_event(T, H, W, 60.0, seed=1)
    seq = RolloutEngine(net, H, W, nums, 60.0, 250.0, max_frames=T)
    a = seq.rollout(ev).clone()
    for use_graph, levels in ((False, False), (True, False), (False, True), (True, True)):      # three chains | the level pipeline (four, a frame apart)
        ovl = RolloutEngine(net, H, W, nums, 60.0, 250.0, max_frames=T, overlap=True, use_graph=use_graph, levels=levels)
        assert ovl.levels == levels
        b = ovl.rollout(ev)
        torch.cuda.synchronize()
        assert torch.equal(a, b)
        for x, y in zip(seq.final_states(), ovl.final_states()):
            assert torch.equal(x, y)
        b2 = ovl.rollout(ev)          # second event through the same captured graphs
        assert torch.equal(a, b2)


@pytest.mark.parametrize("group", ["0", "2", "4", "6"])
def test_grouped_iterations_and_piecewise_runs_match_sequential(dev, monkeypatch, group):
    """The overlapped graphs hold GROUP steady-state iterations per replay and the frame counters are pairs of words advanced by the
    head / input-assembly launches themselves (urnn_*_rollout_f32): whatever the group size, with the head on a chain of its own or
    in front of the encoder, and however an event is cut into run() calls (each call starts without a pending head and ends by
    flushing one), frames and final states equal the one-chain engine's bit for bit."""
    import urnn_amd.weights as uw
    from urnn_amd.rollout import RolloutEngine
    monkeypatch.setenv("URNN_TUNING", "1")          # (the Python host reads URNN_TUNE_* only under this switch)
    monkeypatch.setenv("URNN_TUNE_GROUP", group)
    monkeypatch.setenv("URNN_TUNE_HEAD_CHAIN", "0" if group == "2" else "1")
    H, W, nums, T = 32, 48, 3, 23
    net, _ = make_net(H, W, 9, 3, dev)
    ev = uw.make_event(T, H, W, 60.0, seed=2)
    seq = RolloutEngine(net, H, W, nums, 60.0, 250.0, max_frames=T)
    a = seq.rollout(ev).clone()
    ovl = RolloutEngine(net, H, W, nums, 60.0, 250.0, max_frames=T, overlap=True, levels=False)      # (the three-chain schedule on a small plane)
    for pieces in ((23,), (3, 7, 1, 12), (1, 1, 2, 5, 6, 8), (10, 13)):
        ovl.load_event(ev)
        ovl.reset()
        ovl.out_masked.zero_()
        for n in pieces:
            ovl.run(n)
        ovl.check_status()
        assert int(ovl.t2[T % 2]) == T and int(ovl.te2[(T + 1) % 2]) == T + 1       # the words the next head / encoder pass would read
        assert torch.equal(a, ovl.out_masked[:T]), f"GROUP={group}, run() calls of {pieces} frames"
        for x, y in zip(seq.final_states(), ovl.final_states()):
            assert torch.equal(x, y)
    assert ovl._group == int(group) and ovl._head_own_chain == (group != "2")
    # graphs dropped in the middle of an event (a weight update between run() calls): the re-capture warms up on real frames -- none
    # of the finished ones may change, whatever the parity of the frame it happens at
    for cut in (6, 9, 22):
        ovl.load_event(ev)
        ovl.reset()
        ovl.run(cut)
        ovl._graphs2 = None
        ovl.run(T - cut)
        ovl.check_status()
        assert torch.equal(a, ovl.out_masked[:T]), f"re-capture after {cut} frames"


def test_three_chain_run_lengths_seen_again_replay_as_chunk_graphs(dev, monkeypatch):
    """The three-chain schedule (the big planes'; here forced on a small one): a run length that comes again from the same frame parity
    is captured as one graph per <= CHUNK_FRAMES frames (first iteration, steady iterations and trailing head inside) and replayed from
    then on -- same launches in the same order: frames and states stay bit-identical to the one-chain engine, for one chunk and for
    several (first / middle / last), from frame zero and in mid-event, and across an eviction of the oldest graphs."""
    import urnn_amd.weights as uw
    from urnn_amd.rollout import RolloutEngine
    H, W, nums, T = 32, 48, 3, 30
    net, _ = make_net(H, W, 9, 3, dev)
    ev = uw.make_event(T, H, W, 60.0, seed=6)
    seq = RolloutEngine(net, H, W, nums, 60.0, 250.0, max_frames=T)
    a = seq.rollout(ev).clone()
    want = [s.clone() for s in seq.final_states()]
    for chunk in (120, 8):
        monkeypatch.setattr(RolloutEngine, "CHUNK_FRAMES", chunk)
        ovl = RolloutEngine(net, H, W, nums, 60.0, 250.0, max_frames=T, overlap=True, levels=False)
        for rep in range(3):
            assert torch.equal(a, ovl.rollout(ev)), f"CHUNK_FRAMES={chunk}, rollout {rep}"
            assert all(torch.equal(x, y) for x, y in zip(want, ovl.final_states()))
        chunks = [k for k in ovl._graphs2 if isinstance(k, tuple) and k[0] == "chunk"]
        assert len(chunks) == (1 if chunk == 120 else 4) and all(k[1] == (k[3] is True) for k in chunks)      # (30 frames = 8 + 8 + 7 + 7)
        for rep in range(3):                            # the same cut into two run() calls, the second from an odd / even frame
            for cut in (13, 16):
                ovl.load_event(ev)
                ovl.reset()
                ovl.out_masked.zero_()
                ovl.run(cut)
                ovl.run(T - cut)
                ovl.check_status()
                assert torch.equal(a, ovl.out_masked[:T]), f"CHUNK_FRAMES={chunk}, run({cut}) + run({T - cut}), repetition {rep}"
        assert len([k for k in ovl._graphs2 if isinstance(k, tuple) and k[0] == "chunk"]) <= ovl.CHUNK_GRAPHS
        assert torch.equal(a, ovl.rollout(ev))


@pytest.mark.parametrize("plan", ["default", "F", "A", "B"])
@pytest.mark.parametrize("shape", [(32, 48, 1), (64, 64, 1), (24, 40, 2), (320, 384, 1)])
@pytest.mark.parametrize("use_graph", [True, False])
def test_level_pipeline_and_piecewise_runs_match_sequential(dev, monkeypatch, shape, use_graph, plan):
    """The level pipeline (RolloutEngine(levels=True), the default on small planes with overlap=True): four or five units on four streams, cut by
    level of the network, each a frame behind the one that feeds it, the encoder states in rings of six or ten buffers.  A re-scheduling only -- frames and
    final states equal the one-chain engine's bit for bit however an event is cut into run() calls (each call fills and drains the pipeline;
    calls shorter than four frames run eagerly), through the captured graphs of every frame % 6, and across a re-capture in mid-event."""
    import urnn_amd.weights as uw
    from urnn_amd.rollout import RolloutEngine
    if plan != "default":                           # (LEVEL_PLANS: F = forward hand-overs, no barrier inside a replay; A / B = a barrier per iteration)
        monkeypatch.setenv("URNN_TUNING", "1")
        monkeypatch.setenv("URNN_TUNE_LEVEL_PLAN", plan)
    H, W, B = shape
    nums, T = 3, 23
    net, _ = make_net(H, W, 9, 3, dev)
    ev = uw.make_event(T, H, W, 60.0, seed=2, batch=B)
    # (the reference schedule: one chain -- except on the plane where a one-chain engine takes the four-tiles-per-block cooperative cells,
    # which agree with the three kernels to 2e-6, not bit for bit: there the three-chain schedule, which launches what the pipeline launches)
    seq = RolloutEngine(net, H, W, nums, 60.0, 250.0, max_frames=T, batch=B, overlap=H * W > 65536, levels=False)
    a = seq.rollout(ev).clone()
    ovl = RolloutEngine(net, H, W, nums, 60.0, 250.0, max_frames=T, batch=B, overlap=True, use_graph=use_graph)
    assert ovl.levels and len(ovl._side) == 4
    for pieces in ((23,), (3, 7, 1, 12), (6, 6, 11), (1, 1, 2, 5, 6, 8), (10, 13), (7, 8, 8)):
        ovl.load_event(ev)
        ovl.reset()
        ovl.out_masked.zero_()
        for n in pieces:
            ovl.run(n)
        ovl.check_status()
        assert int(ovl.t2[T % 2]) == T and int(ovl.te2[T % 2]) == T                 # the words the next head / input assembly would read
        assert torch.equal(a, ovl.out_masked[:T]), f"run() calls of {pieces} frames"
        for x, y in zip(seq.final_states(), ovl.final_states()):
            assert torch.equal(x, y)
    if use_graph:
        assert sorted({k[0] for k in ovl._graphs2} - {"run"}) == ["drain", "fill", "group", "steady"]
        assert len([k for k in ovl._graphs2 if k[0] != "run"]) == 4 * ovl._lvP
        for cut in (6, 9, 16):
            ovl.load_event(ev)
            ovl.reset()
            ovl.run(cut)
            ovl._graphs2 = None
            ovl.run(T - cut)
            ovl.check_status()
            assert torch.equal(a, ovl.out_masked[:T]), f"re-capture after {cut} frames"
    assert torch.equal(a, ovl.rollout(ev))
    if use_graph:
        # a run length that comes again from the same frame phase gets a graph of its own (captured the second time, replayed from then on)
        for _ in range(3):
            assert torch.equal(a, ovl.rollout(ev))
            for x, y in zip(seq.final_states(), ovl.final_states()):
                assert torch.equal(x, y)
        assert ("run", 0, T) in ovl._graphs2
        for n in (12, 13, 14, 15, 16):                  # ... and only the last few lengths are kept
            for _ in range(2):
                ovl.reset()
                ovl.run(n)
        assert len([k for k in ovl._graphs2 if k[0] == "run"]) == ovl.WHOLE_RUN_GRAPHS and ("run", 0, T) not in ovl._graphs2
        assert torch.equal(a, ovl.rollout(ev))


@pytest.mark.parametrize("levels", [False, True])
@pytest.mark.parametrize("T", [1, 2, 3, 5])
def test_very_short_events_on_the_overlapped_schedule(dev, T, levels):
    """Events shorter than a group of iterations (and than the capture warm-up's frames): prologue + first iteration + trailing head
    only, output buffers of one to five rows -- same frames and states as the one-chain engine, twice through the same graphs."""
    import urnn_amd.weights as uw
    from urnn_amd.rollout import RolloutEngine
    H, W, nums = 32, 48, 3
    net, _ = make_net(H, W, 9, 3, dev)
    ev = uw.make_event(T, H, W, 60.0, seed=4)
    seq = RolloutEngine(net, H, W, nums, 60.0, 250.0, max_frames=T)
    a = seq.rollout(ev).clone()
    ovl = RolloutEngine(net, H, W, nums, 60.0, 250.0, max_frames=T, overlap=True, levels=levels)
    for _ in range(2):
        b = ovl.rollout(ev)
        assert b.shape == a.shape and torch.equal(a, b)
        for x, y in zip(seq.final_states(), ovl.final_states()):
            assert torch.equal(x, y)


def test_batched_events_match_single_events(dev):
    """Event batching (a build-side extension): per-sample semantics -- a batch of two events must
    equal the two events rolled out one by one."""
    import urnn_amd.weights as uw
    from urnn_amd.rollout import RolloutEngine
    H, W, nums, T = 40, 24, 3, 6
    net, _ = make_net(H, W, 9, 4, dev)
    ev2 = uw.make_event(T, H, W, 60.0, seed=7, batch=2)
    # the reference normalises every sample with sample 0's DEM range (Dynamic2DFlood.py:305-306); mirror that here
    eng2 = RolloutEngine(net, H, W, nums, 60.0, 250.0, batch=2, max_frames=T)
    both = eng2.rollout(ev2).clone()
    eng1 = RolloutEngine(net, H, W, nums, 60.0, 250.0, batch=1, max_frames=T)
    for b in range(2):
        ev1 = {k: (v[b:b + 1] if k not in ("max_DEM", "min_DEM") else v[0:1]) for k, v in ev2.items()}
        one = eng1.rollout(ev1)
        assert torch.equal(one[:, 0], both[:, b])


def test_full_size_step_properties(dev):
    """BASELINE config 2 size (500x500, C=63): two timesteps through the engine; size-independent properties --
    finite outputs, masked == raw * [cls >= 0.5] exactly, cls in [0,1], states bounded by the GRU convex blend
    (|h'| <= max(|h|, 1)), and the run is bit-reproducible."""
    import urnn_amd.weights as uw
    from urnn_amd.rollout import RolloutEngine
    H = W = 500
    nums, T = 30, 3
    net, _ = make_net(H, W, 63, 0, dev)
    ev = uw.make_event(T, H, W, 6.0, seed=42)
    eng = RolloutEngine(net, H, W, nums, 6.0, 250.0, max_frames=T, keep_raw=True)
    a = eng.rollout(ev).clone()
    cls, raw = eng.out_cls[:T].clone(), eng.out_raw[:T].clone()
    assert torch.isfinite(a).all() and torch.isfinite(raw).all()
    assert (cls >= 0).all() and (cls <= 1).all()
    assert torch.equal(a, raw * (cls >= 0.5).float())
    for s in eng.states:
        assert torch.isfinite(s).all() and s.abs().max() <= 1.0 + 1e-6
    b = eng.rollout(ev)
    assert torch.equal(a, b)


def test_full_size_cell_vs_oracle(dev):
    """The largest GEMM (decoder stage-1 Skip-ConvGRU, 500x500, K=224) against the CPU oracle on a 500x500 plane."""
    import urnn_amd.weights as uw
    from oracle import oracle as orc
    H = W = 500
    net, sd = make_net(H, W, 63, 0, dev)
    rs = np.random.RandomState(11)
    x = (0.5 * rs.standard_normal((1, 96, H, W))).astype(np.float32)
    e = (0.5 * rs.standard_normal((1, 64, H, W))).astype(np.float32)
    d = (0.5 * rs.standard_normal((1, 64, H, W))).astype(np.float32)
    got = net.decoder.rnn1.step(*(torch.from_numpy(v).to(dev) for v in (x, e, d))).cpu().numpy()
    ref = orc.gru_cell(x, e, d, orc.OracleNet(sd).dec[1])
    assert_close(got, ref, 1e-4, "dec1 cell at 500x500")


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["enc1", "dec1"])
def test_fused_reset_gate_cell_vs_three_pass_and_oracle(dev, which):
    """URNN_PHASE_FUSED_R (the rollout engine's cell: reset gate recomputed inside the candidate kernel, its raw planes never
    stored; ConvRNN.py:165-180) against the three-pass cell and the CPU oracle on a 500x500 plane, both full-resolution cells.
    The update gate's raw planes and the gates' folded statistics must be BIT-identical (same accumulators); the candidate sums
    its hidden-state channels in another order inside each 16-k MFMA group, so h' is compared at rounding level."""
    from oracle import oracle as orc
    from urnn_amd import ops
    H = W = 500
    net, sd = make_net(H, W, 63, 0, dev)
    rs = np.random.RandomState(12)
    if which == "enc1":
        cell, ocell = net.encoder.rnn1, orc.OracleNet(sd).enc[0]
        x = (0.5 * rs.standard_normal((1, 16, H, W))).astype(np.float32)
        e = None
    else:
        cell, ocell = net.decoder.rnn1, orc.OracleNet(sd).dec[1]
        x = (0.5 * rs.standard_normal((1, 96, H, W))).astype(np.float32)
        e = (0.5 * rs.standard_normal((1, 64, H, W))).astype(np.float32)
    h = (0.5 * rs.standard_normal((1, 64, H, W))).astype(np.float32)
    tx, th = torch.from_numpy(x).to(dev), torch.from_numpy(h).to(dev)
    te = None if e is None else torch.from_numpy(e).to(dev)
    nbytes = ops.gru_cell_workspace_bytes(1, 64, H, W)
    ws_a = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    ws_b = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    plain = cell.step(tx, te, th, ws=ws_a)
    fused = cell.step(tx, te, th, phases=ops.PHASE_ALL | ops.PHASE_FUSED_R, ws=ws_b)
    torch.cuda.synchronize()
    P = H * W
    S0 = ops.STATUS_AREA_BYTES                 # the workspace's status area (include/urnn_hip.h), then the raw gate planes
    g1_a, g1_b = ws_a[S0:S0 + 2 * 64 * P * 4].view(torch.float32), ws_b[S0:S0 + 2 * 64 * P * 4].view(torch.float32)
    assert torch.equal(g1_a[:64 * P], g1_b[:64 * P]), "raw update gate differs"
    assert float(g1_b[64 * P:].abs().max()) == 0.0, "the fused cell wrote reset-gate planes"      # (the workspace was zero-filled)
    assert float(g1_a[64 * P:].abs().max()) > 0.0
    d = float((plain - fused).abs().max())
    print(f"{which}: fused vs three-pass max |dh'| = {d:.2e}")
    assert d <= 2e-6
    ref = orc.gru_cell(x, e, h, ocell) if e is not None else orc.gru_cell(x, None, h, ocell)
    assert_close(fused.cpu().numpy(), ref, 1e-4, f"{which} cell (reset gate recomputed) at 500x500")
    again = cell.step(tx, te, th, phases=ops.PHASE_ALL | ops.PHASE_FUSED_R, ws=ws_b)
    assert torch.equal(again, fused)


def test_rollout_cell_on_an_odd_full_resolution_plane(dev):
    """A full-resolution cell whose plane is not a multiple of four pixels (501 x 499) cannot take the fused candidate kernel; called the
    way the rollout engine calls it (URNN_PHASE_FUSED_R) its candidate GEMM then runs on the fp32 matrix instruction (include/urnn_hip.h)
    -- another arithmetic than the plain call's f16 pieces, both within 1e-4 of the oracle (ConvRNN.py:111-194)."""
    from oracle import oracle as orc
    from urnn_amd import ops
    H, W = 501, 499
    net, sd = make_net(H, W, 63, 0, dev)
    rs = np.random.RandomState(13)
    x = (0.5 * rs.standard_normal((1, 16, H, W))).astype(np.float32)
    h = (0.5 * rs.standard_normal((1, 64, H, W))).astype(np.float32)
    tx, th = torch.from_numpy(x).to(dev), torch.from_numpy(h).to(dev)
    cell = net.encoder.rnn1
    plain = cell.step(tx, None, th)
    flagged = cell.step(tx, None, th, phases=ops.PHASE_ALL | ops.PHASE_FUSED_R)
    ref = orc.gru_cell(x, None, h, orc.OracleNet(sd).enc[0])
    assert_close(plain.cpu().numpy(), ref, 1e-4, "enc1 cell at 501x499, plain call")
    assert_close(flagged.cpu().numpy(), ref, 1e-4, "enc1 cell at 501x499, rollout call")
    d = float((plain - flagged).abs().max())
    print(f"501x499: rollout call vs plain call max |dh'| = {d:.2e}")
    assert 0.0 < d <= 3e-6


_ORACLE_CACHE = {}


def _oracle_rollout(sd, ev, T, nums, rain_max, cum_max, key):
    """CPU oracle rollout, cached per test module (the 500x500 one costs ~1 s per frame on the GPU box's host cores)."""
    from oracle import oracle as orc
    if key not in _ORACLE_CACHE:
        frames, states, aux = orc.rollout(orc.OracleNet(sd), ev, T, nums, rain_max, cum_max, want_aux=True)
        _ORACLE_CACHE[key] = (frames, states, np.stack([a["reg_raw"] for a in aux]), np.stack([a["cls"] for a in aux]))
    return _ORACLE_CACHE[key]


_TORCH32_CACHE = {}


def _torch_fp32_state_errors(sd, ev, T, nums, rain_max, cum_max, dev, ref_states, key):
    """How far the REFERENCE'S OWN arithmetic -- the same modules evaluated by plain float32 torch ops on this GPU
    (tests/torch_ref.py, pinned to the reference goldens) -- ends up from the oracle's final states: the yardstick for what
    any fp32 implementation can promise after T recurrent steps (conftest.rel_err metric)."""
    import torch_ref
    from urnn_amd.dataset import preprocess_inputs
    if key not in _TORCH32_CACHE:
        H, W = ref_states[0].shape[-2:]
        p = {k: torch.from_numpy(v).to(dev) for k, v in sd.items()}
        st = [torch.zeros(s.shape, device=dev) for s in ref_states]
        with torch.no_grad():
            for t in range(T):
                x = preprocess_inputs(t, ev, dev, nums=nums, rain_max=rain_max, cumsum_rain_max=cum_max)[:, 0]
                _, _, _, st = torch_ref.step(p, x, st, H, W)
        _TORCH32_CACHE[key] = [rel_err(a.cpu().numpy(), b) for a, b in zip(st, ref_states)]
    return _TORCH32_CACHE[key]


def _check_rollout_vs_oracle(eng, frames, T, ref, what, state_yardstick=None):
    """cls and the pre-mask regression of EVERY frame within 1e-4 of the oracle; the masked depth away from the threshold; the
    final states within 1e-4 -- or, for long full-size rollouts where fp32 roundoff itself accumulates past that
    (state_yardstick = the plain-fp32-torch errors of _torch_fp32_state_errors), no further from the oracle than 3x what the
    reference's own fp32 arithmetic is (the rule of test_rollout_vs_reference)."""
    ref_frames, ref_states, ref_raw, ref_cls = ref
    got_raw, got_cls = eng.out_raw[:T].cpu().numpy(), eng.out_cls[:T].cpu().numpy()
    # both floors are printed (SURVEY 8c proposes 1e-3 * max|b|; conftest.rel_err explains the 0.1 * max|b| the 1e-4 bar uses)
    print(f"{what}: worst frame, pre-mask reg: {rel_err(got_raw, ref_raw):.2e} (floor 0.1 max) / {rel_err(got_raw, ref_raw, 1e-3):.2e} (strict floor "
          f"1e-3 max); cls: {rel_err(got_cls, ref_cls):.2e} / {rel_err(got_cls, ref_cls, 1e-3):.2e}")
    assert_close(got_raw, ref_raw, 1e-4, f"pre-mask reg, {what}")
    assert_close(got_cls, ref_cls, 1e-4, f"cls, {what}")
    report = []
    for k, (got, want) in enumerate(zip(eng.final_states(), ref_states)):
        g = got.cpu().numpy()
        err = rel_err(g, want)
        bar = 1e-4 if state_yardstick is None else max(1e-4, 3.0 * state_yardstick[k])
        report.append((k, f"{err:.2e}", f"strict {rel_err(g, want, 1e-3):.2e}", None if state_yardstick is None else f"torch {state_yardstick[k]:.2e}"))
        assert err <= bar, f"final state {k}, {what}: rel err {err:.3e} > {bar:.1e} (plain fp32 torch: {state_yardstick and state_yardstick[k]})"
    print(f"{what}: final-state errors (state, HIP vs oracle, same under the strict floor, plain fp32 torch vs oracle): {report}")
    excluded = masked_parity(frames, ref_frames, ref_cls, ref_raw, 1e-4)
    assert excluded < 1e-3 * frames.size, f"{excluded} threshold pixels excluded, {what}"     # |cls - 0.5| <= 1e-5: ~1e-4 of the pixels


@pytest.mark.parametrize("overlap", [True, False])
def test_full_size_rollout_vs_oracle(dev, overlap):
    """BASELINE configs[1] (location1: 500x500, C = 63) end to end on the schedule bench.py times -- captured hipGraph,
    two overlapped kernel chains (overlap=True), input assembly folded into the first stage -- and on the one-chain
    schedule: T = 36 frames (a tenth of the event; one SWP window of the reference, location1_scratch.yaml:56-58)
    against the CPU oracle: cls and the pre-mask regression of every frame within 1e-4, the masked depth wherever the oracle is
    not within 1e-5 of the wet/dry threshold, every final state (see _check_rollout_vs_oracle) (test.py:326-377)."""
    import urnn_amd.weights as uw
    from urnn_amd.rollout import RolloutEngine
    H = W = 500
    nums, T = 30, 36
    net, sd = make_net(H, W, 2 * nums + 3, 0, dev)
    ev = uw.make_event(T, H, W, 6.0, seed=42)
    eng = RolloutEngine(net, H, W, nums, 6.0, 250.0, max_frames=T, keep_raw=True, overlap=overlap, use_graph=True)
    frames = eng.rollout(ev).cpu().numpy()
    ref = _oracle_rollout(sd, ev, T, nums, 6.0, 250.0, ("location1", T))
    yard = _torch_fp32_state_errors(sd, ev, T, nums, 6.0, 250.0, dev, ref[1], ("location1", T))
    _check_rollout_vs_oracle(eng, frames, T, ref, f"500x500 overlap={overlap}", state_yardstick=yard)
    again = eng.rollout(ev).cpu().numpy()           # second event through the same captured graphs
    assert np.array_equal(frames, again)


@pytest.mark.parametrize("mode", ["fp32_mfma", "fp32_cand"])
def test_fp32_mfma_matrix_mode_rollout_vs_oracle(dev, mode):
    """urnn_set_matrix_mode(URNN_MATRIX_FP32_MFMA): every GEMM on the exact fp32 matrix instructions; URNN_MATRIX_FP32_CAND: only the
    full-resolution cells' candidate GEMM (the modes for digit-by-digit comparisons of long rollouts) -- the
    location1 rollout of test_full_size_rollout_vs_oracle under each: same bars; the engine re-captures its graphs when the
    process-wide mode changes and goes back to the default mode's bits afterwards."""
    import urnn_amd.weights as uw
    from urnn_amd import ops
    from urnn_amd.rollout import RolloutEngine
    H = W = 500
    nums, T = 30, 36
    net, sd = make_net(H, W, 2 * nums + 3, 0, dev)
    ev = uw.make_event(T, H, W, 6.0, seed=42)
    eng = RolloutEngine(net, H, W, nums, 6.0, 250.0, max_frames=T, keep_raw=True, overlap=True, use_graph=True)
    default = eng.rollout(ev).cpu().numpy()
    ref = _oracle_rollout(sd, ev, T, nums, 6.0, 250.0, ("location1", T))
    yard = _torch_fp32_state_errors(sd, ev, T, nums, 6.0, 250.0, dev, ref[1], ("location1", T))
    with ops.matrix_mode(mode):
        frames = eng.rollout(ev).cpu().numpy()
        _check_rollout_vs_oracle(eng, frames, T, ref, f"500x500 {mode}", state_yardstick=yard)
    assert not np.array_equal(frames, default)          # (another arithmetic: equal to rounding, not to the bit)
    assert np.array_equal(eng.rollout(ev).cpu().numpy(), default)


def test_mid_event_slice_vs_oracle(dev):
    """Frames 60 .. 179 of the location1 event (BASELINE configs[1]: 500x500, C = 63, seed 42 -- the stretch where roundoff is
    amplified most, profiles/r02_parity_T360.txt) on the benchmarked schedule: the engine rolls frames 0 .. 59, hands its states
    to the CPU oracle and to plain float32 torch (the reference's own arithmetic), and all three run the next 120 frames
    (tests/slice_parity.py; test.py:352-371).  Every frame's cls / pre-mask regression and the final states within
    max(1e-4, 3x the torch-fp32 error of the same frame); relaxed- and strict-floor errors are printed side by side; and over
    the slice the HIP path must not be further from the oracle than the reference's arithmetic is (mean error ratio <= 1)."""
    import slice_parity
    res = slice_parity.run_slice(dev, t0=60, n=120)
    slice_parity.report(res)
    for r in res["rows"]:
        assert r[1] <= max(1e-4, 3.0 * r[2]), f"frame {r[0]}: pre-mask reg {r[1]:.2e} (torch-fp32 {r[2]:.2e})"
        assert r[3] <= max(1e-4, 3.0 * r[4]), f"frame {r[0]}: cls {r[3]:.2e} (torch-fp32 {r[4]:.2e})"
    for k, eh, et, _, _ in res["states"]:
        assert eh <= max(1e-4, 3.0 * et), f"state {k}: {eh:.2e} (torch-fp32 {et:.2e})"
    ratio_reg = float(np.mean([r[1] / r[2] for r in res["rows"]]))
    ratio_cls = float(np.mean([r[3] / r[4] for r in res["rows"]]))
    assert ratio_reg <= 1.0 and ratio_cls <= 1.0, f"mean HIP / torch-fp32 error ratio: reg {ratio_reg:.2f}, cls {ratio_cls:.2f}"


def test_benchmarked_schedule_is_bit_stable_over_whole_events(dev):
    """Eight whole location1 events (360 frames each at 500x500) through the captured two-chain schedule: frames, class map and
    final states identical to the first run every time -- an ordering hazard between the two kernel chains or inside a ring
    shows up as a rare difference (tools/soak_rollout.py runs 60)."""
    import urnn_amd.weights as uw
    from urnn_amd.rollout import RolloutEngine
    H = W = 500
    nums, T = 30, 360
    net, sd = make_net(H, W, 2 * nums + 3, 0, dev)
    ev = uw.make_event(T, H, W, 6.0, seed=5)
    eng = RolloutEngine(net, H, W, nums, 6.0, 250.0, max_frames=T, overlap=True, use_graph=True)
    ref = eng.rollout(ev).clone()
    ref_cls = eng.out_cls[:T].clone()
    ref_states = [s.clone() for s in eng.final_states()]
    for i in range(8):
        out = eng.rollout(ev)
        assert torch.equal(out, ref) and torch.equal(eng.out_cls[:T], ref_cls), f"event {i}"
        assert all(torch.equal(a, b) for a, b in zip(eng.final_states(), ref_states)), f"event {i}: states"


@pytest.mark.skipif(int(os.environ.get("URNN_LONG_T", "0")) < 1, reason="opt-in: URNN_LONG_T=<frames> (the C oracle does ~0.9 frames/s at 500x500)")
def test_whole_event_rollout_vs_oracle(dev):
    """Opt-in evidence run (URNN_LONG_T=360 is the whole location1 event): the benchmarked schedule against the CPU oracle for
    T frames at 500x500.  Roundoff accumulates through the recurrence, so next to the HIP errors the test measures what the
    REFERENCE'S OWN arithmetic (plain float32 torch, tests/torch_ref.py) does on the same frames and holds cls, the pre-mask
    regression (per frame) and the final states to max(1e-4, 3x that).  Prints the error profile over time (kept under
    profiles/ by the run that produced it)."""
    import torch_ref
    import urnn_amd.weights as uw
    from oracle import oracle as orc
    from urnn_amd.dataset import preprocess_inputs
    from urnn_amd.rollout import RolloutEngine
    H = W = 500
    nums, T = 30, int(os.environ["URNN_LONG_T"])
    net, sd = make_net(H, W, 2 * nums + 3, 0, dev)
    ev = uw.make_event(T, H, W, 6.0, seed=42)
    eng = RolloutEngine(net, H, W, nums, 6.0, 250.0, max_frames=T, keep_raw=True, overlap=True, use_graph=True)
    from urnn_amd import ops
    mode = os.environ.get("URNN_LONG_MODE", "fp32")          # the GEMM arithmetic of the record (include/urnn_hip.h urnn_set_matrix_mode)
    print(f"matrix mode {mode}")
    with ops.matrix_mode(mode):
        eng.rollout(ev)
    hip_raw, hip_cls = eng.out_raw[:T].cpu().numpy(), eng.out_cls[:T].cpu().numpy()
    onet = orc.OracleNet(sd)
    pt = {k: torch.from_numpy(v).to(dev) for k, v in sd.items()}
    # frame by frame (oracle.rollout's loop, test.py:326-377), so that nothing but the current frame is kept
    ost = orc.zero_states(1, H, W)
    tst = [torch.zeros(s.shape, device=dev) for s in eng.final_states()]
    rows, worst = [], 0.0
    for t in range(T):
        _, ost, aux0 = onet.step(orc.preprocess_inputs(t, ev, nums, 6.0, 250.0)[:, 0], ost, True)
        with torch.no_grad():
            x = preprocess_inputs(t, ev, dev, nums=nums, rain_max=6.0, cumsum_rain_max=250.0)[:, 0]
            _, tcls, traw, tst = torch_ref.step(pt, x, tst, H, W)
        ref_raw, ref_cls = aux0["reg_raw"], aux0["cls"]
        e_hr, e_hc = rel_err(hip_raw[t].reshape(ref_raw.shape), ref_raw), rel_err(hip_cls[t].reshape(ref_cls.shape), ref_cls)
        e_tr, e_tc = rel_err(traw.cpu().numpy().reshape(ref_raw.shape), ref_raw), rel_err(tcls.cpu().numpy().reshape(ref_cls.shape), ref_cls)
        rows.append((t, e_hr, e_tr, e_hc, e_tc))
        worst = max(worst, e_hr / max(1e-4, 3 * e_tr), e_hc / max(1e-4, 3 * e_tc))
        if t % 20 == 0 or t == T - 1:
            print(f"frame {t:4d}: reg HIP {e_hr:.2e} torch-fp32 {e_tr:.2e} | cls HIP {e_hc:.2e} torch-fp32 {e_tc:.2e}", flush=True)
    srep = []
    for k, (got, want, tt) in enumerate(zip(eng.final_states(), ost, tst)):
        eh, et = rel_err(got.cpu().numpy(), want), rel_err(tt.cpu().numpy(), want)
        srep.append((k, eh, et))
        worst = max(worst, eh / max(1e-4, 3 * et))
    print("final states (state, HIP vs oracle, torch-fp32 vs oracle):", [(k, f"{a:.2e}", f"{b:.2e}") for k, a, b in srep])
    print(f"max over frames: reg HIP {max(r[1] for r in rows):.2e} torch {max(r[2] for r in rows):.2e}; "
          f"cls HIP {max(r[3] for r in rows):.2e} torch {max(r[4] for r in rows):.2e}; worst error / bar = {worst:.2f}")
    assert worst <= 1.0


def _sub_err(got, want, plane_max):
    """conftest.rel_err restricted to a subset of a tensor: the floor is 0.1 x the max |.| of the WHOLE reference tensor."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return float((np.abs(got - want) / np.maximum(np.abs(want), 0.1 * max(float(plane_max), 1e-30))).max())


@pytest.mark.parametrize("trace", ["whole_event_500x500_T360.npz", "whole_event_128x128_T360.npz"])
def test_whole_event_vs_committed_oracle_trace(dev, trace):
    """The headline parity claim in the DEFAULT suite (reference loop test.py:326-377): the benchmarked schedule (hipGraph, three
    kernel chains) over ALL 360 frames of the event, against a sparse trace of the CPU oracle committed under tests/golden
    (tests/golden/make_whole_event_trace.py generated it with oracle/ on the GPU box's host cores): every 4th frame -- every 2nd
    through the rain peak, frames 60-180 -- on 4096 fixed random pixels AND on that frame's adversarial pixels: the 1024 where
    plain float32 torch is furthest from the oracle and the 1024 whose class score is closest to the wet/dry threshold outside the
    exclusion band; plus a subset of every final state.  Bars per sampled frame:
      * max(1e-4, 3 x what the reference's own arithmetic -- float32 torch, recorded in the trace on the same pixels -- is away
        from the oracle on that frame) under the tests' floor (0.1 x the plane's max), and
      * under SURVEY 8c's strict floor (1e-3 x the plane's max): no further from the oracle than 3 x float32 torch.
    The trace also records torch's FULL-plane error per frame; the test prints it beside the sampled figures.
    500x500: BASELINE configs[1], the full-resolution cells on the fused candidate kernel (recurrent product on the fp32
    instruction); 128x128: every cell on the small-plane kernels, whose recurrent product stays on f16 pieces."""
    import urnn_amd.weights as uw
    from urnn_amd.rollout import RolloutEngine
    path = os.path.join(os.path.dirname(__file__), "golden", trace)
    if not os.path.isfile(path):
        pytest.fail(f"{trace} missing: generate it with tests/golden/make_whole_event_trace.py")
    g = np.load(path)
    H, W, nums, T = int(g["H"]), int(g["W"]), int(g["nums"]), int(g["T"])
    rain_max, cum_max = float(g["rain_max"]), float(g["cumsum_max"])
    net, sd = make_net(H, W, 2 * nums + 3, int(g["weights_seed"]), dev)
    ev = uw.make_event(T, H, W, rain_max, seed=int(g["event_seed"]))
    eng = RolloutEngine(net, H, W, nums, rain_max, cum_max, max_frames=T, keep_raw=True, overlap=True, use_graph=True)
    eng.rollout(ev)
    fr, pix = g["frames"], g["pixels"]
    adv = g["adv_idx"] if "adv_idx" in g.files else None
    fr_d = torch.from_numpy(fr).long().to(dev)
    raw_f = eng.out_raw[:T, 0].reshape(T, -1)[fr_d]
    cls_f = eng.out_cls[:T, 0].reshape(T, -1)[fr_d]
    pix_d = torch.from_numpy(pix).long().to(dev)
    raw, cls = raw_f[:, pix_d].cpu().numpy(), cls_f[:, pix_d].cpu().numpy()
    if adv is not None:
        adv_d = torch.from_numpy(adv.astype(np.int64)).to(dev)
        raw_a, cls_a = torch.gather(raw_f, 1, adv_d).cpu().numpy(), torch.gather(cls_f, 1, adv_d).cpu().numpy()

    def strict(got, want, plane_max):
        got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
        return float((np.abs(got - want) / np.maximum(np.abs(want), 1e-3 * max(float(plane_max), 1e-30))).max())

    worst, wr, wc, wtr, wfull, wstrict = 0.0, 0.0, 0.0, 0.0, 0.0, 0.0
    for i, t in enumerate(fr):
        rmax, cmax = g["oracle_raw_plane_max"][i], g["oracle_cls_plane_max"][i]
        er = _sub_err(raw[i], g["oracle_raw"][i], rmax)
        ec = _sub_err(cls[i], g["oracle_cls"][i], cmax)
        tr, tc = float(g["torch32_reg_err"][i]), float(g["torch32_cls_err"][i])
        worst = max(worst, er / max(1e-4, 3 * tr), ec / max(1e-4, 3 * tc))
        line = f"frame {int(t):4d}: reg HIP {er:.2e} torch-fp32 {tr:.2e} | cls HIP {ec:.2e} torch-fp32 {tc:.2e}"
        if adv is not None:
            era = _sub_err(raw_a[i], g["adv_oracle_raw"][i], rmax)
            eca = _sub_err(cls_a[i], g["adv_oracle_cls"][i], cmax)
            tra, tca = float(g["torch32_reg_err_adv"][i]), float(g["torch32_cls_err_adv"][i])
            worst = max(worst, era / max(1e-4, 3 * tra), eca / max(1e-4, 3 * tca))
            # SURVEY 8c's strict floor on random + adversarial pixels together: HIP no further from the oracle than 3 x float32 torch
            srs = max(strict(raw[i], g["oracle_raw"][i], rmax), strict(raw_a[i], g["adv_oracle_raw"][i], rmax))
            scs = max(strict(cls[i], g["oracle_cls"][i], cmax), strict(cls_a[i], g["adv_oracle_cls"][i], cmax))
            wstrict = max(wstrict, srs / max(3 * float(g["torch32_reg_err_strict"][i]), 1e-30), scs / max(3 * float(g["torch32_cls_err_strict"][i]), 1e-30))
            wfull = max(wfull, float(g["torch32_reg_err_full"][i]))
            line += f" || adversarial pixels: reg HIP {era:.2e} torch-fp32 {tra:.2e} (its full plane {float(g['torch32_reg_err_full'][i]):.2e}) | cls HIP {eca:.2e} torch-fp32 {tca:.2e}"
            er, ec, tr = max(er, era), max(ec, eca), max(tr, tra)
        wr, wc, wtr = max(wr, er), max(wc, ec), max(wtr, tr)
        if i % 15 == 0 or i == len(fr) - 1:
            print(line)
    srep = []
    for k, st in enumerate(eng.final_states()):
        es = _sub_err(st.reshape(-1)[torch.from_numpy(g[f"state{k}_idx"]).to(dev)].cpu().numpy(), g[f"state{k}_oracle"], g["state_plane_max"][k])
        et = float(g["torch32_state_err"][k])
        srep.append((k, f"{es:.2e}", f"{et:.2e}"))
        worst = max(worst, es / max(1e-4, 3 * et))
    print(f"{trace}: {len(fr)} of {T} frames x {len(pix)} random" + (f" + {adv.shape[1]} adversarial" if adv is not None else "") +
          f" pixels: worst frame reg HIP {wr:.2e} (torch-fp32 on the same pixels {wtr:.2e}" + (f", on its full plane {wfull:.2e}" if adv is not None else "") +
          f"), cls {wc:.2e}; final states (state, HIP, torch-fp32) {srep}; worst error / bar = {worst:.2f}" +
          (f"; strict 1e-3 floor: worst HIP / (3 x torch-fp32) = {wstrict:.2f}" if adv is not None else ""))
    assert worst <= 1.0
    assert wstrict <= 1.0


def _subset_err(got, want, plane_max, floor_frac):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return float((np.abs(got - want) / np.maximum(np.abs(want), floor_frac * max(float(plane_max), 1e-30))).max())


@pytest.mark.parametrize("trace", ["reference_trace_500x500_T360.npz", "reference_trace_400x560_T72.npz"])
def test_whole_event_vs_reference_trace(dev, trace):
    """The headline config pinned to the REFERENCE ITSELF over the whole event (reference loop test.py:352-371,
    model.py:65-121).  tests/golden/make_reference_trace.py ran the reference's own ED in the build container -- as shipped (float32,
    "ref32") and as a float64 copy of the same modules ("ref64") -- for location1 (500x500, C = 63, T = 360) and Futian (400x560,
    C = 15, T = 72, spatial rain) and kept, per sampled frame, 4096 random pixels + the 1024 where ref32 and ref64 differ most + the
    1024 nearest the wet/dry threshold, plus subsets of the final states.  The benchmarked schedule (hipGraph, three kernel chains)
    is compared with ref64; the yardstick is what the reference's own float32 evaluation (ref32) is away from ref64 on the same values.

    Both are float32 evaluations of one graph with independent roundoff, amplified through the recurrence at a few pixels around the
    rain peak: WHICH frame carries the worst pixel differs between them (ref32: frame 76, HIP: frame 84), so a bar of 1.5 x ref32's
    error on the very same frame is exceeded by ANY float32 implementation somewhere -- measured (tools/parity_reference_trace.py,
    profiles/r06_parity_reference_trace.txt): under the tests' floor the default mode 1.18 x at 2 of 121 frames, the exact-fp32-MFMA
    mode 0.74 x; under SURVEY 8c's strict floor the default mode 2.23 x and the exact-fp32-MFMA mode 2.12 x.  The bars therefore are
      R1  floor 0.1 x plane max, every sampled frame t:  HIP(t) <= max(1e-4, 1.5 x max ref32(t') over sampled |t' - t| <= 8)
      R2  strict floor 1e-3 x plane max, every frame:    HIP(t) <= max(1e-4, 3 x the same windowed maximum)
      R3  both floors, whole event: HIP's worst frame <= 1.25 x ref32's worst frame, HIP's mean over frames <= 1.1 x ref32's mean
      R4  final states: within max(1e-4, 1.5 x ref32) (floor 0.1) / max(1e-4, 3 x ref32) (strict floor)
    and the unwindowed per-frame ratio, HIP against ref32, and the number of frames where HIP is the closer one are printed."""
    import urnn_amd.weights as uw
    from urnn_amd.rollout import RolloutEngine
    path = os.path.join(os.path.dirname(__file__), "golden", trace)
    if not os.path.isfile(path):
        pytest.fail(f"{trace} missing: generate it in the build container with tests/golden/make_reference_trace.py")
    g = np.load(path)
    H, W, nums, T = int(g["H"]), int(g["W"]), int(g["nums"]), int(g["T"])
    rain_max, cum_max, spatial = float(g["rain_max"]), float(g["cumsum_max"]), bool(int(g["spatial"]))
    net, sd = make_net(H, W, 2 * nums + 3, int(g["weights_seed"]), dev)
    ev = uw.make_event(T, H, W, rain_max, seed=int(g["event_seed"]), spatial_rain=spatial)
    eng = RolloutEngine(net, H, W, nums, rain_max, cum_max, max_frames=T, spatial_rain=spatial, keep_raw=True, overlap=True, use_graph=True)
    eng.rollout(ev)
    fr = g["frames"].astype(np.int64)
    fr_d = torch.from_numpy(fr).to(dev)
    idx = np.concatenate([np.broadcast_to(g["pixels"].astype(np.int64), (len(fr), g["pixels"].size)), g["adv_idx"].astype(np.int64)], axis=1)
    idx_d = torch.from_numpy(np.ascontiguousarray(idx)).to(dev)
    raw = torch.gather(eng.out_raw[:T, 0].reshape(T, -1)[fr_d], 1, idx_d).cpu().numpy()
    cls = torch.gather(eng.out_cls[:T, 0].reshape(T, -1)[fr_d], 1, idx_d).cpu().numpy()
    r64 = np.concatenate([g["r64_raw"], g["a64_raw"]], axis=1)
    c64 = np.concatenate([g["r64_cls"], g["a64_cls"]], axis=1)
    r32 = np.concatenate([g["r32_raw"], g["a32_raw"]], axis=1)
    c32 = np.concatenate([g["r32_cls"], g["a32_cls"]], axis=1)
    E = {}
    for floor in (0.1, 1e-3):
        for what, got, want in (("reg", raw, r64), ("cls", cls, c64), ("reg32", r32, r64), ("cls32", c32, c64), ("regv32", raw, r32), ("clsv32", cls, c32)):
            pm = g["ref64_raw_plane_max"] if what.startswith("reg") else g["ref64_cls_plane_max"]
            E[(floor, what)] = np.asarray([_subset_err(got[i], want[i], pm[i], floor) for i in range(len(fr))])

    def windowed(a, w=8):
        return np.asarray([a[np.abs(fr - t) <= w].max() for t in fr])
    fail = []
    for floor, mult, nm in ((0.1, 1.5, "floor 0.1 x plane max"), (1e-3, 3.0, "STRICT floor 1e-3 x plane max (SURVEY 8c)")):
        for q in ("reg", "cls"):
            hip, ref = E[(floor, q)], E[(floor, q + "32")]
            ratio_w = hip / np.maximum(1e-4, mult * windowed(ref))
            ratio_0 = hip / np.maximum(1e-4, 1.5 * ref)
            i = int(ratio_w.argmax())
            print(f"{trace} [{nm}] {q}: worst frame HIP vs ref64 {hip.max():.2e} (frame {int(fr[hip.argmax()])}), the reference's float32 vs its float64 {ref.max():.2e} "
                  f"(frame {int(fr[ref.argmax()])}); mean over {len(fr)} frames {hip.mean():.2e} / {ref.mean():.2e}; HIP closer to ref64 than ref32 is on {int((hip < ref).sum())} frames; "
                  f"HIP vs ref32 worst {E[(floor, q + 'v32')].max():.2e}; per-frame bar: worst HIP / max(1e-4, {mult} x windowed ref32) = {ratio_w.max():.2f} (frame {int(fr[i])}); "
                  f"unwindowed HIP / max(1e-4, 1.5 x ref32 on the same frame) = {ratio_0.max():.2f}, over 1 on {int((ratio_0 > 1).sum())} frames")
            if ratio_w.max() > 1.0:
                fail.append(f"{nm} {q}: frame {int(fr[i])} is {ratio_w.max():.2f} x the per-frame bar")
            if hip.max() > max(1e-4, 1.25 * ref.max()):
                fail.append(f"{nm} {q}: worst frame {hip.max():.2e} > 1.25 x the reference's own {ref.max():.2e}")
            if hip.mean() > max(1e-5, 1.1 * ref.mean()):
                fail.append(f"{nm} {q}: mean over frames {hip.mean():.2e} > 1.1 x the reference's own {ref.mean():.2e}")
    srep = []
    for k, st in enumerate(eng.final_states()):
        got = st.reshape(-1)[torch.from_numpy(g[f"state{k}_idx"]).to(dev)].cpu().numpy()
        smax = g["state_plane_max"][k]
        for floor, mult in ((0.1, 1.5), (1e-3, 3.0)):
            eh = _subset_err(got, g[f"state{k}_ref64"], smax, floor)
            e32 = _subset_err(g[f"state{k}_ref32"], g[f"state{k}_ref64"], smax, floor)
            srep.append((k, floor, f"{eh:.2e}", f"{e32:.2e}"))
            if eh > max(1e-4, mult * e32):
                fail.append(f"final state {k} (floor {floor}): {eh:.2e} vs the reference's float32 {e32:.2e}")
    print(f"{trace} final states (state, floor, HIP vs ref64, ref32 vs ref64): {srep}")
    print(f"{trace}: the reference's own FULL-plane float32-vs-float64 error, worst frame: reg {float(g['ref32_reg_err_full'].max()):.2e} (strict {float(g['ref32_reg_err_full_strict'].max()):.2e}), "
          f"cls {float(g['ref32_cls_err_full'].max()):.2e} (strict {float(g['ref32_cls_err_full_strict'].max()):.2e}); wet/dry flips ref32 vs ref64 per frame: max {int(g['ref32_flips'].max())}")
    assert not fail, "; ".join(fail)


@pytest.mark.parametrize("name,H,W,nums,T,B,rain_max,cum_max,spatial", [
    ("futian", 400, 560, 6, 72, 1, 5.0, 100.0, True),      # BASELINE configs[4] at its own T (futian_scratch.yaml:41-51,66-68: duration 72)
    ("ukea", 52, 120, 6, 36, 1, 10.0, 150.0, True),        # BASELINE configs[4], ukea_scratch.yaml:43-53,68-70
    ("lite128xB8", 128, 128, 3, 6, 8, 60.0, 250.0, False),  # BASELINE configs[2] grid, 8 events per GPU (lite.yaml:31-36)
])
def test_config_size_rollouts_vs_oracle(dev, name, H, W, nums, T, B, rain_max, cum_max, spatial):
    """The other BASELINE shapes at their own size on the benchmarked schedule (graph, three kernel chains) against the CPU oracle:
    Futian 400x560 and UKEA 52x120 with spatial rainfall (C = 15), and the lite 128x128 grid with 8 events per GPU."""
    import urnn_amd.weights as uw
    from urnn_amd.rollout import RolloutEngine
    net, sd = make_net(H, W, 2 * nums + 3, 17, dev)
    ev = uw.make_event(T, H, W, rain_max, seed=23, spatial_rain=spatial, batch=B)
    eng = RolloutEngine(net, H, W, nums, rain_max, cum_max, batch=B, max_frames=T, spatial_rain=spatial, keep_raw=True,
                        overlap=True, use_graph=True)
    frames = eng.rollout(ev).cpu().numpy()
    ref = _oracle_rollout(sd, ev, T, nums, rain_max, cum_max, (name, T))
    yard = _torch_fp32_state_errors(sd, ev, T, nums, rain_max, cum_max, dev, ref[1], (name, T)) if T >= 12 else None
    _check_rollout_vs_oracle(eng, frames, T, ref, name, state_yardstick=yard)


def test_batched_spatial_rollout_vs_oracle(dev):
    """Two events with spatial rainfall, ragged (non-square, tail tiles) grid, against the CPU oracle over T=5 frames."""
    import urnn_amd.weights as uw
    from oracle import oracle as orc
    from urnn_amd.rollout import RolloutEngine
    H, W, nums, T, B = 28, 44, 4, 5, 2
    net, sd = make_net(H, W, 2 * nums + 3, 21, dev)
    ev = uw.make_event(T, H, W, 5.0, seed=13, spatial_rain=True, batch=B)
    eng = RolloutEngine(net, H, W, nums, 5.0, 100.0, batch=B, max_frames=T, spatial_rain=True, keep_raw=True, overlap=True)
    frames = eng.rollout(ev).cpu().numpy()
    ref_frames, ref_states, aux = orc.rollout(orc.OracleNet(sd), ev, T, nums, 5.0, 100.0, want_aux=True)
    raw = eng.out_raw[:T].cpu().numpy()
    cls = eng.out_cls[:T].cpu().numpy()
    ref_raw = np.stack([a["reg_raw"] for a in aux])
    ref_cls = np.stack([a["cls"] for a in aux])
    assert_close(raw, ref_raw, 1e-4, "pre-mask reg")
    assert_close(cls, ref_cls, 1e-4, "cls")
    for k, (got, ref) in enumerate(zip(eng.final_states(), ref_states)):
        assert_close(got.cpu().numpy(), ref, 1e-4, f"final state {k}")
    masked_parity(frames, ref_frames, ref_cls, ref_raw, 1e-4)


def test_short_event_and_long_history(dev):
    """T shorter than the rainfall history window (left zero padding all the way, Dynamic2DFlood.py:347-364)."""
    import urnn_amd.weights as uw
    from oracle import oracle as orc
    from urnn_amd.rollout import RolloutEngine
    H, W, nums, T = 16, 16, 10, 3
    net, sd = make_net(H, W, 2 * nums + 3, 5, dev)
    ev = uw.make_event(T, H, W, 6.0, seed=2)
    eng = RolloutEngine(net, H, W, nums, 6.0, 250.0, max_frames=T, keep_raw=True)
    eng.rollout(ev)
    _, ref_states, aux = orc.rollout(orc.OracleNet(sd), ev, T, nums, 6.0, 250.0, want_aux=True)
    assert_close(eng.out_raw[:T, 0].cpu().numpy(), np.stack([a["reg_raw"][0] for a in aux]), 1e-4, "pre-mask reg")
    for k in range(6):
        assert_close(eng.states[k].cpu().numpy(), ref_states[k], 1e-4, f"state {k}")


def test_grid_not_multiple_of_four_is_rejected(dev):
    """The reference only works on grids that are multiples of 4 (its decoder torch.cat fails otherwise); the HIP path
    raises as well instead of producing garbage."""
    net, _ = make_net(18, 22, 9, 1, dev)
    from urnn_amd.general import initialize_states
    x = torch.zeros(1, 1, 9, 18, 22, device=dev)
    with pytest.raises(RuntimeError):
        net(x, *initialize_states(dev, 18, 22))


def test_large_grid_addressing(dev):
    """1024 x 768 (3x the benchmark plane): 32-bit lane offsets and tile arithmetic stay in range; one step, finite, masked
    output consistent, bit-reproducible."""
    import urnn_amd.weights as uw
    from urnn_amd.rollout import RolloutEngine
    H, W, nums, T = 1024, 768, 3, 2
    net, _ = make_net(H, W, 9, 2, dev)
    ev = uw.make_event(T, H, W, 60.0, seed=4)
    eng = RolloutEngine(net, H, W, nums, 60.0, 250.0, max_frames=T, keep_raw=True)
    a = eng.rollout(ev).clone()
    assert torch.isfinite(a).all()
    assert torch.equal(a, eng.out_raw[:T] * (eng.out_cls[:T] >= 0.5).float())
    for s in eng.states:
        assert torch.isfinite(s).all() and s.abs().max() <= 1.0 + 1e-6
    assert torch.equal(a, eng.rollout(ev))


def test_event_folder_evaluation_end_to_end(dev, tmp_path):
    """`.npy` event folders -> loader -> Inference on the GPU -> metrics (test.py:411-520), against the oracle run over the
    same loader items: scalar- and spatial-rain locations, events shorter than the duration, rank sharding."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import synth_dataset as sd
    from oracle import oracle as orc
    from urnn_amd.dataset import r_MinMaxScaler
    from urnn_amd.evaluate import evaluate_events
    from urnn_amd.events import Dynamic2DFlood
    from urnn_amd.metrics import compute_metrics
    root = str(tmp_path)
    lst = sd.write_tree(root)
    ds = Dynamic2DFlood(root, "test", event_list_file=lst, duration=sd.DURATION)
    nums, rain_max, cum_max, flood_max = 3, 6.0, 30.0, 5000.0
    net, sdict = make_net(sd.H, sd.W, 2 * nums + 3, 5, dev)
    m0, s0, out0 = evaluate_events(net, ds, dev, historical_nums=nums, rain_max=rain_max, cumsum_rain_max=cum_max,
                                   flood_max=flood_max, flood_thres=100.0, rank=0, world_size=2, keep_outputs=True)
    m1, _, out1 = evaluate_events(net, ds, dev, historical_nums=nums, rain_max=rain_max, cumsum_rain_max=cum_max,
                                  flood_max=flood_max, flood_thres=100.0, rank=1, world_size=2, keep_outputs=True)
    assert len(m0) == len(m1) == 3 and not set(m0) & set(m1)
    assert set(s0) == {"mean", "std"}
    onet = orc.OracleNet(sdict)
    outs = {**out0, **out1}
    mets = {**m0, **m1}
    for index in range(len(ds)):
        inputs, target, event_dir = ds.batched(index)
        name = os.path.join(os.path.basename(os.path.dirname(event_dir)), os.path.basename(event_dir))
        ev = {k: v.numpy() for k, v in inputs.items()}
        frames, _, auxs = orc.rollout(onet, ev, sd.DURATION, nums, rain_max, cum_max, want_aux=True)
        ref = frames[:, 0]
        got = outs[name] / flood_max
        # thresholded output: compare away from the wet/dry decision boundary
        cls = np.stack([a["cls"][0] for a in auxs]) if isinstance(auxs[0], dict) and "cls" in auxs[0] else None
        keep = np.abs(cls - 0.5) > 1e-5 if cls is not None else np.ones_like(ref, bool)
        assert_close(np.where(keep, got, 0), np.where(keep, ref, 0), 1e-4, f"evaluate {name}")
        want = compute_metrics(r_MinMaxScaler(ref, max=flood_max, min=0), target[0].numpy(), flood_thres=100.0)
        for k, v in mets[name].items():
            assert v == pytest.approx(want[k], rel=1e-3, abs=1e-6), (name, k)


def test_inference_keeps_one_engine_per_shape(dev):
    """Mixed-resolution event streams (BASELINE configs[4]): alternating shapes reuse their engines (no re-capture) and give
    the same frames as the first visit."""
    import urnn_amd.inference as inf
    import urnn_amd.weights as uw
    shapes = [(16, 24, 3, 5, False), (8, 12, 6, 4, True)]
    nets = [make_net(H, W, 2 * n + 3, 3, dev)[0] for H, W, n, _, _ in shapes]
    events = [uw.make_event(T, H, W, 6.0, seed=7 + i, spatial_rain=sp) for i, (H, W, n, T, sp) in enumerate(shapes)]
    inf._ENGINES.clear()
    first, engines = [], []
    for rnd in range(3):
        for i, (H, W, n, T, sp) in enumerate(shapes):
            out = inf.Inference(nets[i], events[i], dev, historical_nums=n, rain_max=6.0, cumsum_rain_max=100.0,
                                input_height=H, input_width=W)
            assert out.shape == (T, H, W)
            if rnd == 0:
                first.append(out)
                engines.append(next(reversed(inf._ENGINES.values())))
            else:
                assert np.array_equal(out, first[i])
                assert next(reversed(inf._ENGINES.values())) is engines[i]
    assert len(inf._ENGINES) == 2
    inf._ENGINES.clear()


def test_inference_small_then_large_shape(dev):
    """Engines own their scratch: a small grid first, a larger one next (round 1's process-wide pool re-allocated under the
    first engine's captured graph here), then the small one again -- bit-equal to its first visit."""
    import urnn_amd.inference as inf
    import urnn_amd.weights as uw
    shapes = [(8, 12, 3, 5), (64, 64, 3, 5)]
    nets = [make_net(H, W, 2 * n + 3, 3, dev)[0] for H, W, n, _ in shapes]
    events = [uw.make_event(T, H, W, 6.0, seed=11 + i) for i, (H, W, n, T) in enumerate(shapes)]
    inf._ENGINES.clear()

    def run(i):
        H, W, n, T = shapes[i]
        return inf.Inference(nets[i], events[i], dev, historical_nums=n, rain_max=6.0, cumsum_rain_max=100.0, input_height=H, input_width=W)
    small = run(0)
    big = run(1)
    for _ in range(2):
        assert np.array_equal(run(0), small)
        assert np.array_equal(run(1), big)
    inf._ENGINES.clear()


def test_engine_follows_weight_changes(dev):
    """A captured timestep holds pointers to packed copies of the weights: `load_state_dict` on the same net, and a `Trainer`
    that re-homes and updates the parameters, must make the cached engine re-pack and re-capture (evaluate -> train ->
    evaluate on one net)."""
    import urnn_amd.inference as inf
    import urnn_amd.weights as uw
    from urnn_amd.training import Trainer
    H, W, n, T = 16, 16, 3, 4
    ev = uw.make_event(T, H, W, 60.0, seed=5)
    kw = dict(historical_nums=n, rain_max=60.0, cumsum_rain_max=250.0, input_height=H, input_width=W)
    inf._ENGINES.clear()
    net, _ = make_net(H, W, 2 * n + 3, 1, dev)
    a1 = inf.Inference(net, ev, dev, **kw)
    other, sd2 = make_net(H, W, 2 * n + 3, 2, dev)
    want2 = inf.Inference(other, ev, dev, **kw)
    assert not np.array_equal(a1, want2)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd2.items()})
    assert np.array_equal(inf.Inference(net, ev, dev, **kw), want2)
    tr = Trainer(net, H, W, n, 60.0, 250.0, lr=1e-2)
    label = torch.rand(1, T, H, W, device=dev)
    tr.train_window(ev, label[:, :2], 0, 2)
    torch.cuda.synchronize()
    fresh, _ = make_net(H, W, 2 * n + 3, 2, dev)
    fresh.load_state_dict({k: v.detach().clone() for k, v in net.state_dict().items()})
    want3 = inf.Inference(fresh, ev, dev, **kw)
    got3 = inf.Inference(net, ev, dev, **kw)
    assert not np.array_equal(want3, want2) and np.array_equal(got3, want3)
    inf._ENGINES.clear()


def test_repeated_launches_are_bit_identical_at_full_size(dev):
    """Races inside a kernel show up as rare run-to-run differences on identical inputs (the bf16-split k-loop once refilled a
    ring slot whose ds_read had not been waited for: one launch in ~30 differed).  Every cell of the network at its own
    500x500-level plane, 60 launches each, must reproduce the first launch bit for bit."""
    import urnn_amd.weights as uw
    from urnn_amd.rollout import RolloutEngine
    H = W = 500
    nums = 30
    net, _ = make_net(H, W, 2 * nums + 3, 0, dev)
    eng = RolloutEngine(net, H, W, nums, 6.0, 250.0, max_frames=4, use_graph=False)
    eng.load_event(uw.make_event(4, H, W, 6.0, seed=42))
    eng.reset()
    eng.run(2)
    torch.cuda.synchronize()
    enc, dec = net.encoder, net.decoder
    e1, e2, e3, d1, d2, d3 = [s.clone() for s in eng.states]
    a1, a2, a3, u3, u2 = (t.clone() for t in (eng.a1, eng.a2, eng.a3, eng.u3, eng.u2))
    feat = dec.stage1(d3).clone()
    cases = {"enc1": lambda: enc.rnn1.step(a1, None, e1), "enc2": lambda: enc.rnn2.step(a2, None, e2), "enc3": lambda: enc.rnn3.step(a3, None, e3),
             "dec3": lambda: dec.rnn3.step(None, e3, d1), "dec2": lambda: dec.rnn2.step(u3, e2, d2), "dec1": lambda: dec.rnn1.step(u2, e1, d3),
             "stage2": lambda: enc.stage2(e1), "stage3": lambda: enc.stage3(e2), "deconv2": lambda: dec.stage2(d2), "deconv3": lambda: dec.stage3(d1),
             "dec stage1": lambda: dec.stage1(d3), "head": lambda: torch.stack(net.head.run(feat, want_raw=True))}
    launches = int(os.environ.get("URNN_REPEAT_LAUNCHES", "60"))
    failures = []
    for name, fn in cases.items():
        ref = fn().clone()
        bad = [i + 1 for i in range(launches) if not torch.equal(fn(), ref)]
        if bad:
            failures.append(f"{name}: {len(bad)} of {launches} launches differ from the first (first at launch {bad[0]})")
    assert not failures, "; ".join(failures)


def test_scalar_rain_engine_keeps_its_graph_across_dem_ranges(dev):
    """Events of different catchments (different DEM min / max, Dynamic2DFlood.py:231-232) through one scalar-rain engine: the
    DEM is normalised outside the captured timestep, so the graphs are kept -- and the frames equal a fresh engine's."""
    import urnn_amd.weights as uw
    from urnn_amd.rollout import RolloutEngine
    H, W, nums, T = 32, 48, 3, 5
    net, _ = make_net(H, W, 9, 3, dev)
    eng = RolloutEngine(net, H, W, nums, 60.0, 250.0, max_frames=T, overlap=True)
    ev_a = uw.make_event(T, H, W, 60.0, seed=1)
    ev_b = uw.make_event(T, H, W, 60.0, seed=2)
    ev_b["absolute_DEM"] = ev_b["absolute_DEM"] * 0.5 + 700.0
    ev_b["max_DEM"], ev_b["min_DEM"] = ev_b["absolute_DEM"].reshape(1, -1).max(1), ev_b["absolute_DEM"].reshape(1, -1).min(1)
    a = eng.rollout(ev_a).clone()
    graphs = eng._graphs2
    b = eng.rollout(ev_b).clone()
    assert eng._graphs2 is graphs and graphs is not None
    fresh = RolloutEngine(net, H, W, nums, 60.0, 250.0, max_frames=T, overlap=True)
    assert torch.equal(b, fresh.rollout(ev_b)) and not torch.equal(a, b)


def test_weight_beyond_the_f16_range_runs_on_the_exact_instruction(dev):
    """Operand range of the default matrix mode (include/urnn_hip.h; reference checkpoint path test.py:380-408): a layer with a
    weight >= 64 cannot be carried as f16 pieces of w * 2^10 -- the host notices when it packs the layer and launches THAT layer
    under URNN_MATRIX_FP32_MFMA; the rollout stays finite and within 1e-4 of the oracle (which has no range limit)."""
    import urnn_amd.weights as uw
    from oracle import oracle as orc
    from urnn_amd.rollout import RolloutEngine
    H, W, nums, T = 32, 48, 3, 3
    net, sd = make_net(H, W, 2 * nums + 3, 5, dev)
    sd = {k: v.copy() for k, v in sd.items()}
    big = {"decoder.stage2.deconv2_leaky_1.weight": 100.0, "encoder.rnn2.conv1.0.weight": -80.0}
    for k, v in big.items():
        sd[k].reshape(-1)[3] = v
        # keep the activations themselves in range: the big weight multiplies one input channel of one output channel
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    ev = uw.make_event(T, H, W, 60.0, seed=9)
    eng = RolloutEngine(net, H, W, nums, 60.0, 250.0, max_frames=T, keep_raw=True, overlap=True, use_graph=True)
    frames = eng.rollout(ev).cpu().numpy()
    assert net.decoder.stage2._cache.wide and net.encoder.rnn2._cache.wide and not net.encoder.rnn1._cache.wide
    assert np.isfinite(frames).all()
    ref_frames, ref_states, aux = orc.rollout(orc.OracleNet(sd), ev, T, nums, 60.0, 250.0, want_aux=True)
    assert_close(eng.out_raw[:T].cpu().numpy(), np.stack([a["reg_raw"] for a in aux]), 1e-4, "pre-mask reg with out-of-range weights")
    for k, (got, want) in enumerate(zip(eng.final_states(), ref_states)):
        assert_close(got.cpu().numpy(), want, 1e-4, f"state {k} with out-of-range weights")


def test_activation_beyond_the_f16_range_raises(dev):
    """... and an ACTIVATION beyond the pieces' range (|x| >= 2047: here encoder stage 1 is made to output ~3000) turns into inf
    in the next matrix product.  The kernels that fold norm statistics flag it in their workspace's status word; the engine
    reads the word once per event and raises, naming the first layer whose output is not finite -- never NaN maps with rc = 0.
    Under ops.matrix_mode("fp32_mfma") the same net and event roll out finite."""
    import urnn_amd.weights as uw
    from urnn_amd import ops
    from urnn_amd.inference import Inference
    from urnn_amd.rollout import RolloutEngine
    H, W, nums, T = 32, 48, 3, 2
    net, sd = make_net(H, W, 2 * nums + 3, 5, dev)
    sd = {k: v.copy() for k, v in sd.items()}
    sd["encoder.stage1.conv1_leaky_1.bias"][:] = 3000.0
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    ev = uw.make_event(T, H, W, 60.0, seed=9)
    for overlap in (False, True):
        eng = RolloutEngine(net, H, W, nums, 60.0, 250.0, max_frames=T, overlap=overlap, use_graph=True)
        with pytest.raises(FloatingPointError, match="encoder.rnn1"):
            eng.rollout(ev)
    with pytest.raises(FloatingPointError):
        Inference(net, {k: torch.from_numpy(np.asarray(v)) for k, v in ev.items()}, dev, historical_nums=nums, rain_max=60.0,
                  cumsum_rain_max=250.0, input_height=H, input_width=W)
    with ops.matrix_mode("fp32_mfma"):
        eng = RolloutEngine(net, H, W, nums, 60.0, 250.0, max_frames=T, overlap=True, use_graph=True)
        assert torch.isfinite(eng.rollout(ev)).all()


def test_fused_tails_rollout_matches_the_default_schedule(dev):
    """RolloutEngine(fused_tails=True): the end of enc1 / enc2 / dec1 runs together with the stage conv behind it (and dec1's with the
    head's first LayerNorm statistics) -- encoder.py:170-185, decoder.py:150-164, flood_head.py:131-140.  On a grid whose cells do not
    take the cooperative launch (160x200: 500 / 125 blocks at full / half resolution with two chains) the six states stay bit-identical
    to the default schedule over a rollout, the frames within 1e-5 (the head's first statistics are grouped differently)."""
    import urnn_amd.weights as uw
    from urnn_amd.rollout import RolloutEngine
    H, W, nums, T = 160, 200, 3, 4
    net, _ = make_net(H, W, 2 * nums + 3, 21, dev)
    ev = uw.make_event(T, H, W, 60.0, seed=4)
    base = RolloutEngine(net, H, W, nums, 60.0, 250.0, max_frames=T, keep_raw=True, overlap=True, use_graph=True)
    base.rollout(ev)
    tails = RolloutEngine(net, H, W, nums, 60.0, 250.0, max_frames=T, keep_raw=True, overlap=True, use_graph=True, fused_tails=True)
    tails.rollout(ev)
    assert tails._tail_of("enc1") is not None and tails._tail_of("dec1") is not None and base._tail_of("dec1") is None
    for k, (a, b) in enumerate(zip(tails.final_states(), base.final_states())):
        assert torch.equal(a, b), f"state {k} differs with fused tails"
    assert_close(tails.out_raw[:T].cpu().numpy(), base.out_raw[:T].cpu().numpy(), 1e-5, "pre-mask reg with fused tails")
    assert_close(tails.out_cls[:T].cpu().numpy(), base.out_cls[:T].cpu().numpy(), 1e-5, "cls with fused tails")


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,B", [(64, 64, 1), (52, 120, 2), (256, 256, 1), (256, 512, 1), (500, 500, 1)])   # the last two: the stem-epilogue / part0 hand-off branch (B H W >= 131 072) and the cooperative half-resolution cells
def test_c_abi_step_equals_the_one_chain_engine(dev, H, W, B):
    """urnn_step_f32 (include/urnn_hip.h: ED.forward, model.py:65-121, as ONE call for a host that is not Python) against the one-chain
    rollout engine, which enqueues the same launches through the per-module entries: frames and the six states bit-identical over a few
    steps -- on a plane whose cells and head take their cooperative forms, with two events per GPU, and on one whose full-resolution
    cells recompute the reset gate (URNN_PHASE_FUSED_R)."""
    import urnn_amd.weights as uw
    from urnn_amd import ops
    from urnn_amd.rollout import RolloutEngine
    from urnn_amd.step import StepNet
    nums, T = 3, 4
    C = 2 * nums + 3
    net, _ = make_net(H, W, C, 11, dev)
    ev = uw.make_event(T, H, W, 60.0, seed=5, spatial_rain=True, batch=B)
    eng = RolloutEngine(net, H, W, nums, 60.0, 250.0, batch=B, max_frames=T, spatial_rain=True, use_graph=False, keep_raw=True)
    want = eng.rollout(ev).clone()
    want_states = [s.clone() for s in eng.final_states()]
    sn = StepNet(net, B, H, W, C)
    states = [torch.zeros_like(s) for s in want_states]
    masked, cls, raw = (torch.zeros((T, B, H, W), dtype=torch.float32, device=dev) for _ in range(3))
    for t in range(T):
        x_t = ops.preprocess(eng.rain, eng.cumsum, eng.dem, eng.imperv, eng.manhole, eng.dem_min, eng.dem_max, t, nums, 60.0, 250.0)
        sn.step(x_t, states, masked, cls, raw, frame_index=torch.tensor([t], dtype=torch.int32, device=dev))
    assert sn.status() == (0, 0)
    assert torch.equal(masked, want), f"frames differ: max {float((masked - want).abs().max()):.3e}"
    assert torch.equal(raw, eng.out_raw[:T])
    for k, (a, b) in enumerate(zip(states, want_states)):
        assert torch.equal(a, b), f"state {k} differs"
