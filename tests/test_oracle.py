"""Pins the CPU oracle (oracle/urnn_oracle.c) against fixtures produced by the REFERENCE itself
(tests/golden/make_golden.py).  CPU-only; this is what makes the oracle trustworthy as the
checker for the HIP path."""
import numpy as np
import pytest

from conftest import assert_close, masked_parity
from oracle import oracle as orc
import urnn_amd.weights as uw

TOL = 1e-4  # the north-star bar, metric of conftest.rel_err


@pytest.fixture(scope="module")
def kern(golden):
    g = golden("kernels_16x16.npz")
    sd = uw.make_state_dict(int(g["H"]), int(g["W"]), int(g["C"]), seed=int(g["weights_seed"]))
    return g, sd, orc.OracleNet(sd)


@pytest.mark.parametrize("B", [1, 2])
def test_stage_convs(kern, B):
    g, sd, net = kern
    t = f"B{B}"
    for i, pool in ((1, False), (2, True), (3, True)):
        y = orc.stage_conv(g[f"s{i}_in_{t}"], sd[f"encoder.stage{i}.conv{i}_leaky_1.weight"],
                           sd[f"encoder.stage{i}.conv{i}_leaky_1.bias"], pool)
        assert_close(y, g[f"s{i}_out_{t}"], TOL, f"stage{i}")
    y = orc.stage_conv(g[f"dc1_in_{t}"], sd["decoder.stage1.conv3_leaky_1.weight"], sd["decoder.stage1.conv3_leaky_1.bias"], False)
    assert_close(y, g[f"dc1_out_{t}"], TOL, "dec stage1 conv")


@pytest.mark.parametrize("B", [1, 2])
def test_gru_cells(kern, B):
    g, sd, net = kern
    t = f"B{B}"
    for i in (1, 2, 3):
        y = orc.gru_cell(g[f"enc{i}_x_{t}"], None, g[f"enc{i}_h_{t}"], net.enc[i - 1])
        assert_close(y, g[f"enc{i}_out_{t}"], TOL, f"enc cell {i}")
    for i in (3, 2, 1):
        x = None if i == 3 else g[f"dec{i}_x_{t}"]
        y = orc.gru_cell(x, g[f"dec{i}_e_{t}"], g[f"dec{i}_d_{t}"], net.dec[i])
        assert_close(y, g[f"dec{i}_out_{t}"], TOL, f"dec cell {i}")


@pytest.mark.parametrize("B", [1, 2])
def test_deconvs(kern, B):
    g, sd, net = kern
    t = f"B{B}"
    y = orc.deconv2x2(g[f"dc3_in_{t}"], sd["decoder.stage3.deconv1_leaky_1.weight"], sd["decoder.stage3.deconv1_leaky_1.bias"])
    assert_close(y, g[f"dc3_out_{t}"], TOL, "deconv stage3")
    y = orc.deconv2x2(g[f"dc2_in_{t}"], sd["decoder.stage2.deconv2_leaky_1.weight"], sd["decoder.stage2.deconv2_leaky_1.bias"])
    assert_close(y, g[f"dc2_out_{t}"], TOL, "deconv stage2")


@pytest.mark.parametrize("B", [1, 2])
def test_head(kern, B):
    g, sd, net = kern
    t = f"B{B}"
    masked, cls, raw = orc.head(g[f"head_in_{t}"], net.hp)
    assert_close(cls, g[f"head_cls_{t}"], TOL, "cls")
    assert_close(raw, g[f"head_raw_{t}"], TOL, "raw reg")
    masked_parity(masked, g[f"head_masked_{t}"], g[f"head_cls_{t}"], g[f"head_raw_{t}"], TOL)


@pytest.mark.parametrize("B", [1, 2])
def test_full_step(kern, B):
    g, sd, net = kern
    t = f"B{B}"
    st = [g[f"step_state{k}_{t}"] for k in range(6)]
    out, new, _ = net.step(g[f"step_x_{t}"][:, 0], st)
    for k in range(6):
        assert_close(new[k], g[f"step_newstate{k}_{t}"], TOL, f"state {k}")
    # masked output: every pixel whose reference class is not within 1e-5 of the threshold
    excluded = masked_parity(out, g[f"step_reg_{t}"][:, 0], g[f"step_cls_{t}"], g[f"step_raw_{t}"], TOL)
    assert excluded <= 1


def test_preprocess(golden):
    g = golden("preprocess.npz")
    H, W, nums, T = int(g["H"]), int(g["W"]), int(g["nums"]), int(g["T"])
    for spatial in (0, 1):
        for B in (1, 2):
            ev = uw.make_event(T, H, W, float(g["rain_max"]), seed=int(g["event_seed"]), spatial_rain=bool(spatial), batch=B)
            for t in (0, nums - 1, nums, T - 1):
                y = orc.preprocess_inputs(t, ev, nums, float(g["rain_max"]), float(g["cumsum_max"]))
                assert_close(y, g[f"pre_sp{spatial}_B{B}_t{t}"], 1e-6, f"preprocess sp={spatial} B={B} t={t}")


@pytest.mark.parametrize("name", ["rollout_64x64_T30.npz", "rollout_24x40_T8_spatial.npz"])
def test_rollout(golden, name):
    g = golden(name)
    H, W, nums, T = int(g["H"]), int(g["W"]), int(g["nums"]), int(g["T"])
    sd = uw.make_state_dict(H, W, 2 * nums + 3, seed=int(g["weights_seed"]))
    ev = uw.make_event(T, H, W, float(g["rain_max"]), seed=int(g["event_seed"]), spatial_rain=bool(int(g["spatial"])))
    net = orc.OracleNet(sd)
    frames, states, aux = orc.rollout(net, ev, T, nums, float(g["rain_max"]), float(g["cumsum_max"]), want_aux=True)
    every = int(g["every"])
    cls = np.stack([a["cls"][0] for a in aux])[::every]
    raw = np.stack([a["reg_raw"][0] for a in aux])[::every]
    tol = 1e-4  # recurrent accumulation over T steps
    assert_close(cls, g["cls"], tol, "cls over rollout")
    assert_close(raw, g["raw"], tol, "pre-mask reg over rollout")
    for k in range(6):
        assert_close(states[k], g[f"final_state{k}"], tol, f"final state {k}")
    nflip = masked_parity(frames[::every, 0], g["reg"], g["cls"], g["raw"], tol)
    assert nflip < 0.001 * g["reg"].size


# ---- tests/torch_ref.py (the autograd / float64 restatement used by the full-size GPU tests) pinned to the same goldens ----------
def _torch_params(sd, dtype):
    import torch
    return {k: torch.from_numpy(v).to(dtype) for k, v in sd.items()}


@pytest.mark.parametrize("B", [1, 2])
def test_torch_ref_vs_reference(kern, B):
    import torch
    import torch_ref
    g, sd, _ = kern
    t = f"B{B}"
    H, W = int(g["H"]), int(g["W"])
    p = _torch_params(sd, torch.float64)
    st = [torch.from_numpy(g[f"step_state{k}_{t}"]).double() for k in range(6)]
    with torch.no_grad():
        masked, cls, raw, new = torch_ref.step(p, torch.from_numpy(g[f"step_x_{t}"][:, 0]).double(), st, H, W)
    for k in range(6):
        assert_close(new[k].numpy(), g[f"step_newstate{k}_{t}"], TOL, f"torch_ref state {k}")
    ref = g[f"step_reg_{t}"][:, 0]
    assert (np.abs(masked.numpy() - ref) > 1e-4 * max(1e-3, np.abs(ref).max())).mean() < 0.01


def test_torch_ref_window_gradients_vs_reference_autograd(golden):
    """Two timesteps from zero states, the loss, one backward through both steps: every parameter gradient of the restatement
    equals the reference's own autograd (tests/golden/make_train_golden.py)."""
    import torch
    import torch_ref
    g = golden("train_window_16x16.npz")
    H, W, nums, steps = int(g["win_H"]), int(g["win_W"]), int(g["win_nums"]), int(g["win_steps"])
    sd = uw.make_state_dict(H, W, 2 * nums + 3, seed=int(g["win_weights_seed"]))
    p = {k: v.requires_grad_(True) for k, v in _torch_params(sd, torch.float64).items()}
    ev = uw.make_event(steps + 1, H, W, float(g["win_rain_max"]), seed=int(g["win_event_seed"]))
    states = [torch.from_numpy(s).double() for s in orc.zero_states(1, H, W)]
    outs = []
    for t in range(steps):
        x = torch.from_numpy(orc.preprocess_inputs(t, ev, nums, float(g["win_rain_max"]), float(g["win_cumsum_max"]))[:, 0]).double()
        masked, _, _, states = torch_ref.step(p, x, states, H, W)
        outs.append(masked)
    reg = torch.stack(outs, dim=1)
    assert_close(reg.detach().numpy(), g["win_reg"], TOL, "window outputs")
    loss = torch_ref.wmse(reg, torch.from_numpy(g["win_target"]).double())
    assert float(loss) == pytest.approx(float(g["win_loss_reg"]), rel=1e-4)
    loss.backward()
    for name in sd:
        if not int(g[f"win_hasgrad_{name}"]):
            continue
        ref = g[f"win_grad_{name}"]
        grad = p[name].grad            # None: cut off by the wet/dry comparison (the reference stores zeros there)
        got = np.zeros(ref.shape) if grad is None else grad.numpy().reshape(ref.shape)
        scale = np.abs(ref).max()
        if scale == 0.0:
            assert np.abs(got).max() == 0.0, name
        else:
            assert np.abs(got - ref).max() / scale <= 1e-3, name


# ---- the oracle pinned to the REFERENCE ITSELF at the headline size and horizon -------------------------------------
def _sub_err(a, b, plane_max, floor_frac):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float((np.abs(a - b) / np.maximum(np.abs(b), floor_frac * max(float(plane_max), 1e-30))).max())


def test_oracle_trace_vs_reference_trace_whole_event(golden):
    """500x500, C = 63, all 360 frames (test.py:352-371, model.py:65-121).  Two committed traces of the SAME seeded event on the SAME
    pixels: tests/golden/whole_event_500x500_T360.npz -- the C oracle (oracle/urnn_oracle.c), generated by make_whole_event_trace.py --
    and tests/golden/reference_trace_500x500_T360.npz -- the reference's own modules in float32 and as a float64 copy, generated in the
    build container by make_reference_trace.py.  Before round 6 the oracle was pinned to the reference by fixtures of <= 30 steps and
    <= 64x64 only.  On every sampled frame (4096 random pixels + the oracle trace's 2048 adversarial ones) and on the final-state
    subsets the oracle must sit within 1e-4 of the reference's float64 result under the tests' floor (0.1 x plane max; measured
    4.0e-5), and under SURVEY 8c's strict floor (1e-3 x plane max) within 3 x what the reference's own float32 evaluation is away from
    its float64 one on that frame's full plane (measured: 1.9 x at worst; the float32 reference itself reaches 3.6e-3 there)."""
    ref, orc_t = golden("reference_trace_500x500_T360.npz"), golden("whole_event_500x500_T360.npz")
    for k in ("H", "W", "nums", "T", "weights_seed", "event_seed"):
        assert int(ref[k]) == int(orc_t[k]), k
    assert np.array_equal(ref["frames"], orc_t["frames"]) and np.array_equal(ref["pixels"], orc_t["pixels"])
    worst, worst_strict, w32 = 0.0, 0.0, 0.0
    for i, t in enumerate(ref["frames"]):
        rmax, cmax = ref["ref64_raw_plane_max"][i], ref["ref64_cls_plane_max"][i]
        pairs = [(orc_t["oracle_raw"][i], ref["r64_raw"][i], rmax), (orc_t["adv_oracle_raw"][i], ref["oracle_adv_ref64_raw"][i], rmax),
                 (orc_t["oracle_cls"][i], ref["r64_cls"][i], cmax), (orc_t["adv_oracle_cls"][i], ref["oracle_adv_ref64_cls"][i], cmax)]
        e = max(_sub_err(a, b, m, 0.1) for a, b, m in pairs)
        es_reg = max(_sub_err(a, b, m, 1e-3) for a, b, m in pairs[:2])
        es_cls = max(_sub_err(a, b, m, 1e-3) for a, b, m in pairs[2:])
        worst = max(worst, e)
        worst_strict = max(worst_strict, es_reg / max(1e-4, 3.0 * float(ref["ref32_reg_err_full_strict"][i])), es_cls / max(1e-4, 3.0 * float(ref["ref32_cls_err_full_strict"][i])))
        w32 = max(w32, _sub_err(ref["r32_raw"][i], ref["r64_raw"][i], rmax, 0.1))
        assert e <= 1e-4, f"frame {int(t)}: oracle vs reference fp64 {e:.2e}"
    for k in range(6):
        assert np.array_equal(ref[f"state{k}_idx"], orc_t[f"state{k}_idx"])
        es = _sub_err(orc_t[f"state{k}_oracle"], ref[f"state{k}_ref64"], ref["state_plane_max"][k], 0.1)
        assert es <= 1e-4, f"final state {k}: {es:.2e}"
        worst = max(worst, es)
    print(f"C oracle vs the reference's float64 rollout, 121 of 360 frames at 500x500: worst {worst:.2e} (floor 0.1 x max; the reference's own float32 on the "
          f"random pixels: {w32:.2e}); strict floor: worst error / (3 x the reference's float32 full-plane error) = {worst_strict:.2f}")
    assert worst_strict <= 1.0
