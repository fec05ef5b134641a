"""GPU parity tests of the HIP kernels (through the C ABI) against the reference-generated goldens and
against the CPU oracle on fresh seeded inputs.  Run on the MI355X box: pytest -m gpu."""
import numpy as np
import pytest
import torch

from conftest import assert_close, masked_parity

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def gnet(golden, dev):
    import urnn_amd.weights as uw
    from urnn_amd.net_config import load_net_config
    from urnn_amd.networks import ED, get_network_params
    g = golden("kernels_16x16.npz")
    H, W, C = int(g["H"]), int(g["W"]), int(g["C"])
    sd = uw.make_state_dict(H, W, C, seed=int(g["weights_seed"]))
    ep, dp = get_network_params(False, H, W, C, load_net_config())
    net = ED(False, ep, dp, 0.5, False, H, W)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return g, net.to(dev).eval(), sd


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("B", [1, 2])
def test_stage_convs_vs_reference(gnet, dev, B):
    g, net, _ = gnet
    t = f"B{B}"
    for i in (1, 2, 3):
        y = getattr(net.encoder, f"stage{i}")(T(g[f"s{i}_in_{t}"], dev))
        assert_close(y.cpu().numpy(), g[f"s{i}_out_{t}"], TOL, f"encoder stage{i}")
    y = net.decoder.stage1(T(g[f"dc1_in_{t}"], dev))
    assert_close(y.cpu().numpy(), g[f"dc1_out_{t}"], TOL, "decoder stage1 conv")


@pytest.mark.parametrize("B", [1, 2])
def test_deconvs_vs_reference(gnet, dev, B):
    g, net, _ = gnet
    t = f"B{B}"
    assert_close(net.decoder.stage3(T(g[f"dc3_in_{t}"], dev)).cpu().numpy(), g[f"dc3_out_{t}"], TOL, "deconv stage3")
    assert_close(net.decoder.stage2(T(g[f"dc2_in_{t}"], dev)).cpu().numpy(), g[f"dc2_out_{t}"], TOL, "deconv stage2")


@pytest.mark.parametrize("B", [1, 2])
def test_gru_cells_vs_reference(gnet, dev, B):
    g, net, _ = gnet
    t = f"B{B}"
    for i in (1, 2, 3):
        rnn = getattr(net.encoder, f"rnn{i}")
        y = rnn(T(g[f"enc{i}_x_{t}"], dev)[None], T(g[f"enc{i}_h_{t}"], dev))[0]
        assert_close(y.cpu().numpy(), g[f"enc{i}_out_{t}"], TOL, f"encoder cell {i}")
    for i in (3, 2, 1):
        rnn = getattr(net.decoder, f"rnn{i}")
        st = torch.cat((T(g[f"dec{i}_e_{t}"], dev), T(g[f"dec{i}_d_{t}"], dev)), 1)
        x = None if i == 3 else T(g[f"dec{i}_x_{t}"], dev)[None]
        y = rnn(x, st)[0]
        assert_close(y.cpu().numpy(), g[f"dec{i}_out_{t}"], TOL, f"decoder cell {i}")


@pytest.mark.parametrize("I,F,skip,H,W,B,with_x", [
    (5, 32, 0, 12, 20, 2, True),      # odd input width (zero-weight pad row), one norm group per gate
    (16, 128, 1, 10, 10, 1, True),    # F=128: candidate GEMM in two n-groups of two blocks
    (7, 128, 0, 6, 14, 3, True),
    (96, 96, 1, 9, 7, 2, False),      # x == None (decoder stage 3), odd plane: dword DMA tiles
    (33, 64, 1, 40, 36, 1, True),     # 16-byte DMA tiles, odd I in a skip cell
    (16, 64, 0, 64, 66, 2, True),     # 128-pixel tiles with a ragged last tile
])
def test_gru_cell_shapes_vs_oracle(dev, I, F, skip, H, W, B, with_x):
    """The cell through the C ABI for widths and planes the published network never builds, against the oracle."""
    from oracle import oracle as orc
    from urnn_amd import ops
    rs = np.random.RandomState(1000 + I + F + H)
    K = I + (2 * F if skip else F)
    p = {"W1": rs.normal(0, 1 / np.sqrt(K), (2 * F, K)).astype(np.float32), "b1": rs.normal(0, 0.1, 2 * F).astype(np.float32),
         "g1": rs.uniform(0.5, 1.5, 2 * F).astype(np.float32), "be1": rs.normal(0, 0.1, 2 * F).astype(np.float32),
         "W2": rs.normal(0, 1 / np.sqrt(K), (F, K)).astype(np.float32), "b2": rs.normal(0, 0.1, F).astype(np.float32),
         "g2": rs.uniform(0.5, 1.5, F).astype(np.float32), "be2": rs.normal(0, 0.1, F).astype(np.float32)}
    x = rs.normal(0, 1, (B, I, H, W)).astype(np.float32) if with_x else None
    e = rs.normal(0, 1, (B, F, H, W)).astype(np.float32) if skip else None
    h = rs.normal(0, 1, (B, F, H, W)).astype(np.float32)
    want = orc.gru_cell(x, e, h, p)
    packed = ops.pack_gru(T(p["W1"].reshape(2 * F, K, 1, 1), dev), T(p["b1"], dev), T(p["W2"].reshape(F, K, 1, 1), dev),
                          T(p["b2"], dev), I, F, bool(skip))
    got = ops.gru_cell(None if x is None else T(x, dev), None if e is None else T(e, dev), T(h, dev), packed,
                       T(p["g1"], dev), T(p["be1"], dev), T(p["g2"], dev), T(p["be2"], dev), I)
    assert_close(got.cpu().numpy(), want, TOL, f"gru cell I={I} F={F} skip={skip} {H}x{W} B={B}")


def test_norm_statistics_with_large_means(dev):
    """GroupNorm (ConvRNN.py:94-104) / LayerNorm (network_blocks.py:93) statistics when the pre-norm activations sit far from
    zero -- |mean| / std = 30, as a trained checkpoint's biases can make them -- against the oracle's two-pass double
    statistics.  Raw fp32 (sum, sum of squares) partials lose the variance to cancellation here (rstd off by ~5e-4); the
    kernels' partials are centred per tile (urnn_common.h tile_x2)."""
    from oracle import oracle as orc
    from urnn_amd import ops
    rs = np.random.RandomState(77)
    I, F, H, W, B = 16, 64, 96, 104, 2                    # 9984 pixels: 78 tiles of 128 pixels per (sample, group)
    K = I + F
    p = {"W1": rs.normal(0, 1 / np.sqrt(K), (2 * F, K)).astype(np.float32), "b1": (30.0 + rs.normal(0, 0.1, 2 * F)).astype(np.float32),
         "g1": rs.uniform(0.5, 1.5, 2 * F).astype(np.float32), "be1": rs.normal(0, 0.1, 2 * F).astype(np.float32),
         "W2": rs.normal(0, 1 / np.sqrt(K), (F, K)).astype(np.float32), "b2": (-30.0 + rs.normal(0, 0.1, F)).astype(np.float32),
         "g2": rs.uniform(0.5, 1.5, F).astype(np.float32), "be2": rs.normal(0, 0.1, F).astype(np.float32)}
    x = rs.normal(0, 1, (B, I, H, W)).astype(np.float32)
    h = rs.normal(0, 0.5, (B, F, H, W)).astype(np.float32)
    want = orc.gru_cell(x, None, h, p)
    packed = ops.pack_gru(T(p["W1"].reshape(2 * F, K, 1, 1), dev), T(p["b1"], dev), T(p["W2"].reshape(F, K, 1, 1), dev), T(p["b2"], dev),
                          I, F, False)
    got = ops.gru_cell(T(x, dev), None, T(h, dev), packed, T(p["g1"], dev), T(p["be1"], dev), T(p["g2"], dev), T(p["be2"], dev), I)
    assert_close(got.cpu().numpy(), want, TOL, "cell with pre-norm |mean|/std = 30")

    C = 16
    conv_w = (0.5 * np.eye(C)[None].repeat(5, 0) + rs.normal(0, 0.02, (5, C, C))).astype(np.float32)
    hp = {"conv_w": conv_w, "ln_w": rs.uniform(0.5, 1.5, (5, C, H, W)).astype(np.float32), "ln_b": rs.normal(0, 0.1, (5, C, H, W)).astype(np.float32),
          "cls_w": rs.normal(0, 0.3, (1, C)).astype(np.float32), "cls_b": rs.normal(0, 0.1, 1).astype(np.float32),
          "reg_w": rs.normal(0, 0.3, (1, C)).astype(np.float32), "reg_b": rs.normal(0, 0.1, 1).astype(np.float32)}
    feat = (30.0 + rs.normal(0, 0.5, (B, C, H, W))).astype(np.float32)       # stems conv output: mean ~15, std ~0.5
    masked, cls, raw = orc.head(feat, hp)
    gm, gc, gr = ops.head(T(feat, dev), T(hp["conv_w"], dev), T(hp["ln_w"], dev), T(hp["ln_b"], dev), T(hp["cls_w"].reshape(-1), dev),
                          T(hp["cls_b"], dev), T(hp["reg_w"].reshape(-1), dev), T(hp["reg_b"], dev), 0.5, want_raw=True)
    assert_close(gc.cpu().numpy(), cls, TOL, "head cls with pre-norm |mean|/std = 30")
    assert_close(gr.cpu().numpy(), raw, TOL, "head raw with pre-norm |mean|/std = 30")


def test_gru_cell_in_place(gnet, dev):
    g, net, _ = gnet
    x, h = T(g["enc1_x_B1"], dev), T(g["enc1_h_B1"], dev)
    ref = net.encoder.rnn1.step(x, None, h)
    hh = h.clone()
    net.encoder.rnn1.step(x, None, hh, out=hh)
    assert torch.equal(ref, hh)


@pytest.mark.parametrize("B", [1, 2])
def test_head_vs_reference(gnet, dev, B):
    g, net, _ = gnet
    t = f"B{B}"
    masked, cls, raw = net.head.run(T(g[f"head_in_{t}"], dev), want_raw=True)
    assert_close(cls.cpu().numpy(), g[f"head_cls_{t}"], TOL, "cls")
    assert_close(raw.cpu().numpy(), g[f"head_raw_{t}"], TOL, "raw reg")
    masked_parity(masked.cpu().numpy(), g[f"head_masked_{t}"], g[f"head_cls_{t}"], g[f"head_raw_{t}"], TOL)
    out = net.head(T(g[f"head_in_{t}"], dev)[None])
    assert out.shape == (1, B, 2, 16, 16)
    assert torch.equal(out[0, :, 1], cls)


@pytest.mark.parametrize("B", [1, 2])
def test_full_step_vs_reference(gnet, dev, B):
    g, net, _ = gnet
    t = f"B{B}"
    res = net(T(g[f"step_x_{t}"], dev), *[T(g[f"step_state{k}_{t}"], dev) for k in range(6)])
    assert res[0].shape == g[f"step_reg_{t}"].shape
    for k in range(6):
        assert_close(res[1 + k].cpu().numpy(), g[f"step_newstate{k}_{t}"], TOL, f"new state {k}")
    # masked depth: every pixel whose reference class is not within 1e-5 of the wet/dry threshold, no pixel budget
    excluded = masked_parity(res[0].cpu().numpy()[:, 0], g[f"step_reg_{t}"][:, 0], g[f"step_cls_{t}"], g[f"step_raw_{t}"], TOL)
    assert excluded <= 1


def test_preprocess_vs_reference(golden, dev):
    import urnn_amd.weights as uw
    from urnn_amd.dataset import preprocess_inputs
    g = golden("preprocess.npz")
    H, W, nums, Tn = int(g["H"]), int(g["W"]), int(g["nums"]), int(g["T"])
    for spatial in (0, 1):
        for B in (1, 2):
            ev = uw.make_event(Tn, H, W, float(g["rain_max"]), seed=int(g["event_seed"]), spatial_rain=bool(spatial), batch=B)
            for t in (0, nums - 1, nums, Tn - 1):
                y = preprocess_inputs(t, ev, dev, nums=nums, rain_max=float(g["rain_max"]), cumsum_rain_max=float(g["cumsum_max"]))
                ref = g[f"pre_sp{spatial}_B{B}_t{t}"]
                assert y.shape == ref.shape
                assert_close(y.cpu().numpy(), ref, 1e-6, f"preprocess sp={spatial} B={B} t={t}")


# ---- ragged / odd shapes against the oracle --------------------------------------------------------------
@pytest.mark.parametrize("H,W,B", [(20, 12, 1), (36, 52, 2), (100, 60, 1), (8, 8, 3), (4, 4, 1)])
def test_full_step_vs_oracle_shapes(dev, H, W, B):
    import urnn_amd.weights as uw
    from oracle import oracle as orc
    from urnn_amd.net_config import load_net_config
    from urnn_amd.networks import ED, get_network_params
    C = 15
    sd = uw.make_state_dict(H, W, C, seed=H * 100 + W)
    ep, dp = get_network_params(False, H, W, C, load_net_config())
    net = ED(False, ep, dp, 0.5, False, H, W)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net = net.to(dev).eval()
    rs = np.random.RandomState(5)
    x = (0.5 * rs.standard_normal((B, 1, C, H, W))).astype(np.float32)
    st = [(0.5 * rs.standard_normal(s.shape)).astype(np.float32) for s in orc.zero_states(B, H, W)]
    res = net(T(x, dev), *[T(s, dev) for s in st])
    ref_out, ref_st, aux = orc.OracleNet(sd).step(x[:, 0], st, want_aux=True)
    for k in range(6):
        assert_close(res[1 + k].cpu().numpy(), ref_st[k], TOL, f"state {k} at {H}x{W} B={B}")
    masked_parity(res[0].cpu().numpy()[:, 0], ref_out, aux["cls"], aux["reg_raw"], TOL)


def test_argument_errors(dev):
    from urnn_amd import ops
    from urnn_amd._lib import UrnnError
    x = torch.zeros(1, 8, 4, 4, device=dev)
    with pytest.raises(RuntimeError):
        ops.stage_conv(torch.zeros(1, 8, 4, 4), torch.zeros(10), 16, False)       # CPU tensor: no fallback
    packed = ops.pack_conv(torch.zeros(16, 8, 1, 1, device=dev), torch.zeros(16, device=dev))
    with pytest.raises(UrnnError):
        ops.stage_conv(x, packed, 16, True, out=torch.zeros(1, 16, 2, 2, device=dev)[:, :, :, 1:].contiguous()[..., :0].new_zeros(3)[1:])


@pytest.mark.parametrize("I,F,skip,H,W,B,with_x", [
    (96, 96, 0, 125, 125, 1, True),    # encoder stage 3 at 500x500: 245 blocks (needs every CU but eleven), odd plane
    (96, 96, 1, 125, 125, 1, False),   # decoder stage 3: x == None, skip
    (16, 64, 0, 64, 64, 1, True),      # the 64x64 config's full-resolution cell: 64 blocks of 8 waves
    (96, 96, 1, 32, 32, 2, True),      # K = 288 (the largest panel: 72 KB), two samples
    (64, 96, 0, 13, 15, 3, True),      # ragged: 195 pixels per sample, a partial last tile and an empty pixel block
    (96, 64, 1, 26, 60, 1, True),      # the 52x120 config's half-resolution... decoder cell shape with F = 64
])
def test_cooperative_cell_equals_three_kernels(dev, I, F, skip, H, W, B, with_x):
    """URNN_PHASE_COOP (ConvRNN.py:111-194): the whole cell of a small plane as one launch -- gate GEMM | grid barrier | candidate GEMM
    | grid barrier | blend, raw gates and candidate kept in registers -- against the three-kernel cell: bit-identical (same pieces,
    same MFMA order, same summation orders of the statistics), in place and out of place, over repeated launches (the barrier words
    are reused), and within 1e-4 of the oracle."""
    from oracle import oracle as orc
    from urnn_amd import ops
    from urnn_amd._lib import lib
    rs = np.random.RandomState(77 + I + F + H)
    K = I + (2 * F if skip else F)
    p = {"W1": rs.normal(0, 1 / np.sqrt(K), (2 * F, K)).astype(np.float32), "b1": rs.normal(0, 0.1, 2 * F).astype(np.float32),
         "g1": rs.uniform(0.5, 1.5, 2 * F).astype(np.float32), "be1": rs.normal(0, 0.1, 2 * F).astype(np.float32),
         "W2": rs.normal(0, 1 / np.sqrt(K), (F, K)).astype(np.float32), "b2": rs.normal(0, 0.1, F).astype(np.float32),
         "g2": rs.uniform(0.5, 1.5, F).astype(np.float32), "be2": rs.normal(0, 0.1, F).astype(np.float32)}
    x = rs.normal(0, 1, (B, I, H, W)).astype(np.float32) if with_x else None
    e = rs.normal(0, 1, (B, F, H, W)).astype(np.float32) if skip else None
    h = rs.normal(0, 1, (B, F, H, W)).astype(np.float32)
    blocks = lib().urnn_gru_cell_coop_blocks(B, I, F, H, W, int(skip), int(with_x))
    assert blocks == B * ((H * W + 63) // 64), f"expected a cooperative launch for this shape, the library plans {blocks} blocks"
    packed = ops.pack_gru(T(p["W1"].reshape(2 * F, K, 1, 1), dev), T(p["b1"], dev), T(p["W2"].reshape(F, K, 1, 1), dev),
                          T(p["b2"], dev), I, F, bool(skip))
    args = (None if x is None else T(x, dev), None if e is None else T(e, dev))
    aff = (T(p["g1"], dev), T(p["be1"], dev), T(p["g2"], dev), T(p["be2"], dev))
    ws = ops.workspace(ops.gru_cell_workspace_bytes(B, F, H, W), dev)
    three = ops.gru_cell(*args, T(h, dev), packed, *aff, I, phases=ops.PHASE_ALL, ws=ws)
    one = ops.gru_cell(*args, T(h, dev), packed, *aff, I, phases=ops.PHASE_ALL | ops.PHASE_COOP, ws=ws)
    torch.cuda.synchronize()
    assert ops.workspace_status(ws) == 0
    if not torch.equal(one, three):
        d = (one - three).abs()
        idx = [int(v) for v in np.unravel_index(int(d.argmax()), d.shape)]
        nbad = int((d > 0).sum())
        bad_bc = sorted({(int(a), int(c)) for a, c in torch.nonzero(d.amax(dim=(2, 3)) > 0).tolist()})[:12]
        pytest.fail(f"cooperative cell differs from the three kernels: max {float(d.max()):.3e} at (b, ch, y, x) = {idx}, {nbad} of {d.numel()} "
                    f"values differ; (sample, channel) pairs with differences: {bad_bc}")
    assert_close(one.cpu().numpy(), orc.gru_cell(x, e, h, p), TOL, "cooperative cell vs oracle")
    hh = T(h, dev)
    for _ in range(25):                                   # in place, barrier state reused launch after launch
        hh.copy_(T(h, dev))
        ops.gru_cell(*args, hh, packed, *aff, I, out=hh, phases=ops.PHASE_ALL | ops.PHASE_COOP, ws=ws)
        assert torch.equal(hh, three)
    assert ops.workspace_status(ws) == 0


@pytest.mark.parametrize("I,F,skip,H,W,B", [
    (64, 96, 0, 250, 250, 1),     # encoder stage 2 at 500x500: 977 tiles, 245 blocks of four (the last one holds a single, partial tile)
    (96, 96, 1, 250, 250, 1),     # decoder stage 2: K = 288, eighteen 16-k groups
    (96, 96, 1, 200, 280, 1),     # Futian's half resolution: 875 tiles
    (96, 64, 1, 150, 152, 1),     # F = 64: two channel blocks per role, 357 tiles
    (64, 96, 0, 110, 110, 2),     # two samples of 190 tiles: a block straddles the samples; the samples' last tile has 4 valid pixels
])
def test_cooperative_tiles_cell_vs_three_kernels_and_oracle(dev, I, F, skip, H, W, B):
    """URNN_PHASE_COOP on a half-resolution plane (ConvRNN.py:111-194; urnn_coop_tiles.hip): the whole cell as one cooperative launch of
    persistent blocks with four 64-pixel tiles each -- raw gates and candidate stay in the accumulators between the phases, two grid
    barriers for the two GroupNorms.  Same f16-piece arithmetic and MFMA order as the three-kernel cell; the GroupNorm partials are
    summed in another order inside a tile, so h' agrees with the three kernels to rounding (<= 2e-6 of max |h'|) rather than to the bit;
    within 1e-4 of the oracle; in place, bit-stable over repeated launches (the barrier words are reused); status word clean."""
    from oracle import oracle as orc
    from urnn_amd import ops
    from urnn_amd._lib import lib
    rs = np.random.RandomState(91 + I + F + H)
    K = I + (2 * F if skip else F)
    p = {"W1": rs.normal(0, 1 / np.sqrt(K), (2 * F, K)).astype(np.float32), "b1": rs.normal(0, 0.1, 2 * F).astype(np.float32),
         "g1": rs.uniform(0.5, 1.5, 2 * F).astype(np.float32), "be1": rs.normal(0, 0.1, 2 * F).astype(np.float32),
         "W2": rs.normal(0, 1 / np.sqrt(K), (F, K)).astype(np.float32), "b2": rs.normal(0, 0.1, F).astype(np.float32),
         "g2": rs.uniform(0.5, 1.5, F).astype(np.float32), "be2": rs.normal(0, 0.1, F).astype(np.float32)}
    x = rs.normal(0, 1, (B, I, H, W)).astype(np.float32)
    e = rs.normal(0, 1, (B, F, H, W)).astype(np.float32) if skip else None
    h = rs.normal(0, 1, (B, F, H, W)).astype(np.float32)
    tiles = B * ((H * W + 63) // 64)
    blocks = lib().urnn_gru_cell_coop_blocks(B, I, F, H, W, int(skip), 1)
    assert blocks == (tiles + 3) // 4, f"expected the four-tiles-per-block cooperative launch ({(tiles + 3) // 4} blocks), the library plans {blocks}"
    packed = ops.pack_gru(T(p["W1"].reshape(2 * F, K, 1, 1), dev), T(p["b1"], dev), T(p["W2"].reshape(F, K, 1, 1), dev),
                          T(p["b2"], dev), I, F, bool(skip))
    args = (T(x, dev), None if e is None else T(e, dev))
    aff = (T(p["g1"], dev), T(p["be1"], dev), T(p["g2"], dev), T(p["be2"], dev))
    ws = ops.workspace(ops.gru_cell_workspace_bytes(B, F, H, W), dev)
    three = ops.gru_cell(*args, T(h, dev), packed, *aff, I, phases=ops.PHASE_ALL, ws=ws)
    one = ops.gru_cell(*args, T(h, dev), packed, *aff, I, phases=ops.PHASE_ALL | ops.PHASE_COOP, ws=ws)
    torch.cuda.synchronize()
    assert ops.workspace_status(ws) == 0
    d = float((one - three).abs().max()) / float(three.abs().max())
    print(f"cooperative tiles vs three kernels: max |dh'| / max |h'| = {d:.2e}; bit-identical: {torch.equal(one, three)}")
    assert d <= 2e-6
    ref = orc.gru_cell(x, e, h, p)
    assert_close(one.cpu().numpy(), ref, TOL, "cooperative tiles cell vs oracle")
    hh = T(h, dev)
    for _ in range(10):                                   # in place, barrier state reused launch after launch
        hh.copy_(T(h, dev))
        ops.gru_cell(*args, hh, packed, *aff, I, out=hh, phases=ops.PHASE_ALL | ops.PHASE_COOP, ws=ws)
        assert torch.equal(hh, one)
    assert ops.workspace_status(ws) == 0


def test_cooperative_cell_is_not_planned_for_large_planes_or_other_modes(dev):
    from urnn_amd import ops
    from urnn_amd._lib import lib
    L = lib()
    assert L.urnn_gru_cell_coop_blocks(1, 64, 96, 250, 250, 0, 1) == 245        # 977 tiles: four per block (urnn_coop_tiles.hip)
    assert L.urnn_gru_cell_coop_blocks(1, 16, 64, 500, 500, 0, 1) == 0
    assert L.urnn_gru_cell_coop_blocks(1, 16, 128, 64, 64, 0, 1) == 0           # F = 128: sixteen waves of gates
    assert L.urnn_gru_cell_coop_blocks(1, 96, 96, 125, 125, 0, 1) == 245
    with ops.matrix_mode("fp32_mfma"):
        assert L.urnn_gru_cell_coop_blocks(1, 96, 96, 125, 125, 0, 1) == 0      # the exact-fp32 mode keeps its own kernels


@pytest.mark.parametrize("I,F,skip,H,W,B,Cout,pool,with_head", [
    (16, 64, 0, 64, 66, 2, 64, True, False),     # encoder stage 1 -> stage-2 conv + pool (the pooled plane's last block is partial)
    (64, 96, 0, 30, 34, 1, 96, True, False),     # encoder stage 2 -> stage-3 conv + pool: three output blocks, twelve waves
    (96, 64, 1, 37, 41, 2, 16, False, True),     # decoder stage 1 -> final conv + the head's first LayerNorm statistics, odd plane
    (96, 64, 1, 256, 260, 1, 16, False, True),   # ... on a plane large enough for the fused-reset-gate cell (66 560 pixels)
])
def test_cell_tail_equals_blend_then_conv(dev, I, F, skip, H, W, B, Cout, pool, with_head):
    """urnn_gru_cell_tail_f32 (encoder.py:170-185, decoder.py:150-164, flood_head.py:131-140): GroupNorm finalize + blend + the stage
    conv that consumes the new state in ONE launch, against the cell followed by urnn_stage_conv_f32: the new state and the conv
    output bit-identical; with the head's statistics taken in the same launch, the head's outputs within 1e-5 of the four-pass head
    (its first LayerNorm's sums are grouped differently) and both within 1e-4 of the oracle."""
    from oracle import oracle as orc
    from urnn_amd import ops
    rs = np.random.RandomState(500 + I + F + H)
    K = I + (2 * F if skip else F)
    p = {"W1": rs.normal(0, 1 / np.sqrt(K), (2 * F, K)).astype(np.float32), "b1": rs.normal(0, 0.1, 2 * F).astype(np.float32),
         "g1": rs.uniform(0.5, 1.5, 2 * F).astype(np.float32), "be1": rs.normal(0, 0.1, 2 * F).astype(np.float32),
         "W2": rs.normal(0, 1 / np.sqrt(K), (F, K)).astype(np.float32), "b2": rs.normal(0, 0.1, F).astype(np.float32),
         "g2": rs.uniform(0.5, 1.5, F).astype(np.float32), "be2": rs.normal(0, 0.1, F).astype(np.float32)}
    x = rs.normal(0, 1, (B, I, H, W)).astype(np.float32)
    e = rs.normal(0, 1, (B, F, H, W)).astype(np.float32) if skip else None
    h = rs.normal(0, 1, (B, F, H, W)).astype(np.float32)
    wc = rs.normal(0, 1 / np.sqrt(F), (Cout, F, 1, 1)).astype(np.float32)
    bc = rs.normal(0, 0.1, Cout).astype(np.float32)
    assert ops.gru_cell_tail_applies(B, F, H, W, Cout, pool)
    packed = ops.pack_gru(T(p["W1"].reshape(2 * F, K, 1, 1), dev), T(p["b1"], dev), T(p["W2"].reshape(F, K, 1, 1), dev),
                          T(p["b2"], dev), I, F, bool(skip))
    cpk = ops.pack_conv(T(wc, dev), T(bc, dev))
    args = (T(x, dev), None if e is None else T(e, dev))
    aff = (T(p["g1"], dev), T(p["be1"], dev), T(p["g2"], dev), T(p["be2"], dev))
    for flags in (0, ops.PHASE_FUSED_R):
        want_h = ops.gru_cell(*args, T(h, dev), packed, *aff, I, phases=ops.PHASE_ALL | flags)
        want_c = ops.stage_conv(want_h, cpk, Cout, pool)
        hw = T(rs.normal(0, 0.25, (16, 16)).astype(np.float32), dev) if with_head else None
        part0 = ops.head_tail_partial(B, H, W, dev) if with_head else None
        hh = T(h, dev)
        got_h, got_c = ops.gru_cell_tail(*args, hh, packed, *aff, I, cpk, Cout, pool, out=hh, phases=ops.PHASE_ALL | flags,
                                         head_w=hw, head_partial0=part0)                  # in place, like the rollout engine
        assert torch.equal(got_h, want_h), f"new state differs: max {float((got_h - want_h).abs().max()):.3e}"
        assert torch.equal(got_c, want_c), f"conv output differs: max {float((got_c - want_c).abs().max()):.3e}"
    ref_h = orc.gru_cell(x, e, h, p)
    assert_close(got_h.cpu().numpy(), ref_h, TOL, "tail: new state vs oracle")
    assert_close(got_c.cpu().numpy(), orc.stage_conv(ref_h, wc.reshape(Cout, F), bc, pool), TOL, "tail: conv output vs oracle")
    if with_head:
        P = H * W
        conv_w = torch.cat([hw[None], T(rs.normal(0, 0.25, (4, 16, 16)).astype(np.float32), dev)])
        ln_w = T(rs.uniform(0.5, 1.5, (5, 16, H, W)).astype(np.float32), dev)
        ln_b = T(rs.normal(0, 0.1, (5, 16, H, W)).astype(np.float32), dev)
        cw, cb_, rw, rb = (T(rs.normal(0, 0.3, 16).astype(np.float32), dev), T(np.zeros(1, np.float32), dev),
                           T(rs.normal(0, 0.3, 16).astype(np.float32), dev), T(np.full(1, 0.1, np.float32), dev))
        four = ops.head(got_c, conv_w, ln_w, ln_b, cw, cb_, rw, rb, 0.5, want_raw=True)
        three = ops.head(got_c, conv_w, ln_w, ln_b, cw, cb_, rw, rb, 0.5, want_raw=True, partial0=part0)
        for a_, b_, what in ((three[1], four[1], "cls"), (three[2], four[2], "pre-mask reg")):
            assert_close(a_.cpu().numpy(), b_.cpu().numpy(), 1e-5, f"three-pass head vs four-pass head, {what}")


@pytest.mark.parametrize("H,W,B", [(64, 64, 1), (52, 120, 2), (37, 41, 3), (128, 128, 1)])
def test_cooperative_head_equals_four_passes(dev, H, W, B):
    """urnn_head_coop_f32 (flood_head.py:131-177): the head of a small plane as ONE launch -- its four passes with three grid barriers,
    every thread keeping its pixels' branch activations in registers -- against the four-launch head: bit-identical outputs, launch
    after launch (the barrier words are reused), including an odd plane (4-byte accesses) and several samples."""
    from urnn_amd import ops
    from urnn_amd._lib import lib
    rs = np.random.RandomState(900 + H + W)
    assert 0 < lib().urnn_head_coop_blocks_f32(B, H, W) <= 128
    feat = T(rs.normal(0, 1, (B, 16, H, W)).astype(np.float32), dev)
    conv_w = T(rs.normal(0, 0.25, (5, 16, 16)).astype(np.float32), dev)
    ln_w = T(rs.uniform(0.5, 1.5, (5, 16, H, W)).astype(np.float32), dev)
    ln_b = T(rs.normal(0, 0.1, (5, 16, H, W)).astype(np.float32), dev)
    cw, cb_, rw, rb = (T(rs.normal(0, 0.3, 16).astype(np.float32), dev), T(np.zeros(1, np.float32), dev),
                       T(rs.normal(0, 0.3, 16).astype(np.float32), dev), T(np.full(1, 0.1, np.float32), dev))
    ws = ops.workspace(ops.head_workspace_bytes(B, 16, H, W), dev)
    four = ops.head(feat, conv_w, ln_w, ln_b, cw, cb_, rw, rb, 0.5, want_raw=True, ws=ws)
    for _ in range(20):
        one = ops.head(feat, conv_w, ln_w, ln_b, cw, cb_, rw, rb, 0.5, want_raw=True, ws=ws, coop=True)
        for a_, b_, what in zip(one, four, ("masked", "cls", "pre-mask reg")):
            assert torch.equal(a_, b_), f"cooperative head differs in {what}: max {float((a_ - b_).abs().max()):.3e}"
    assert ops.workspace_status(ws) == 0
    assert lib().urnn_head_coop_blocks_f32(1, 500, 500) == 0         # 489 blocks > the device's CUs: the 500x500 head stays on its four launches


@pytest.mark.parametrize("H,W,B", [(64, 64, 1), (100, 60, 2)])
def test_frame_loop_forms_store_the_next_frame_index(dev, H, W, B):
    """urnn_head_rollout_f32 / urnn_preprocess_rollout_f32 / urnn_stage1_scalar_rain_rollout_f32 (one iteration of test.py:326-377 in a
    captured loop, frame index on the device): same outputs as the plain entries for the frame *frame_index / *t_dev names, that
    value + 1 stored to the second word (two words used alternately are a frame counter), the launch's own word refused."""
    import urnn_amd.weights as uw
    from urnn_amd import ops
    from urnn_amd._lib import UrnnError, lib
    from urnn_amd.dataset import event_to_device
    rs = np.random.RandomState(77 + H)
    nums, Tn = 4, 6
    ev = event_to_device(uw.make_event(Tn, H, W, 6.0, seed=3, batch=B), dev)
    pair = torch.tensor([2, 40], dtype=torch.int32, device=dev)
    own, nxt = pair[0:1], pair[1:2]
    args = (ev["rain"], ev["cumsum"], ev["dem"], ev["imperv"], ev["manhole"], ev["dem_min"], ev["dem_max"], 0, nums, 6.0, 250.0)
    plain = ops.preprocess(*args, t_dev=own)
    both = ops.preprocess(*args, t_dev=own, t_next=nxt)
    assert torch.equal(plain, both) and pair.tolist() == [2, 3]
    again = ops.preprocess(*args, t_dev=nxt, t_next=own)          # the next frame: the words swap roles
    assert pair.tolist() == [4, 3] and torch.equal(again, ops.preprocess(*args[:7], 3, *args[8:]))
    pair.copy_(torch.tensor([2, 40], dtype=torch.int32))
    # scalar-rain stage 1
    Cout = 32
    w1 = T(rs.normal(0, 0.3, (Cout, 2 * nums + 3)).astype(np.float32), dev)
    b1 = T(rs.normal(0, 0.1, Cout).astype(np.float32), dev)
    S = ops.stage1_static(ev["dem"], ev["imperv"], ev["manhole"], ev["dem_min"], ev["dem_max"], w1, nums)
    plain = ops.stage1_scalar_rain(S, ev["rain"], ev["cumsum"], w1, b1, 0, nums, 6.0, 250.0, t_dev=own)
    both = ops.stage1_scalar_rain(S, ev["rain"], ev["cumsum"], w1, b1, 0, nums, 6.0, 250.0, t_dev=own, t_next=nxt)
    assert torch.equal(plain, both) and pair.tolist() == [2, 3]
    with pytest.raises(UrnnError):
        ops.stage1_scalar_rain(S, ev["rain"], ev["cumsum"], w1, b1, 0, nums, 6.0, 250.0, t_dev=own, t_next=own)
    # head: four launches, one cooperative launch
    feat = T(rs.normal(0, 1, (B, 16, H, W)).astype(np.float32), dev)
    conv_w = T(rs.normal(0, 0.25, (5, 16, 16)).astype(np.float32), dev)
    ln_w = T(rs.uniform(0.5, 1.5, (5, 16, H, W)).astype(np.float32), dev)
    ln_b = T(rs.normal(0, 0.1, (5, 16, H, W)).astype(np.float32), dev)
    cw, cb_, rw, rb = (T(rs.normal(0, 0.3, 16).astype(np.float32), dev), T(np.zeros(1, np.float32), dev),
                       T(rs.normal(0, 0.3, 16).astype(np.float32), dev), T(np.full(1, 0.1, np.float32), dev))
    ws = ops.workspace(ops.head_workspace_bytes(B, 16, H, W), dev)
    four = ops.head(feat, conv_w, ln_w, ln_b, cw, cb_, rw, rb, 0.5, want_raw=True, ws=ws)
    for coop in (False, True):
        if coop and not 0 < lib().urnn_head_coop_blocks_f32(B, H, W) <= 128:
            continue
        bufs = [torch.zeros((Tn, B, H, W), device=dev) for _ in range(3)]
        pair.copy_(torch.tensor([2, 40], dtype=torch.int32))
        ops.head(feat, conv_w, ln_w, ln_b, cw, cb_, rw, rb, 0.5, out_masked=bufs[0], out_cls=bufs[1], out_raw=bufs[2], frame_index=own,
                 ws=ws, coop=coop, frame_next=nxt)
        assert pair.tolist() == [2, 3]
        for buf, ref, what in zip(bufs, four, ("masked", "cls", "pre-mask reg")):
            assert torch.equal(buf[2], ref), f"frame-loop head (coop={coop}) differs in {what}"
            assert float(buf[:2].abs().max()) == 0 and float(buf[3:].abs().max()) == 0
    with pytest.raises(UrnnError):
        ops.head(feat, conv_w, ln_w, ln_b, cw, cb_, rw, rb, 0.5, out_masked=bufs[0], out_cls=bufs[1], out_raw=bufs[2], frame_index=own,
                 ws=ws, frame_next=own)
    assert ops.workspace_status(ws) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,B", [(256, 512, 1), (200, 336, 2)])
def test_last_conv_takes_the_heads_first_statistics(dev, H, W, B):
    """urnn_stage_conv_stem_f32 (decoder.py:150-164 + flood_head.py:131-140): the decoder's 64 -> 16 conv with the head's first LayerNorm
    partials taken in its epilogue -- the feature map bit-identical to urnn_stage_conv_f32, the head run from those partials
    (three passes over the plane instead of four) within 1e-5 of the four-pass head (the sums are grouped differently: 128-pixel
    tiles of the conv against 256-pixel blocks of head_k1) and both within 1e-4 of the oracle."""
    from oracle import oracle as orc
    from urnn_amd import ops
    rs = np.random.RandomState(77 + H)
    assert ops.stage_conv_stem_applies(B, 64, 16, H, W) and not ops.stage_conv_stem_applies(1, 64, 16, 64, 64)
    x = rs.normal(0, 1, (B, 64, H, W)).astype(np.float32)
    wc = rs.normal(0, 1 / 8.0, (16, 64, 1, 1)).astype(np.float32)
    bc = rs.normal(0, 0.1, 16).astype(np.float32)
    cpk = ops.pack_conv(T(wc, dev), T(bc, dev))
    conv_w = T(rs.normal(0, 0.25, (5, 16, 16)).astype(np.float32), dev)
    ln_w = T(rs.uniform(0.5, 1.5, (5, 16, H, W)).astype(np.float32), dev)
    ln_b = T(rs.normal(0, 0.1, (5, 16, H, W)).astype(np.float32), dev)
    cw, cb_, rw, rb = (T(rs.normal(0, 0.3, 16).astype(np.float32), dev), T(np.zeros(1, np.float32), dev),
                       T(rs.normal(0, 0.3, 16).astype(np.float32), dev), T(np.full(1, 0.1, np.float32), dev))
    want = ops.stage_conv(T(x, dev), cpk, 16, False)
    part0 = ops.head_tail_partial(B, H, W, dev)
    got = ops.stage_conv(T(x, dev), cpk, 16, False, head_w=conv_w, head_partial0=part0)
    assert torch.equal(got, want), f"feature map differs: max {float((got - want).abs().max()):.3e}"
    four = ops.head(want, conv_w, ln_w, ln_b, cw, cb_, rw, rb, 0.5, want_raw=True)
    three = ops.head(got, conv_w, ln_w, ln_b, cw, cb_, rw, rb, 0.5, want_raw=True, partial0=part0)
    for a_, b_, what in ((three[1], four[1], "cls"), (three[2], four[2], "pre-mask reg")):
        assert_close(a_.cpu().numpy(), b_.cpu().numpy(), 1e-5, f"three-pass head vs four-pass head, {what}")
    ref_f = orc.stage_conv(x, wc.reshape(16, 64), bc, False)
    assert_close(got.cpu().numpy(), ref_f, TOL, "feature map vs oracle")


def _poisoned(nbytes, dev):
    """A workspace whose scratch is all-ones bits (NaN as float32) behind the zeroed status area."""
    from urnn_amd import ops
    ws = torch.full((int(nbytes),), 0xFF, dtype=torch.uint8, device=dev)
    ws[:min(ops.STATUS_AREA_BYTES, int(nbytes))].zero_()
    return ws


@pytest.mark.parametrize("I,F,skip,H,W,B,flags", [
    (16, 64, 0, 64, 64, 1, 0),          # small plane, three kernels
    (16, 64, 0, 64, 64, 1, 64),         # ... one cooperative launch
    (96, 96, 1, 37, 41, 2, 0),          # ragged plane, two samples
    (64, 96, 0, 250, 250, 1, 0),        # half resolution, three kernels (grouped gate GEMM, two-stream candidate)
    (64, 96, 0, 250, 250, 1, 64),       # ... the four-tiles-per-block cooperative launch
    (16, 64, 0, 500, 500, 1, 32),       # full resolution with the reset gate recomputed (URNN_PHASE_FUSED_R)
])
def test_cell_and_head_do_not_read_uninitialised_scratch(dev, I, F, skip, H, W, B, flags):
    """ops.workspace() hands out torch.empty memory with only the status area zeroed: every kernel must write its scratch
    before it reads it.  A cell and the head run on a workspace POISONED with NaN bit patterns must give the bits of the zero-filled run
    (a masked read of the form 0 * garbage would turn into NaN)."""
    from urnn_amd import ops
    rs = np.random.RandomState(5 + I + F + H)
    K = I + (2 * F if skip else F)
    packed = ops.pack_gru(T(rs.normal(0, 1 / np.sqrt(K), (2 * F, K, 1, 1)).astype(np.float32), dev), T(rs.normal(0, 0.1, 2 * F).astype(np.float32), dev),
                          T(rs.normal(0, 1 / np.sqrt(K), (F, K, 1, 1)).astype(np.float32), dev), T(rs.normal(0, 0.1, F).astype(np.float32), dev), I, F, bool(skip))
    aff = tuple(T(a.astype(np.float32), dev) for a in (rs.uniform(0.5, 1.5, 2 * F), rs.normal(0, 0.1, 2 * F), rs.uniform(0.5, 1.5, F), rs.normal(0, 0.1, F)))
    x = T(rs.normal(0, 1, (B, I, H, W)).astype(np.float32), dev)
    e = T(rs.normal(0, 1, (B, F, H, W)).astype(np.float32), dev) if skip else None
    h = T(rs.normal(0, 1, (B, F, H, W)).astype(np.float32), dev)
    nbytes = ops.gru_cell_workspace_bytes(B, F, H, W)
    clean = ops.gru_cell(x, e, h, packed, *aff, I, phases=ops.PHASE_ALL | flags, ws=torch.zeros(nbytes, dtype=torch.uint8, device=dev))
    ws = _poisoned(nbytes, dev)
    dirty = ops.gru_cell(x, e, h, packed, *aff, I, phases=ops.PHASE_ALL | flags, ws=ws)
    torch.cuda.synchronize()
    assert ops.workspace_status(ws) == 0
    assert bool(torch.isfinite(dirty).all()) and torch.equal(clean, dirty)
    # the head on the same plane
    C = 16
    feat = T(rs.normal(0, 1, (B, C, H, W)).astype(np.float32), dev)
    conv_w = T(rs.normal(0, 0.25, (5, C, C)).astype(np.float32), dev)
    ln_w, ln_b = T(rs.uniform(0.5, 1.5, (5, C, H * W)).astype(np.float32), dev), T(rs.normal(0, 0.1, (5, C, H * W)).astype(np.float32), dev)
    cls_w, reg_w = T(rs.normal(0, 0.25, C).astype(np.float32), dev), T(rs.normal(0, 0.25, C).astype(np.float32), dev)
    cls_b, reg_b = T(np.asarray([0.05], np.float32), dev), T(np.asarray([-0.02], np.float32), dev)
    hb = ops.head_workspace_bytes(B, C, H, W)
    for coop in ((False, True) if lib_head_coop(B, H, W) else (False,)):
        a = ops.head(feat, conv_w, ln_w, ln_b, cls_w, cls_b, reg_w, reg_b, 0.5, want_raw=True, ws=torch.zeros(hb, dtype=torch.uint8, device=dev), coop=coop)
        wsh = _poisoned(hb, dev)
        b = ops.head(feat, conv_w, ln_w, ln_b, cls_w, cls_b, reg_w, reg_b, 0.5, want_raw=True, ws=wsh, coop=coop)
        torch.cuda.synchronize()
        assert ops.workspace_status(wsh) == 0
        for u, v in zip(a, b):
            assert bool(torch.isfinite(v).all()) and torch.equal(u, v), f"head (coop={coop}) read uninitialised scratch"


def lib_head_coop(B, H, W):
    from urnn_amd._lib import lib
    return lib().urnn_head_coop_blocks_f32(B, H, W) > 0
