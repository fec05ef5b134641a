"""CPU-only checks: the C-ABI library builds, loads and exports every symbol include/urnn_hip.h declares (no compute
calls without a GPU); host-side logic (architecture table, state shapes, checkpoint key handling, event sharding);
the product path fails loudly instead of falling back when there is no GPU."""
import os
import re

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="session")
def built():
    import __graft_entry__ as g
    g.build()
    from urnn_amd import _lib
    return _lib


def test_header_is_plain_c():
    """include/urnn_hip.h is the boundary a C / cgo / JNI consumer compiles against: it must be valid C99 on its own."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    header = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "urnn_hip.h")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", header], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_library_exports_every_declared_symbol(built):
    header = open(os.path.join(REPO, "include", "urnn_hip.h")).read()
    declared = set(re.findall(r"\b(urnn_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 15
    lib = built.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"liburnn_hip.so does not export {name}"
    assert declared == set(built.SIGNATURES), "ctypes signature table and header disagree"
    assert lib.urnn_abi_version() == 3


def test_packed_sizes_and_argument_errors_without_gpu(built):
    lib = built.lib()
    # pure host-side size arithmetic
    # fp32 slab [KT][NB][64] + bias row, then the same slab as bf16 pieces [ceil(KT/8)][NB][3][64][4 dwords] (gradient GEMMs) and as
    # scaled f16 pieces [ceil(KT/8)][NB][2][64][4 dwords] (forward GEMMs)
    split = lambda KT, NB: ((KT + 7) // 8) * NB * 3 * 256
    f16 = lambda KT, NB: ((KT + 7) // 8) * NB * 2 * 256
    assert lib.urnn_packed_conv_floats(63, 16) == 32 * 1 * 64 + 32 + split(32, 1) + f16(32, 1)
    # gate slabs [F/32 groups][KT][z|r][64] + b1, candidate slab(s) [KT][F/32 blocks][64] + b2, then both in split and f16 form
    # (the f16 gate slab is ONE group of all four 32-column blocks z0 r0 z1 r1 -- same size as two groups of two -- followed by
    # the gate bias in that order: + 2F)
    # F = 64 cells whose x | e part is whole 16-k groups also carry the fused-reset-gate candidate slab (URNN_PHASE_FUSED_R): nXE
    # groups of [r0 r1 c0 c1] + 4 hidden-state groups of [r0 r1], 4 groups of W2[:, h] x [c0 c1] (two f16 pieces x 256 dwords per
    # block), bias [b1 r | b2]
    fused = lambda nXE: nXE * 4 * 512 + 4 * 2 * 512 + 4 * 2 * 512 + 128
    assert lib.urnn_packed_gru_floats(16, 64, 0) == 2 * (40 * 2 * 64) + 128 + 40 * 2 * 64 + 64 + 3 * split(40, 2) + f16(40, 4) + f16(40, 2) + 128 + fused(1)
    assert lib.urnn_packed_gru_floats(96, 64, 1) == 2 * (112 * 2 * 64) + 128 + 112 * 2 * 64 + 64 + 3 * split(112, 2) + f16(112, 4) + f16(112, 2) + 128 + fused(10)
    assert lib.urnn_packed_gru_floats(64, 96, 0) == 3 * (80 * 2 * 64) + 192 + 80 * 3 * 64 + 96 + 3 * split(80, 2) + split(80, 3) + 2 * f16(80, 3) + f16(80, 3) + 192
    assert lib.urnn_gru_cell_workspace_bytes(1, 64, 500, 500) > 3 * 64 * 250000 * 4
    # argument validation happens before any HIP call
    assert lib.urnn_stage_conv_f32(0, 0, 0, 1, 8, 16, 4, 4, 0, 0.2, 0) == -2      # URNN_ENULL
    assert b"NULL" in lib.urnn_last_error()
    assert lib.urnn_gru_cell_f32(0, 0, 16, 16, 16, 16, 16, 16, 16, 16, 1 << 30, 1, 16, 48, 4, 4, 1e-5, 0) == -1  # F % 32
    assert b"multiple of 32" in lib.urnn_last_error()
    # planes whose per-sample segments pass 4 GiB (32-bit DMA offsets) are refused, not mis-addressed
    assert lib.urnn_gru_cell_f32(0, 0, 16, 16, 16, 16, 16, 16, 16, 16, 1 << 30, 1, 16, 128, 4096, 4096, 1e-5, 0) == -1
    assert b"4-GiB" in lib.urnn_last_error()
    assert lib.urnn_stage_conv_f32(16, 16, 16, 1, 64, 64, 5000, 5000, 0, 0.2, 0) == -1
    assert lib.urnn_deconv2x2_f32(16, 16, 16, 1, 96, 96, 2000, 2000, 0.2, 0) == -1


def test_no_cpu_fallback():
    from urnn_amd import ops
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.stage_conv(torch.zeros(1, 8, 4, 4), torch.zeros(8), 16, False)


def test_architecture_table_and_state_shapes():
    from urnn_amd.net_config import get_input_channels, get_state_shapes, load_net_config
    cfg = load_net_config()
    assert get_input_channels(cfg, 30) == 63 and get_input_channels(cfg, 3) == 9
    assert get_state_shapes(cfg, 500, 500) == [(1, 64, 500, 500), (1, 96, 250, 250), (1, 96, 125, 125),
                                               (1, 96, 125, 125), (1, 96, 250, 250), (1, 64, 500, 500)]
    assert get_state_shapes(cfg, 64, 64, batch=8)[2] == (8, 96, 16, 16)


def test_parameter_counts_match_reference():
    import urnn_amd.weights as uw
    assert sum(v.size for v in uw.make_state_dict(64, 64, 9).values()) == 1_075_506
    assert len(uw.make_state_dict(16, 16, 9)) == 79


def test_model_mirror_has_reference_keys_and_loads_alias_checkpoints():
    import urnn_amd.weights as uw
    from urnn_amd.net_config import load_net_config
    from urnn_amd.networks import ED, get_network_params
    ep, dp = get_network_params(False, 16, 16, 9, load_net_config())
    net = ED(False, ep, dp, 0.5, False, 16, 16)
    sd = uw.make_state_dict(16, 16, 9, seed=1)
    assert set(net.state_dict()) == set(sd)
    # a reference checkpoint carries alias keys for every checkpoint wrapper and (under DDP) a "module." prefix
    ckpt = {}
    for k, v in sd.items():
        t = torch.from_numpy(v)
        ckpt["module." + k] = t
        head, _, tail = k.partition(".")
        parts = k.split(".")
        ckpt["module." + parts[0] + "." + parts[1] + "_wrapper.module." + ".".join(parts[2:])] = t
        if ".conv1." in k or ".conv2." in k:
            ckpt["module." + k.replace(".conv1.", ".conv1_module_wrapper.module.").replace(".conv2.", ".conv2_module_wrapper.module.")] = t
    assert len(ckpt) > 2 * len(sd)
    res = net.load_state_dict(ckpt)
    assert not res.missing_keys and not res.unexpected_keys
    for k, v in net.state_dict().items():
        assert np.array_equal(v.numpy(), sd[k])


def test_unsupported_architectures_fail_loudly():
    from urnn_amd.networks import CGRU_cell
    from urnn_amd.networks.utils import make_layers
    with pytest.raises(NotImplementedError):
        CGRU_cell(False, (8, 8), 16, 3, 64, "encoder")            # only 1x1 gate convs exist in the published net
    with pytest.raises(NotImplementedError):
        make_layers({"conv1_leaky_1": [9, 16, 3, 1, 1]})


def test_event_flattening_matches_reference_layout():
    import urnn_amd.weights as uw
    from urnn_amd.dataset import event_to_device
    ev = uw.make_event(7, 8, 12, 6.0, seed=0, batch=2)
    flat = event_to_device(ev, torch.device("cpu"))
    assert flat["rain"].shape == (2, 7) and flat["dem"].shape == (2, 8, 12) and flat["T"] == 7
    sp = event_to_device(uw.make_event(7, 8, 12, 6.0, seed=0, spatial_rain=True), torch.device("cpu"))
    assert sp["rain"].shape == (1, 7, 8, 12)
    assert np.allclose(np.cumsum(ev["rainfall"], axis=1), ev["cumsum_rainfall"], rtol=1e-6)


def test_shard_events_is_distributed_sampler():
    from torch.utils.data.distributed import DistributedSampler
    from urnn_amd.distributed import shard_events
    for n, world in ((10, 4), (3, 8), (8, 8), (17, 2)):
        for rank in range(world):
            ref = list(DistributedSampler(range(n), num_replicas=world, rank=rank, shuffle=False))
            assert shard_events(n, rank, world) == ref


def test_swp_window_schedule():
    """split_iter_index of the reference (main.py:162-178): consecutive windows of seq_num steps; a remainder moves the last
    window back so that it ends with the training window."""
    from urnn_amd.training import window_starts
    assert window_starts(0, 12, 36) == [0, 12, 24]
    assert window_starts(5, 4, 12) == [5, 9, 13]
    assert window_starts(0, 4, 10) == [0, 4, 6]
    assert window_starts(0, 8, 8) == [0]


def test_exp_config_keys_match_the_benchmarked_workloads(tmp_path):
    """The experiment YAMLs of the reference (configs/*.yaml, keys of config.py:55-213) parse to the workload tuples bench.py
    ships for the BASELINE configs -- values restated here from location1_scratch.yaml:41-58, lite.yaml:31-58,
    futian_scratch.yaml:41-68, ukea_scratch.yaml:43-70 (the reference tree does not travel with the tests)."""
    import bench
    from urnn_amd.exp_config import load_exp_config, workload
    shipped = {
        "location1": dict(input_height=500, input_width=500, historical_nums=30, rain_max=6.0, cumsum_rain_max=250.0, duration=360,
                          flood_max=5000, seq_num=12, window_size=36),
        "lite128": dict(input_height=128, input_width=128, historical_nums=3, rain_max=60.0, cumsum_rain_max=250.0, duration=36),
        "futian": dict(input_height=400, input_width=560, historical_nums=6, rain_max=5.0, cumsum_rain_max=100.0, duration=72),
        "ukea": dict(input_height=52, input_width=120, historical_nums=6, rain_max=10.0, cumsum_rain_max=150.0, duration=36),
    }
    import yaml
    for name, keys in shipped.items():
        path = tmp_path / f"{name}.yaml"
        path.write_text(yaml.safe_dump({**keys, "lr": 0.01, "loss_name": "FocalBCE_and_WMSE"}))
        cfg = load_exp_config(str(path))
        assert workload(cfg, bench.CONFIGS[name][6]) == bench.CONFIGS[name], name
        assert cfg["lr"] == 0.01 and isinstance(cfg["seq_num"], int)          # unknown keys kept, defaults typed
    assert load_exp_config(str(tmp_path / "location1.yaml"), duration=None, test_list_file="x.txt")["test_list_file"] == "x.txt"


def test_exp_config_location_key(tmp_path):
    """`location` (config.py:121, test.py:738; set in lite.yaml / location2_scratch.yaml): the YAML's catchment reaches the loader,
    the CLI flag overrides it, the default is "" = every location under data_root."""
    import yaml
    from urnn_amd.exp_config import load_exp_config
    path = tmp_path / "loc.yaml"
    path.write_text(yaml.safe_dump({"location": "location2", "input_height": 64}))
    assert load_exp_config(str(path))["location"] == "location2"
    assert load_exp_config(str(path), location="location16")["location"] == "location16"
    assert load_exp_config(str(path), location=None)["location"] == "location2"
    (tmp_path / "none.yaml").write_text("input_height: 64\n")
    assert load_exp_config(str(tmp_path / "none.yaml"))["location"] == ""
    import inspect
    import urnn_amd.evaluate as ev
    src = inspect.getsource(ev.main)
    assert "--location" in src and 'location=cfg["location"]' in src


def test_bench_cell_kernel_work_matches_the_stage_accounting():
    """bench.py's per-launch algorithmic bytes (SURVEY 8d accounting) at 500x500: gates K + F (fused) / K + 2F planes, candidate K + F
    (fused) / K + 2F (three-pass), blend 4F -- the figures DESIGN.md section 6 and the VERDICT's table quote (144 / 288 MB ...)."""
    import bench
    w = bench.cell_kernel_work(500, 500, 1, {"enc1": True, "dec1": True, "enc2": False, "dec2": False})
    MB = 1e6
    assert w["gates"]["enc1"][0] == 144 * MB and w["gates"]["dec1"][0] == 288 * MB
    assert w["gates"]["enc2"][0] == 4.0 * 62500 * (160 + 192) and w["gates"]["dec2"][0] == 4.0 * 62500 * (288 + 192)
    assert w["candidate"]["enc1"][0] == 144 * MB and w["candidate"]["dec1"][0] == 288 * MB
    assert w["candidate"]["dec2"][0] == 4.0 * 62500 * (288 + 192)
    assert w["blend"]["dec1"][0] == 256 * MB and w["blend"]["enc2"][0] == 4.0 * 62500 * 384
    assert w["gates"]["dec1"][1] == 2.0 * 250000 * 128 * 224


def test_python_host_reads_tuning_knobs_only_under_the_switch(monkeypatch):
    """The product path takes every default whatever URNN_TUNE_* the environment holds; URNN_TUNING=1 turns the development knobs on
    (rollout.py / training.py used to read them unconditionally)."""
    from urnn_amd.ops import tuning_env
    monkeypatch.delenv("URNN_TUNING", raising=False)
    monkeypatch.setenv("URNN_TUNE_GROUP", "2")
    assert tuning_env("URNN_TUNE_GROUP", 4) == 4
    monkeypatch.setenv("URNN_TUNING", "1")
    assert tuning_env("URNN_TUNE_GROUP", 4) == "2"
    assert tuning_env("URNN_TUNE_NOT_SET", "x") == "x"
    import re
    for mod in ("rollout.py", "training.py"):
        src = open(os.path.join(REPO, "u-rnn_amd", mod)).read()
        assert not re.search(r'os\.environ\.get\("URNN_TUNE_', src), f"{mod} reads a tuning knob without the switch"


def test_bench_binds_ranks_to_disjoint_cpu_slices():
    """bench.py --bind: N ranks get N disjoint contiguous slices of the CPUs the process may use; auto = from four ranks."""
    import importlib.util
    if not hasattr(os, "sched_setaffinity"):
        pytest.skip("no sched_setaffinity on this platform")
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    before = os.sched_getaffinity(0)
    try:
        assert bench.bind_rank_to_cores(0, 2, "auto") is None and bench.bind_rank_to_cores(0, 8, "off") is None
        if len(before) >= 4:
            world = 4
            slices = []
            for r in range(world):
                os.sched_setaffinity(0, before)
                lo_hi = bench.bind_rank_to_cores(r, world, "auto")
                assert lo_hi is not None
                slices.append(os.sched_getaffinity(0))
            assert all(len(s) == len(before) // world for s in slices)
            assert all(slices[a].isdisjoint(slices[b]) for a in range(world) for b in range(a + 1, world))
    finally:
        os.sched_setaffinity(0, before)
