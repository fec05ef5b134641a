"""SURVEY 8(f) N1 / N3 on the host: the `.npy` event loader and the evaluation metrics against goldens produced by running
the reference's Dynamic2DFlood / compute_metrics over the same seeded files (tests/golden/make_golden.py)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synth_dataset as sd  # noqa: E402

from urnn_amd.events import Dynamic2DFlood  # noqa: E402
from urnn_amd.metrics import compute_metrics, summarize  # noqa: E402


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "events_metrics.npz"))


@pytest.fixture(scope="module")
def tree(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("urbanflood"))
    return root, sd.write_tree(root)


def test_loader_matches_reference_items(gold, tree):
    root, lst = tree
    ds = Dynamic2DFlood(root, "test", event_list_file=lst, duration=sd.DURATION)
    assert len(ds) == int(gold["len"]) == 6
    assert ds.locations == list(gold["locations"]) == ["location2", "location16", "region_b"]      # numeric, then by name
    assert ds.event_names == list(gold["event_names"])                                              # blank lines skipped
    for i in range(len(ds)):
        inp, tgt, event_dir = ds[i]
        assert os.path.relpath(event_dir, root) == str(gold[f"item{i}_dir"])
        assert tgt.dtype == torch.float32 and np.array_equal(tgt.numpy(), gold[f"item{i}_target"])
        assert set(inp) == {"absolute_DEM", "max_DEM", "min_DEM", "impervious", "manhole", "rainfall", "cumsum_rainfall"}
        for k, v in inp.items():
            ref = gold[f"item{i}_{k}"]
            assert tuple(v.shape) == ref.shape, (i, k)
            assert np.array_equal(v.numpy(), ref), (i, k)
    # spatial rain on the third location, scalar elsewhere; padded to the duration
    assert ds[2][0]["rainfall"].shape == (sd.DURATION, 1, sd.H, sd.W) and ds[0][0]["rainfall"].shape == (sd.DURATION, 1, 1, 1)
    assert float(ds[0][0]["rainfall"][sd.T_FILE:].abs().sum()) == 0.0


def test_loader_location_filter_default_listing_and_errors(gold, tree):
    root, lst = tree
    one = Dynamic2DFlood(root, "test", event_list_file=lst, duration=sd.DURATION, location="location16")
    assert len(one) == int(gold["one_len"]) == 2
    with pytest.raises(ValueError, match="not found"):
        Dynamic2DFlood(root, "test", event_list_file=lst, location="nowhere")
    auto = Dynamic2DFlood(root, "test", duration=sd.DURATION)          # no list file: sorted event directories
    assert auto.event_names == sorted(sd.EVENTS)
    with pytest.raises(IndexError):
        auto[len(auto)]
    inp, tgt, _ = auto.batched(0)
    assert inp["rainfall"].shape == (1, sd.DURATION, 1, 1, 1) and inp["absolute_DEM"].shape == (1, 1, 1, sd.H, sd.W)
    assert tgt.shape == (1, sd.DURATION, sd.H, sd.W)
    assert list(auto.shard(1, 4)) == [1, 5] and list(auto.shard(0, 4)) == [0, 4]


def test_metrics_match_reference(gold):
    pred, gt = gold["m_pred"], gold["m_gt"]
    for thr in (150.0, 600.0):
        m = compute_metrics(pred, gt, flood_thres=thr)
        for k, v in m.items():
            ref = float(gold[f"m_{int(thr)}_{k}"])
            assert v == pytest.approx(ref, rel=2e-5, abs=1e-9), (thr, k)      # the reference sums in float32
    m = compute_metrics(torch.from_numpy(pred), torch.from_numpy(gt))
    assert m["CSI"] == pytest.approx(float(gold["m_150_CSI"]), rel=1e-9)
    perfect = compute_metrics(gt, gt)
    assert perfect["R2"] == pytest.approx(1.0) and perfect["MAE"] == 0.0 and perfect["CSI"] == pytest.approx(1.0)
    with pytest.raises(ValueError):
        compute_metrics(pred[0], gt[0])
    s = summarize({"a": m, "b": perfect})
    assert s["mean"]["MAE"] == pytest.approx(m["MAE"] / 2) and s["std"]["MAE"] == pytest.approx(m["MAE"] / 2)
