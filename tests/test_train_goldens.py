"""Training rows, CPU side: the reference-autograd goldens (loss, one ConvGRU cell, one SWP window with the
gradient of all 79 parameter tensors) pin the numpy oracles of oracle/train_oracle.py and the facts the HIP backward relies
on; the HIP path itself is checked against the same goldens in tests/test_hip_train.py."""
import os

import numpy as np
import pytest

from oracle import train_oracle as tro


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "train_window_16x16.npz"))


@pytest.mark.parametrize("thr", [0.0, 0.01])
def test_loss_restatement_matches_reference(gold, thr):
    comps, grad = tro.loss_and_grad(gold["loss_reg"], gold["loss_tgt"], cls_thred=thr)
    for k, v in comps.items():
        assert v == pytest.approx(float(gold[f"loss_thr{thr}_{k}"]), rel=2e-6), k
    ref = gold[f"loss_thr{thr}_dreg"]
    assert np.abs(grad - ref).max() <= 2e-6 * np.abs(ref).max()


def test_window_loss_from_reference_outputs(gold):
    comps, _ = tro.loss_and_grad(gold["win_reg"], gold["win_target"], cls_thred=0.0)
    for k in ("loss", "loss_reg", "loss_reg_label", "loss_reg_pred", "loss_cls"):
        assert comps[k] == pytest.approx(float(gold[f"win_{k}"]), rel=5e-6), k


def test_gradient_golden_covers_every_parameter_and_the_cls_branch_is_cut(gold):
    import urnn_amd.weights as uw
    names = list(uw.make_state_dict(int(gold["win_H"]), int(gold["win_W"]), 2 * int(gold["win_nums"]) + 3,
                                    seed=int(gold["win_weights_seed"])).keys())
    assert len(names) == 79
    for n in names:
        assert int(gold[f"win_hasgrad_{n}"]) == 1
        g = gold[f"win_grad_{n}"]
        assert np.isfinite(g).all()
        cls_branch = n.startswith("head.cls_")
        # the wet/dry mask is a non-differentiable comparison: nothing flows into the classification branch
        assert (np.abs(g).max() == 0.0) == cls_branch, n


@pytest.mark.parametrize("tag", ["enc", "dec", "dec0"])
def test_cell_backward_oracle_matches_reference_autograd(tag):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "train_cell_backward.npz"))
    k = lambda n: g[f"cell_{tag}_{n}"]
    p = {n: k(n) for n in ("W1", "b1", "g1", "be1", "W2", "b2", "g2", "be2")}
    x = k("x") if int(k("with_x")) else None
    e = k("e") if int(k("skip")) else None
    out, grads = tro.gru_cell_backward(x, e, k("h"), p, k("dout"))
    assert np.abs(out - k("out")).max() <= 2e-6 * np.abs(k("out")).max()
    for name, got in grads.items():
        ref = k(name)
        scale = max(np.abs(ref).max(), 1e-30)
        assert np.abs(got - ref).max() <= 2e-5 * scale, (tag, name, np.abs(got - ref).max() / scale)
    if x is None:   # x == 0: its weight columns see no gradient
        assert np.abs(grads["dW1"].reshape(grads["dW1"].shape[0], -1)[:, :int(k("I"))]).max() == 0.0
