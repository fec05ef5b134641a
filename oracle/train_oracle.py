"""CPU restatement of the training loss (test infrastructure, like the rest of oracle/): `FocalBCE_and_WMSE` of the reference
(losses.py:44-249) as the SWP loop applies it (main.py:489-539: pred["reg"] = network output, pred["cls"] = [reg >= cls_thred]
through a non-differentiable `torch.where`), with the closed-form gradient w.r.t. the regression output.  Pinned against
reference-autograd goldens (tests/golden/train_window_16x16.npz, tests/test_train_goldens.py).  numpy, float64."""
import numpy as np

WET_WEIGHT = 20.0      # WMSELoss.factor (losses.py:147)
ALPHA, GAMMA = 0.25, 2.0
LOG_EPS = 1e-9         # FocalBCELoss.forward(inf=1e-9) (losses.py:222)
CLS_WEIGHT = 0.1       # FocalBCE_and_WMSE.forward (losses.py:100)


def wmse(reg, tgt):
    """(loss_reg, wet-cell MSE, dry-cell MSE): 20 * mean((reg - tgt)^2 | tgt > 0) + mean((reg - tgt)^2 | tgt <= 0)
    (losses.py:150-189)."""
    reg, tgt = np.asarray(reg, np.float64), np.asarray(tgt, np.float64)
    wet = tgt > 0
    err2 = (reg - tgt) ** 2
    flood = err2[wet].mean() if wet.any() else np.nan
    dry = err2[~wet].mean() if (~wet).any() else np.nan
    return WET_WEIGHT * flood + dry, flood, dry


def focal_bce(p, y):
    """Focal BCE with mean reduction on probabilities p and labels y (losses.py:204-249)."""
    p, y = np.asarray(p, np.float64), np.asarray(y, np.float64)
    loss = (-ALPHA * (1 - p) ** GAMMA * y * np.log(np.abs(p) + LOG_EPS)
            - (1 - ALPHA) * p ** GAMMA * (1 - y) * np.log(np.abs(1 - p) + LOG_EPS))
    return loss.mean()


def loss_and_grad(reg, tgt, cls_thred=0.0):
    """The training loss on a window's concatenated outputs and its gradient w.r.t. `reg`.
    Returns (dict of the five reference components, dL/dreg).  The classification term sees only the thresholded output, so
    it contributes to the value but not to the gradient."""
    reg64, tgt64 = np.asarray(reg, np.float64), np.asarray(tgt, np.float64)
    loss_reg, flood, dry = wmse(reg64, tgt64)
    cls = (reg64 >= cls_thred).astype(np.float64)
    loss_cls = focal_bce(cls, (tgt64 > 0).astype(np.float64))
    wet = tgt64 > 0
    n_wet, n_dry = int(wet.sum()), int((~wet).sum())
    g = np.zeros_like(reg64)
    if n_wet:
        g[wet] = WET_WEIGHT * 2.0 * (reg64 - tgt64)[wet] / n_wet
    if n_dry:
        g[~wet] = 2.0 * (reg64 - tgt64)[~wet] / n_dry
    comps = {"loss": loss_reg + CLS_WEIGHT * loss_cls, "loss_reg": loss_reg, "loss_reg_label": flood, "loss_reg_pred": dry,
             "loss_cls": loss_cls}
    return comps, g
