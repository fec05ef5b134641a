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


# ---- ConvGRU / Skip-ConvGRU cell: forward + backward (ConvRNN.py:111-194 with 1x1 gates), float64 -------------------------
def _gn_forward(v, gamma, beta, eps):
    """GroupNorm with 32-channel groups over (32, P) per sample.  v (B,C,P) -> y, xhat, rstd (B,G,1,1)."""
    B, C, P = v.shape
    G = C // 32
    vg = v.reshape(B, G, 32 * P)
    mu = vg.mean(axis=2, keepdims=True)
    var = vg.var(axis=2, keepdims=True)
    rstd = 1.0 / np.sqrt(var + eps)
    xhat = ((vg - mu) * rstd).reshape(B, C, P)
    return xhat * gamma[None, :, None] + beta[None, :, None], xhat, rstd


def _gn_backward(dy, xhat, rstd, gamma):
    """Returns (dv, dgamma, dbeta)."""
    B, C, P = dy.shape
    G = C // 32
    dxh = dy * gamma[None, :, None]
    dg, xg = dxh.reshape(B, G, 32 * P), xhat.reshape(B, G, 32 * P)
    dv = rstd * (dg - dg.mean(axis=2, keepdims=True) - xg * (dg * xg).mean(axis=2, keepdims=True))
    return dv.reshape(B, C, P), (dy * xhat).sum(axis=(0, 2)), dy.sum(axis=(0, 2))


def gru_cell_backward(x, e, h, p, dout, eps=1e-5):
    """Gradients of sum(h' * dout) for one cell step.  x may be None (zeros, I channels; no dx returned), e None for the
    encoder cell.  `p` holds W1 (2F,K[,1,1]), b1, g1, be1, W2 (F,K[,1,1]), b2, g2, be2 with K ordered x | e | h.
    Returns (h', grads) with grads keys dx, de, dh, dW1, db1, dg1, dbe1, dW2, db2, dg2, dbe2 (shapes of the parameters)."""
    f64 = lambda a: np.asarray(a, np.float64)
    h = f64(h)
    B, F, H, W = h.shape
    P = H * W
    W1 = f64(p["W1"]).reshape(2 * F, -1)
    W2 = f64(p["W2"]).reshape(F, -1)
    K = W1.shape[1]
    I = K - (2 * F if e is not None else F)
    xs = f64(x).reshape(B, I, P) if x is not None else np.zeros((B, I, P))
    parts = [xs] + ([f64(e).reshape(B, F, P)] if e is not None else [])
    hp = h.reshape(B, F, P)
    do = f64(dout).reshape(B, F, P)
    A = np.concatenate(parts + [hp], axis=1)
    g_raw = np.einsum("nk,bkp->bnp", W1, A) + f64(p["b1"])[None, :, None]
    y1, xh1, rstd1 = _gn_forward(g_raw, f64(p["g1"]), f64(p["be1"]), eps)
    s = 1.0 / (1.0 + np.exp(-y1))
    z, r = s[:, :F], s[:, F:]
    A2 = np.concatenate(parts + [r * hp], axis=1)
    c_raw = np.einsum("nk,bkp->bnp", W2, A2) + f64(p["b2"])[None, :, None]
    y2, xh2, rstd2 = _gn_forward(c_raw, f64(p["g2"]), f64(p["be2"]), eps)
    n = np.tanh(y2)
    hnew = (1 - z) * hp + z * n
    # backward
    dz, dn, dh = do * (n - hp), do * z, do * (1 - z)
    dc, dg2, dbe2 = _gn_backward(dn * (1 - n * n), xh2, rstd2, f64(p["g2"]))
    dW2 = np.einsum("bnp,bkp->nk", dc, A2)
    dA2 = np.einsum("nk,bnp->bkp", W2, dc)
    drh = dA2[:, K - F:]
    dr = drh * hp
    dh = dh + drh * r
    dy1 = np.concatenate([dz * z * (1 - z), dr * r * (1 - r)], axis=1)
    dgr, dg1, dbe1 = _gn_backward(dy1, xh1, rstd1, f64(p["g1"]))
    dW1 = np.einsum("bnp,bkp->nk", dgr, A)
    dA = np.einsum("nk,bnp->bkp", W1, dgr)
    dA[:, :K - F] += dA2[:, :K - F]
    dh = dh + dA[:, K - F:]
    grads = {"dh": dh.reshape(B, F, H, W), "dW1": dW1.reshape(np.shape(p["W1"])), "db1": dgr.sum(axis=(0, 2)), "dg1": dg1,
             "dbe1": dbe1, "dW2": dW2.reshape(np.shape(p["W2"])), "db2": dc.sum(axis=(0, 2)), "dg2": dg2, "dbe2": dbe2}
    if x is not None:
        grads["dx"] = dA[:, :I].reshape(B, I, H, W)
    if e is not None:
        grads["de"] = dA[:, I:I + F].reshape(B, F, H, W)
    return hnew.reshape(B, F, H, W), grads
