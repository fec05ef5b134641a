"""Per-event evaluation metrics of the reference's test script (test.py:607-675), SURVEY.md 8(f) N3: R2, MSE, RMSE, MAE,
PeakR2 and CSI of a predicted flood-depth sequence against the ground truth.  Plain numpy, float64 accumulation."""
import numpy as np


def _np(a):
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    return np.asarray(a)


def _r2(pred, gt):
    res = np.sum((pred - gt) ** 2, dtype=np.float64)
    tot = np.sum((gt - gt.mean(dtype=np.float64)) ** 2, dtype=np.float64)
    return float(1.0 - res / (tot + 1e-10))


def compute_metrics(pred_mm, gt_mm, flood_thres=150.0):
    """Both inputs (T,H,W) in millimetres.  Depth metrics are reported in metres (MSE in m^2); PeakR2 is R2 at the timestep
    of the largest spatial-mean ground-truth depth; CSI compares the temporal-maximum flood extents at ``flood_thres`` mm."""
    pred_mm, gt_mm = _np(pred_mm), _np(gt_mm)
    if pred_mm.shape != gt_mm.shape or pred_mm.ndim != 3:
        raise ValueError(f"compute_metrics: expected two (T,H,W) arrays, got {pred_mm.shape} and {gt_mm.shape}")
    pred = pred_mm.astype(np.float64) / 1000.0
    gt = gt_mm.astype(np.float64) / 1000.0
    err = pred - gt
    mse = float(np.mean(err ** 2))
    t_peak = int(np.argmax(gt.mean(axis=(1, 2))))
    wet_p = pred_mm.max(axis=0) > flood_thres
    wet_g = gt_mm.max(axis=0) > flood_thres
    tp = int(np.sum(wet_p & wet_g))
    fp = int(np.sum(wet_p & ~wet_g))
    fn = int(np.sum(~wet_p & wet_g))
    return {
        "R2": _r2(pred, gt),
        "MSE": mse,
        "RMSE": float(np.sqrt(mse)),
        "MAE": float(np.mean(np.abs(err))),
        "PeakR2": _r2(pred[t_peak].ravel(), gt[t_peak].ravel()),
        "CSI": float(tp / (tp + fp + fn + 1e-10)),
    }


METRIC_NAMES = ("R2", "MSE", "RMSE", "MAE", "PeakR2", "CSI")


def summarize(all_metrics):
    """{event: metrics} -> {"mean": {...}, "std": {...}} over events (population std, as test.py:693-695)."""
    rows = list(all_metrics.values())
    return {"mean": {c: float(np.mean([r[c] for r in rows])) for c in METRIC_NAMES},
            "std": {c: float(np.std([r[c] for r in rows])) for c in METRIC_NAMES}}
