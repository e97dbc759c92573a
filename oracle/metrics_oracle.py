"""CPU restatement of the depth evaluation metrics (TEST INFRASTRUCTURE ONLY).

Restates compute_errors, /root/reference/utils_criterion.py:6-90, branch by branch, as a
pure function of counts and sums so the HIP masked-metric reduction can be checked against
the same intermediate quantities:
  :22-25  mask = gt != 0; empty -> seven zeros
  :33     eps  = 1e-3 if max(gt[mask]) > 1 else 1e-6
  :34     keep = (pred > eps) & (gt > eps)
  :36-54  fallbacks when keep is empty: gt > eps, then & (pred > 0); all-bad -> (1, gt.max, 0,0,0, 1, gt.max)
  :60-65  eps re-evaluated on the kept gt; thresh = max(gt/max(pred,eps), max(pred,eps)/gt); a1..a3
  :67-90  rmse, abs_rel, log10, mae; NaN/inf -> 0
Returns (abs_rel, rmse, a1, a2, a3, log_10, mae) like the reference.

Pinned by: tests/test_oracle_golden.py against tests/golden/metrics_*.npz.
"""
from __future__ import annotations

import numpy as np


def _clean(x):
    x = float(x)
    return 0.0 if (x != x or x == np.inf) else x


def compute_errors(gt, pred, min_depth_threshold=0.0):
    gt = np.asarray(gt)
    pred = np.asarray(pred)
    mask = gt != 0.0
    if mask.sum() == 0:
        return 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0
    p = pred[mask]
    g = gt[mask]
    eps = 1e-3 if g.max() > 1.0 else 1e-6
    keep = (p > eps) & (g > eps)
    if keep.sum() == 0:
        keep = g > eps
        if keep.sum() == 0:
            return 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0
        keep = keep & (p > 0)
        if keep.sum() == 0:
            return 1.0, g.max(), 0.0, 0.0, 0.0, 1.0, g.max()
    p = p[keep]
    g = g[keep]
    eps = 1e-3 if g.max() > 1.0 else 1e-6
    pc = np.maximum(p, eps)
    thresh = np.maximum(g / pc, pc / g)
    a1 = _clean((thresh < 1.25).mean())
    a2 = _clean((thresh < 1.25 ** 2).mean())
    a3 = _clean((thresh < 1.25 ** 3).mean())
    rmse = _clean(np.sqrt(((g - p) ** 2).mean()))
    abs_rel = _clean(np.mean(np.abs(g - p) / g))
    log_10 = _clean(np.abs(np.log10(np.maximum(g, eps)) - np.log10(pc)).mean())
    mae = _clean(np.abs(g - p).mean())
    return abs_rel, rmse, a1, a2, a3, log_10, mae
