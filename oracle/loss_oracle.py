"""CPU restatement of the masked depth loss (TEST INFRASTRUCTURE ONLY).

Restates
  * SIlogLoss.forward                       /root/reference/utils_loss.py:29-49
  * the loss assembly of the train loop     /root/reference/train.py:646-669
    (mask ``gt != 0``; ``* max_depth`` on both sides when cfg.dataset.depth_norm;
     'L1' | 'SIlog' | 'Combined' = l1_w * L1 + silog_w * SIlog)
  * the binaural/adabins mask ``gt > 0``    /root/reference/train_binaural_attention.py:402
Written on the four sufficient statistics (N, sum|e|, sum d, sum d^2) so the same numbers
are what the HIP reduction kernel and the 4-float data-parallel all-reduce carry.

Pinned by: tests/test_oracle_golden.py against tests/golden/loss_*.npz.
"""
from __future__ import annotations

import torch

SILOG_EPS = 1e-6    # utils_loss.py:19 default epsilon


def loss_stats(pred: torch.Tensor, gt: torch.Tensor, scale: float = 1.0,
               mask_mode: str = 'ne0', eps: float = SILOG_EPS):
    """(N, sum|p-g|, sum d, sum d^2), d = log(clamp(p,eps)) - log(clamp(g,eps)).

    ``scale`` is cfg.dataset.max_depth when depth_norm else 1 (train.py:649-652).
    """
    mask = (gt != 0.0) if mask_mode == 'ne0' else (gt > 0.0)
    p = pred[mask] * scale
    g = gt[mask] * scale
    n = torch.tensor(float(p.numel()), dtype=pred.dtype)
    s_abs = (p - g).abs().sum()
    d = torch.log(torch.clamp(p, min=eps)) - torch.log(torch.clamp(g, min=eps))   # utils_loss.py:37-41
    return n, s_abs, d.sum(), (d * d).sum()


def loss_from_stats(n, s_abs, s_d, s_d2, criterion: str, l1_weight: float = 0.5,
                    silog_weight: float = 0.5, silog_lambda: float = 0.5):
    """train.py:654-669 on the statistics; returns the scalar loss."""
    l1 = s_abs / n                                                    # nn.L1Loss (mean)
    var = s_d2 / n - silog_lambda * (s_d / n) ** 2                    # utils_loss.py:46
    silog = torch.sqrt(torch.clamp(var, min=0.0))                     # utils_loss.py:47
    if criterion == 'L1':
        return l1
    if criterion == 'SIlog':
        return silog
    if criterion == 'Combined':
        return l1_weight * l1 + silog_weight * silog
    raise ValueError(f'Unknown criterion: {criterion}')


def masked_loss(pred, gt, criterion='Combined', l1_weight=0.5, silog_weight=0.5,
                silog_lambda=0.5, scale=1.0, mask_mode='ne0'):
    return loss_from_stats(*loss_stats(pred, gt, scale, mask_mode), criterion,
                           l1_weight, silog_weight, silog_lambda)


def masked_loss_grad_numpy(pred, gt, criterion='Combined', l1_weight=0.5, silog_weight=0.5,
                           silog_lambda=0.5, scale=1.0, mask_mode='ne0', eps=SILOG_EPS):
    """Closed-form d loss / d pred in float64 numpy (what the HIP gradient kernel computes).

    dL1/dp   = sign(p*scale - g*scale) * scale / N
    dSIlog/dp = [ (d/N - lambda * mean_d / N) / silog ] * (1/p) * [p*scale >= eps]  (clamp gate)
    """
    import numpy as np
    p64 = pred.astype(np.float64) * scale
    g64 = gt.astype(np.float64) * scale
    mask = (gt != 0.0) if mask_mode == 'ne0' else (gt > 0.0)
    n = float(mask.sum())
    grad = np.zeros_like(p64)
    if n == 0:
        return grad
    w1 = {'L1': 1.0, 'SIlog': 0.0, 'Combined': l1_weight}[criterion]
    w2 = {'L1': 0.0, 'SIlog': 1.0, 'Combined': silog_weight}[criterion]
    grad += w1 * np.sign(p64 - g64) * scale / n
    if w2 != 0.0:
        pc = np.maximum(p64, eps)
        gc = np.maximum(g64, eps)
        d = np.where(mask, np.log(pc) - np.log(gc), 0.0)
        mean_d = d.sum() / n
        var = (d * d).sum() / n - silog_lambda * mean_d ** 2
        if var > 0.0:
            silog = np.sqrt(var)
            dd = (d / n - silog_lambda * mean_d / n) / silog
            gate = (p64 >= eps).astype(np.float64)         # torch clamp(min=eps) passes grad iff p >= eps
            grad += w2 * dd * gate / pc * scale
    return np.where(mask, grad, 0.0)
