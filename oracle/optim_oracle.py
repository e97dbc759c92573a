"""CPU restatement of gradient clipping + optimizer step (TEST INFRASTRUCTURE ONLY).

Restates what /root/reference/train.py:689-691 calls:
  * torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)   (train.py:689)
      total = sqrt(sum_i ||g_i||_2^2); coef = max_norm / (total + 1e-6); g *= min(coef, 1)
  * torch.optim.AdamW(params, lr)  (train.py:473-474)  -> betas (0.9, 0.999), eps 1e-8,
      weight_decay 0.01 (torch default; train.py passes only lr), decoupled decay
  * torch.optim.Adam(params, lr)   (train.py:471-472)  -> weight_decay 0
      (train_binaural_attention.py:314-318 passes weight_decay -> L2-in-gradient)
  * torch.optim.SGD(params, lr)    (train.py:475-476)  -> plain p -= lr * g
in float64 numpy on flat arrays.  torch.optim / clip_grad_norm_ live in the third-party
dependency torch (2.10.0 in this image), not under /root/reference; the restatement is
pinned against torch's own implementation by tests/golden/optim_*.npz.
"""
from __future__ import annotations

import numpy as np


def clip_coef(grads, max_norm=1.0):
    total = np.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads))
    coef = max_norm / (total + 1e-6)
    return total, min(coef, 1.0)


def adamw_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01,
               decoupled=True, grad_scale=1.0):
    """One Adam/AdamW update (torch.optim single-tensor path, amsgrad=False, maximize=False).

    ``step`` is the 1-based step count AFTER increment.  Returns (p, m, v) in float64.
    """
    p = p.astype(np.float64); g = g.astype(np.float64) * grad_scale
    m = m.astype(np.float64); v = v.astype(np.float64)
    if decoupled:
        p = p * (1.0 - lr * weight_decay)
    elif weight_decay != 0.0:
        g = g + weight_decay * p
    m = beta1 * m + (1.0 - beta1) * g
    v = beta2 * v + (1.0 - beta2) * g * g
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = np.sqrt(v) / np.sqrt(bc2) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


def sgd_step(p, g, lr, grad_scale=1.0):
    return p.astype(np.float64) - lr * g.astype(np.float64) * grad_scale
