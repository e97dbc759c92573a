"""Loss functions for depth estimation (mirror of the reference's utils_loss.py).

``SIlogLoss(lambda_scale=0.5, epsilon=1e-6)(pred, target)`` keeps the reference signature and value
(/root/reference/utils_loss.py:9-49): sqrt(clamp(mean(d^2) - lambda * mean(d)^2, 0)) with
d = log(clamp(pred, eps)) - log(clamp(target, eps)) over ALL elements passed in (the caller has
already gathered the valid pixels, train.py:657).  The reductions and the gradient run in libadn
(adn_loss_stats / adn_loss_finish); ``MaskedDepthLoss`` is the fused mask + L1/SIlog/Combined
assembly of train.py:646-669 that never materialises ``pred[mask]``.
"""
import torch
import torch.nn as nn

from . import kernels as K

_CRIT = {'L1': 0, 'SIlog': 1, 'Combined': 2}
_MASK = {'ne0': 0, 'gt0': 1, 'all': 2}


class _MaskedLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, criterion, l1_w, silog_w, lam, scale, mask_mode, eps):
        if not pred.is_cuda:
            raise RuntimeError('libadn loss kernels need HIP device tensors (no CPU path)')
        p = pred.detach().contiguous().float()
        t = target.detach().contiguous().float().expand_as(p).contiguous()
        dev = p.device
        stats = torch.empty(4, dtype=torch.float64, device=dev)
        ws = torch.empty(4096 + 8, dtype=torch.float64, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        grad = torch.empty_like(p) if pred.requires_grad else None
        K.loss_stats(p, t, scale, mask_mode, eps, stats, ws)
        K.loss_finish(p, t, scale, mask_mode, eps, stats, criterion, l1_w, silog_w, lam, loss, grad)
        ctx.save_for_backward(grad)
        ctx.shape = pred.shape
        return loss[0].to(pred.dtype)

    @staticmethod
    def backward(ctx, gloss):
        (grad,) = ctx.saved_tensors
        g = (grad * gloss).view(ctx.shape) if grad is not None else None
        return g, None, None, None, None, None, None, None, None


class SIlogLoss(nn.Module):
    """Scale-Invariant Logarithmic Loss (same constructor and call signature as the reference)."""

    def __init__(self, lambda_scale=0.5, epsilon=1e-6):
        super().__init__()
        self.lambda_scale = lambda_scale
        self.epsilon = epsilon

    def forward(self, pred, target):
        return _MaskedLossFn.apply(pred, target, _CRIT['SIlog'], 0.0, 1.0, float(self.lambda_scale), 1.0,
                                   _MASK['all'], float(self.epsilon))


class MaskedDepthLoss(nn.Module):
    """valid_mask + (optional *max_depth) + L1 / SIlog / Combined in two kernels (train.py:646-669)."""

    def __init__(self, criterion='Combined', l1_weight=0.5, silog_weight=0.5, silog_lambda=0.5, scale=1.0,
                 mask_mode='ne0', epsilon=1e-6):
        super().__init__()
        self.criterion, self.l1_weight, self.silog_weight = criterion, l1_weight, silog_weight
        self.silog_lambda, self.scale, self.mask_mode, self.epsilon = silog_lambda, scale, mask_mode, epsilon

    def forward(self, depth_pred, gtdepth):
        return _MaskedLossFn.apply(depth_pred, gtdepth, _CRIT[self.criterion], float(self.l1_weight),
                                   float(self.silog_weight), float(self.silog_lambda), float(self.scale),
                                   _MASK[self.mask_mode], float(self.epsilon))
