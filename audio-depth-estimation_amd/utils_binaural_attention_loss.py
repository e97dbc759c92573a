"""Edge-aware / smoothness loss of the binaural model family, mirror of
/root/reference/utils_binaural_attention_loss.py (``BinauralAttentionLoss`` :15-156, ``AdaptiveBinauralAttentionLoss``
:159-230; marked deprecated there -- no training script of the reference uses it -- but part of its loss set,
SURVEY section 8f-4).

Same constructor arguments, the same ``(total_loss, loss_dict)`` return convention (``loss_dict`` holds Python floats: a
host sync per call, like the reference's ``.item()``) and the same epoch curriculum.  The arithmetic runs in libadn
(adn_edge_loss: Sobel responses, the 3x3-dilated validity mask, the three masked means and their gradient); ``total_loss``
is an autograd node, so ``total_loss.backward()`` delivers d loss / d pred to whatever produced ``pred_depth``.
"""
import torch
import torch.nn as nn

from . import kernels as K


class _EdgeLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt, lambdas):
        if not pred.is_cuda:
            raise RuntimeError('BinauralAttentionLoss runs on libadn HIP kernels only (no CPU path)')
        p, g = pred.detach().contiguous().float(), gt.detach().contiguous().float()
        B, H, W = p.shape[0], p.shape[-2], p.shape[-1]
        dev = p.device
        stats = torch.zeros(5, dtype=torch.float64, device=dev)
        terms = torch.zeros(4, dtype=torch.float32, device=dev)
        grad = torch.empty_like(p) if pred.requires_grad else None
        ws = torch.empty(K.edge_loss_workspace_bytes(B, H, W) // 4 + 4, dtype=torch.float32, device=dev)
        K.edge_loss(p, g, lambdas, stats, terms, grad, ws)
        ctx.save_for_backward(grad if grad is not None else terms)
        ctx.has_grad = grad is not None
        ctx.mark_non_differentiable(terms)
        return terms[3].clone(), terms

    @staticmethod
    def backward(ctx, gtotal, _gterms):
        if not ctx.has_grad:
            return None, None, None
        (grad,) = ctx.saved_tensors
        return grad * gtotal, None, None


class BinauralAttentionLoss(nn.Module):
    """lambda_recon * L1(valid) + lambda_edge * edge-aware + lambda_smooth * smoothness (reference :15-156)."""

    def __init__(self, lambda_recon=1.0, lambda_edge=0.2, lambda_smooth=0.1):
        super().__init__()
        self.lambda_recon = lambda_recon
        self.lambda_edge = lambda_edge
        self.lambda_smooth = lambda_smooth
        # the reference registers its Sobel filters as buffers (state_dict keys sobel_x / sobel_y): keep them
        self.register_buffer('sobel_x', torch.tensor([[[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]]], dtype=torch.float32).unsqueeze(0))
        self.register_buffer('sobel_y', torch.tensor([[[-1, -2, -1], [0, 0, 0], [1, 2, 1]]], dtype=torch.float32).unsqueeze(0))

    def forward(self, pred_depth, gt_depth):
        total, terms = _EdgeLossFn.apply(pred_depth, gt_depth, (self.lambda_recon, self.lambda_edge, self.lambda_smooth))
        t = terms.detach().cpu().tolist()
        return total, {'loss_total': t[3], 'loss_recon': t[0], 'loss_edge': t[1], 'loss_smooth': t[2]}


class AdaptiveBinauralAttentionLoss(nn.Module):
    """Curriculum of the reference (:159-230): reconstruction only during the warm-up, the edge term ramps in over the
    next two warm-up lengths, the smoothness term over one more."""

    def __init__(self, warmup_epochs=20, total_epochs=200):
        super().__init__()
        self.warmup_epochs = warmup_epochs
        self.total_epochs = total_epochs
        self.base_loss = BinauralAttentionLoss(lambda_recon=1.0, lambda_edge=0.0, lambda_smooth=0.0)

    def weights_at(self, epoch):
        w = self.warmup_epochs
        if epoch < w:
            return 1.0, 0.0, 0.0
        if epoch < 3 * w:
            return 1.0, 0.2 * (epoch - w) / (2 * w), 0.0
        return 1.0, 0.2, 0.1 * min((epoch - 3 * w) / w, 1.0)

    def forward(self, pred_depth, gt_depth, epoch):
        lr, le, ls = self.weights_at(epoch)
        self.base_loss.lambda_recon, self.base_loss.lambda_edge, self.base_loss.lambda_smooth = lr, le, ls
        total, d = self.base_loss(pred_depth, gt_depth)
        d['lambda_recon'], d['lambda_edge'], d['lambda_smooth'] = lr, le, ls
        return total, d
