"""Loss of the Base + Residual model, mirror of /root/reference/utils_base_residual_loss.py.

``BaseResidualLoss`` / ``AdaptiveBaseResidualLoss`` keep the reference's constructor arguments, the ``set_epoch`` /
``get_current_weights`` curriculum (:205-242) and the ``(total_loss, loss_dict)`` return convention.  ``forward``
evaluates the three terms on the device with the libadn kernels; on the outputs of a training-mode forward under
autograd the total is a differentiable node, so the reference's loop runs as written.  The fast path is
``base_residual_engine.BaseResidualTrainer`` (``BaseResidualTrainer.from_criterion``), which fuses this loss with the
backward pass, the clip and the optimizer.  The FFT-based ``FrequencyAwareBaseResidualLoss`` ("experimental",
unused by the trainer) is out of scope.
"""
import torch
import torch.nn as nn

from . import kernels as K


def recon_criterion(use_l1, use_silog):
    """(libadn criterion code, mask mode) of the reconstruction term over valid = gt > 0: Combined with one active weight
    (= weighted SIlog or L1), or the masked MSE (criterion 4, squared-error statistics = mask mode | 4)."""
    return (2, 1) if (use_silog or use_l1) else (4, 1 | 4)


class BaseResidualLoss(nn.Module):
    """lambda_recon * recon(final, gt) + lambda_base * L1(base, lowpass(gt)) + lambda_sparse * mean|residual| over
    valid = gt > 0 (reference :28-160); recon = SIlog (use_silog), else L1 (use_l1), else MSE (:60-65)."""

    def __init__(self, lambda_recon=1.0, lambda_base=1.2, lambda_sparse=0.05, lowpass_kernel=16, use_l1=True,
                 use_silog=False, silog_lambda=0.5):
        super().__init__()
        self.lambda_recon = lambda_recon
        self.lambda_base = lambda_base
        self.lambda_sparse = lambda_sparse
        self.lowpass_kernel = lowpass_kernel
        self.use_l1 = use_l1
        self.use_silog = use_silog
        self.silog_lambda = silog_lambda

    def _evaluate(self, base_depth, residual, final_depth, gt_depth, scale=1.0, want_grads=False):
        """terms f32[4] (and, with ``want_grads``, d(scale*total)/d(base, residual, final) as f32 tensors)."""
        gt = gt_depth.contiguous().float()
        dev = gt.device
        base, resid, final = [t.detach().contiguous().float() for t in (base_depth, residual, final_depth)]
        B, H, W = gt.shape[0], gt.shape[-2], gt.shape[-1]
        f32, f64 = dict(dtype=torch.float32, device=dev), dict(dtype=torch.float64, device=dev)
        ws = torch.empty(max(K.lowpass_workspace_bytes(B, H, W, self.lowpass_kernel), 1 << 16) // 4 + 4, **f32)
        struct = torch.empty_like(gt)
        K.lowpass(gt, self.lowpass_kernel, struct, ws)
        lstats, bstats = torch.zeros(4, **f64), torch.zeros(4, **f64)
        lws = torch.empty(4096 + 8, **f64)
        recon, terms, g_final = torch.zeros(1, **f32), torch.zeros(4, **f32), torch.empty_like(gt)
        lrec, lbase, lsp = self.lambda_recon * scale, self.lambda_base * scale, self.lambda_sparse * scale
        l1w, sw = (0.0, lrec) if self.use_silog else (lrec, 0.0)
        crit, mm = recon_criterion(self.use_l1, self.use_silog)
        K.loss_stats(final, gt, 1.0, mm, 1e-6, lstats, lws)
        K.loss_finish(final, gt, 1.0, mm, 1e-6, lstats, crit, l1w, sw, self.silog_lambda, recon, g_final)
        K.baseres_stats(base, resid, struct, gt, recon, lrec, lbase, lsp, bstats, terms, ws)
        if not want_grads:
            return terms
        # d/d base and d/d residual of the two regularisers alone: the clamp's share of g_final is the model's backward
        g_base, g_res = torch.empty_like(gt), torch.empty_like(gt)
        K.baseres_grad(base, resid, struct, gt, torch.zeros_like(gt), float('inf'), bstats, lbase, lsp, g_base, g_res)
        return terms, (g_base.view(base_depth.shape), g_res.view(residual.shape), g_final.view(final_depth.shape))

    def forward(self, base_depth, residual, final_depth, gt_depth, valid_mask=None):
        """(total, loss_dict) like the reference (:67-160).  On the outputs of a training-mode forward under autograd the
        total is differentiable (train_base_residual.py's loss.backward() works as written)."""
        if valid_mask is None:
            raise NotImplementedError('the libadn loss implements the masked form (valid_mask = gt > 0) of the trainer')
        if not gt_depth.is_cuda:
            raise RuntimeError('BaseResidualLoss runs on libadn HIP kernels only (no CPU path)')
        leaves = (base_depth, residual, final_depth)
        if torch.is_grad_enabled() and any(t.requires_grad for t in leaves):
            total = _BaseResidualLossFunction.apply(self, gt_depth, *leaves)
            terms = self._last_terms
        else:
            terms = self._evaluate(base_depth, residual, final_depth, gt_depth)
            total = terms[3]
        t = terms.cpu().tolist()
        rec = t[0] / self.lambda_recon if self.lambda_recon else 0.0
        return total, {'total': t[3], 'recon': rec, 'base': t[1], 'sparse': t[2]}


class _BaseResidualLossFunction(torch.autograd.Function):
    """total = BaseResidualLoss(base, residual, final) as an autograd node over the model's three outputs."""

    @staticmethod
    def forward(ctx, crit, gt, base, resid, final):
        terms = crit._evaluate(base, resid, final, gt)
        object.__setattr__(crit, '_last_terms', terms)
        ctx.crit, ctx.gt = crit, gt
        ctx.save_for_backward(base, resid, final)
        return terms[3].clone()

    @staticmethod
    def backward(ctx, gout):
        base, resid, final = ctx.saved_tensors
        _, grads = ctx.crit._evaluate(base, resid, final, ctx.gt, scale=float(gout), want_grads=True)
        return (None, None) + grads


class AdaptiveBaseResidualLoss(nn.Module):
    """Curriculum: structure first (high lambda_base), accuracy later (high lambda_recon) (reference :163-242)."""

    def __init__(self, lambda_recon_init=0.3, lambda_base_init=2.0, lambda_sparse=0.05, warmup_epochs=50, lowpass_kernel=16,
                 use_silog=False, silog_lambda=0.5):
        super().__init__()
        self.lambda_recon_init = lambda_recon_init
        self.lambda_recon_final = 1.0
        self.lambda_base_init = lambda_base_init
        self.lambda_base_final = 0.3
        self.lambda_sparse = lambda_sparse
        self.warmup_epochs = warmup_epochs
        self.lowpass_kernel = lowpass_kernel
        self.current_epoch = 0
        self.base_loss = BaseResidualLoss(lambda_recon=lambda_recon_init, lambda_base=lambda_base_init,
                                          lambda_sparse=lambda_sparse, lowpass_kernel=lowpass_kernel, use_silog=use_silog,
                                          silog_lambda=silog_lambda)

    def set_epoch(self, epoch):
        self.current_epoch = epoch
        if epoch < self.warmup_epochs:
            alpha = epoch / self.warmup_epochs
            self.base_loss.lambda_recon = self.lambda_recon_init + alpha * (self.lambda_recon_final - self.lambda_recon_init)
            self.base_loss.lambda_base = self.lambda_base_init + alpha * (self.lambda_base_final - self.lambda_base_init)
        else:
            self.base_loss.lambda_recon = self.lambda_recon_final
            self.base_loss.lambda_base = self.lambda_base_final

    def forward(self, base_depth, residual, final_depth, gt_depth, valid_mask=None):
        return self.base_loss(base_depth, residual, final_depth, gt_depth, valid_mask)

    def get_current_weights(self):
        return {'lambda_recon': self.base_loss.lambda_recon, 'lambda_base': self.base_loss.lambda_base,
                'lambda_sparse': self.base_loss.lambda_sparse}
