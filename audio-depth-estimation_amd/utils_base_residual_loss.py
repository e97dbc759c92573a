"""Loss of the Base + Residual model, mirror of /root/reference/utils_base_residual_loss.py.

``BaseResidualLoss`` / ``AdaptiveBaseResidualLoss`` keep the reference's constructor arguments, the ``set_epoch`` /
``get_current_weights`` curriculum (:205-242) and the ``(total_loss, loss_dict)`` return convention.  ``forward``
evaluates the three terms on the device with the libadn kernels and returns VALUES (no autograd graph); training runs
through ``base_residual_engine.BaseResidualTrainer`` (``BaseResidualTrainer.from_criterion``), which fuses this loss
with the backward pass, the clip and the optimizer.  The FFT-based ``FrequencyAwareBaseResidualLoss`` ("experimental",
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

    def forward(self, base_depth, residual, final_depth, gt_depth, valid_mask=None):
        if valid_mask is None:
            raise NotImplementedError('the libadn loss implements the masked form (valid_mask = gt > 0) of the trainer')
        gt = gt_depth.contiguous().float()
        if not gt.is_cuda:
            raise RuntimeError('BaseResidualLoss runs on libadn HIP kernels only (no CPU path)')
        dev = gt.device
        base, resid, final = [t.contiguous().float() for t in (base_depth, residual, final_depth)]
        B, H, W = gt.shape[0], gt.shape[-2], gt.shape[-1]
        f32, f64 = dict(dtype=torch.float32, device=dev), dict(dtype=torch.float64, device=dev)
        ws = torch.empty(max(K.lowpass_workspace_bytes(B, H, W, self.lowpass_kernel), 1 << 16) // 4 + 4, **f32)
        struct = torch.empty_like(gt)
        K.lowpass(gt, self.lowpass_kernel, struct, ws)
        lstats, bstats = torch.zeros(4, **f64), torch.zeros(4, **f64)
        lws = torch.empty(4096 + 8, **f64)
        recon, terms, scratch = torch.zeros(1, **f32), torch.zeros(4, **f32), torch.empty_like(gt)
        l1w, sw = (0.0, self.lambda_recon) if self.use_silog else (self.lambda_recon, 0.0)
        crit, mm = recon_criterion(self.use_l1, self.use_silog)
        K.loss_stats(final, gt, 1.0, mm, 1e-6, lstats, lws)
        K.loss_finish(final, gt, 1.0, mm, 1e-6, lstats, crit, l1w, sw, self.silog_lambda, recon, scratch)
        K.baseres_stats(base, resid, struct, gt, recon, self.lambda_recon, self.lambda_base, self.lambda_sparse, bstats,
                        terms, ws)
        t = terms.cpu().tolist()
        rec = t[0] / self.lambda_recon if self.lambda_recon else 0.0
        return terms[3], {'total': t[3], 'recon': rec, 'base': t[1], 'sparse': t[2]}


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
