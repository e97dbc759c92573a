"""Knowledge-distillation loss (RGB teacher -> audio student), mirror of /root/reference/utils_distillation_loss.py.

``DistillationLoss`` / ``AdaptiveDistillationLoss`` keep the reference's constructor arguments, ``set_epoch`` /
``get_adaptive_weights`` schedule and the ``(total_loss, loss_dict)`` return convention.  ``forward`` evaluates the
terms on the device with the libadn kernels (csrc/adabins.hip) from an output dict of
``AdaBinsDistillationModel.forward``.  On outputs of a training-mode forward under autograd the total is a
differentiable node (analytic gradients of the five terms from the same kernels), so the reference's loop runs as
written; the fast path is ``adabins_engine.AdaBinsTrainer``, which fuses this loss with the student's backward, the
clip and the optimizer (``AdaBinsTrainer.from_criterion`` takes the weights from an instance of either class).
"""
import torch
import torch.nn as nn

from . import kernels as K


class DistillationLoss(nn.Module):
    """lambda_task * L1(audio, gt) + lambda_response * MSE(audio, rgb) + lambda_feature * mean cosine distance of
    x1..x5 + lambda_bin * (KL_T(mean logits) + MSE(bin centres)) + lambda_sparse * mean|residual|  (reference :20-238)."""

    def __init__(self, lambda_task=2.0, lambda_response=0.3, lambda_feature=0.2, lambda_bin=0.05, lambda_sparse=0.1,
                 temperature=4.0):
        super().__init__()
        self.lambda_task = lambda_task
        self.lambda_response = lambda_response
        self.lambda_feature = lambda_feature
        self.lambda_bin = lambda_bin
        self.lambda_sparse = lambda_sparse
        self.temperature = temperature

    def weights(self):
        return (self.lambda_task, self.lambda_response, self.lambda_feature, self.lambda_bin, self.lambda_sparse)

    @staticmethod
    def _nhwc(t):
        B, C, H, W = t.shape
        out = torch.empty(B, H, W, C, dtype=torch.float32, device=t.device)
        K.nchw_to_nhwc(t.contiguous().float(), out)
        return out

    def _evaluate(self, output, gt_depth, scale=1.0):
        """Terms f32[8] on the device plus everything the gradients need (NHWC f32 copies, statistics, dmean / dcent)."""
        a, r = output['audio'], output['rgb']
        gt = gt_depth.contiguous().float()
        dev = gt.device
        B, nb = a['bin_centers'].shape
        f32 = dict(dtype=torch.float32, device=dev)
        final = a['final_depth'].detach().contiguous().float()
        resid = a['residual'].detach().contiguous().float()
        n = final.numel()
        # final = clamp(base + residual): feed the stored final as "base" with a zero residual for the pixel terms,
        # then the |residual| statistic from the real residual
        zero = torch.zeros(n, **f32)
        stats = torch.zeros(4, dtype=torch.float64, device=dev)
        stats2 = torch.zeros(4, dtype=torch.float64, device=dev)
        ws = torch.empty(1 << 16, **f32)
        tmp = torch.empty(n, **f32)
        tfinal = r['final_depth'].detach().contiguous().float().view(-1) if r is not None else None
        K.distill_pix_stats(final.view(-1), zero, gt.view(-1), tfinal, float('inf'), tmp, stats, ws)
        K.distill_pix_stats(zero, resid.view(-1), gt.view(-1), None, float('inf'), tmp, stats2, ws)
        stats[3] = stats2[3]
        la = self._nhwc(a['bin_logits'].detach())
        HW = la.shape[1] * la.shape[2]
        wsp = torch.empty(max(K.pool_workspace_bytes(B, HW, nb, 1), 16) // 4, **f32)
        ms = torch.empty(B, nb, **f32)
        K.pool(la, None, B, HW, nb, 1, 1.0 / HW, ms, wsp)
        mt, fst, fch, fa_all, fr_all = None, [None] * 5, [0] * 5, [None] * 5, [None] * 5
        if r is not None:
            lr = self._nhwc(r['bin_logits'].detach())
            mt = torch.empty(B, nb, **f32)
            K.pool(lr, None, B, HW, nb, 1, 1.0 / HW, mt, wsp)
            for i, lv in enumerate(('x1', 'x2', 'x3', 'x4', 'x5')):
                fa, fr = self._nhwc(a['features'][lv].detach()), self._nhwc(r['features'][lv].detach())
                C, hw = fa.shape[-1], fa.shape[1] * fa.shape[2]
                w3 = torch.empty(max(K.pool_workspace_bytes(B, hw, C, 3), 16) // 4, **f32)
                fst[i] = torch.empty(B, 3, C, **f32)
                K.pool(fa, fr, B, hw, C, 3, 1.0, fst[i], w3)
                fch[i], fa_all[i], fr_all[i] = C, fa, fr
        cs = a['bin_centers'].detach().contiguous().float()
        ct = r['bin_centers'].detach().contiguous().float() if r is not None else None
        ev = dict(final=final, resid=resid, gt=gt, tfinal=tfinal, zero=zero, stats=stats, ms=ms, mt=mt, cs=cs, ct=ct, fst=fst,
                  fch=fch, fa=fa_all, fr=fr_all, logits_shape=tuple(a['bin_logits'].shape), B=B, nb=nb, HW=HW)
        self._small(ev, scale)
        return ev

    def _small(self, ev, scale):
        """distill_small with the loss weights times ``scale`` (the incoming gradient of the total loss)."""
        f32 = dict(dtype=torch.float32, device=ev['gt'].device)
        ev['terms'] = torch.zeros(8, **f32)
        ev['dmean'], ev['dcent'] = torch.empty(ev['B'], ev['nb'], **f32), torch.empty(ev['B'], ev['nb'], **f32)
        K.distill_small(ev['ms'], ev['mt'], ev['cs'], ev['ct'], ev['fst'], ev['fch'], ev['stats'], self.temperature,
                        tuple(w * scale for w in self.weights()), ev['terms'], ev['dmean'], ev['dcent'])

    def _leaf_grads(self, ev, scale):
        """d(scale * total) / d(final_depth, residual, bin_logits, bin_centers, x1..x5), NCHW f32 like the leaves."""
        lt, lr_, lf, lb, ls = (w * scale for w in self.weights())
        if scale != 1.0:
            self._small(ev, scale)
        f32 = dict(dtype=torch.float32, device=ev['gt'].device)
        n = ev['final'].numel()
        has_t = ev['tfinal'] is not None
        g_final, g_res, junk = torch.empty(n, **f32), torch.empty(n, **f32), torch.empty(n, **f32)
        inf = float('inf')
        K.distill_pix_grad(ev['final'].view(-1), ev['zero'], ev['gt'].view(-1), ev['tfinal'], inf, ev['stats'], lt,
                           lr_ if has_t else 0.0, 0.0, g_final, junk)
        K.distill_pix_grad(ev['zero'], ev['resid'].view(-1), ev['gt'].view(-1), None, inf, ev['stats'], 0.0, 0.0, ls, junk,
                           g_res)
        B, nb, H, W = ev['logits_shape']
        gl = torch.empty(B, H, W, nb, **f32)
        K.bcast_add(gl, ev['dmean'], 1.0 / (H * W), accumulate=False)
        g_logits = torch.empty(B, nb, H, W, **f32)
        K.nhwc_to_nchw(gl, g_logits)
        g_feats = [None] * 5
        if has_t:
            for i in range(5):
                fa = ev['fa'][i]
                ga = torch.zeros_like(fa)
                K.featcos_grad(fa, ev['fr'][i], ev['fst'][i], -lf / (5.0 * B * fa.shape[-1]), ga)
                g_feats[i] = torch.empty(B, fa.shape[-1], fa.shape[1], fa.shape[2], **f32)
                K.nhwc_to_nchw(ga, g_feats[i])
        shp = ev['final'].shape
        return [g_final.view(shp), g_res.view(shp), g_logits, ev['dcent']] + g_feats

    def forward(self, output, gt_depth, valid_mask=None):
        """Returns (total loss as a 0-dim device tensor, dict of python floats) like the reference (:147-238).
        ``valid_mask`` must be ``gt_depth > 0`` (what train_adabins_distillation.py:449 passes) or None.  When the
        student's outputs carry an autograd graph (model.train() under grad) the total is differentiable:
        ``loss.backward()`` sends the analytic gradients of the five terms into the student branch."""
        a = output['audio']
        if not gt_depth.is_cuda:
            raise RuntimeError('DistillationLoss runs on libadn HIP kernels only (no CPU path)')
        if valid_mask is None:
            raise NotImplementedError('DistillationLoss on libadn implements the masked form (valid_mask = gt > 0) that '
                                      'the reference trainer uses')
        leaves = [a['final_depth'], a['residual'], a['bin_logits'], a['bin_centers']] + \
                 [a['features'][lv] for lv in ('x1', 'x2', 'x3', 'x4', 'x5')]
        if torch.is_grad_enabled() and any(t.requires_grad for t in leaves):
            total = _DistillFunction.apply(self, output, gt_depth, *leaves)
            terms = self._last_terms
        else:
            terms = self._evaluate(output, gt_depth)['terms']
            total = terms[6]
        t = terms.cpu().tolist()
        loss_dict = {'task': t[0], 'response': t[1], 'feature': t[2], 'bin': t[3], 'bin_centers': t[4], 'sparse': t[5],
                     'total': t[6]}
        return total, loss_dict


class _DistillFunction(torch.autograd.Function):
    """total = DistillationLoss(output) as an autograd node over the student's output tensors."""

    @staticmethod
    def forward(ctx, crit, output, gt, *leaves):
        ev = crit._evaluate(output, gt)
        object.__setattr__(crit, '_last_terms', ev['terms'])
        ctx.crit, ctx.ev = crit, ev
        return ev['terms'][6].clone()

    @staticmethod
    def backward(ctx, gout):
        grads = ctx.crit._leaf_grads(ctx.ev, float(gout))
        return (None, None, None) + tuple(grads)


class AdaptiveDistillationLoss(nn.Module):
    """Curriculum over the distillation weights (reference :241-337)."""

    def __init__(self, max_epochs=200, temperature=4.0, lambda_sparse=0.1):
        super().__init__()
        self.max_epochs = max_epochs
        self.temperature = temperature
        self.lambda_sparse = lambda_sparse
        self.current_epoch = 0

    def set_epoch(self, epoch):
        self.current_epoch = epoch

    def get_adaptive_weights(self):
        progress = min(1.0, self.current_epoch / self.max_epochs)
        lambda_task = 2.0 + progress
        if progress < 0.1:
            lambda_response = 0.1
        else:
            lambda_response = 0.1 + 0.4 * (progress - 0.1) / 0.9
        if progress < 0.2:
            lambda_feature = 0.05
        elif progress < 0.5:
            lambda_feature = 0.05 + 0.25 * (progress - 0.2) / 0.3
        else:
            lambda_feature = 0.3 - 0.1 * (progress - 0.5) / 0.5
        lambda_bin = 0.05 - 0.03 * progress
        return {'task': lambda_task, 'response': lambda_response, 'feature': lambda_feature, 'bin': lambda_bin,
                'sparse': self.lambda_sparse}

    def criterion(self):
        w = self.get_adaptive_weights()
        return DistillationLoss(lambda_task=w['task'], lambda_response=w['response'], lambda_feature=w['feature'],
                                lambda_bin=w['bin'], lambda_sparse=w['sparse'], temperature=self.temperature)

    def forward(self, output, gt_depth, valid_mask=None):
        total_loss, loss_dict = self.criterion()(output, gt_depth, valid_mask)
        loss_dict['weights'] = self.get_adaptive_weights()
        return total_loss, loss_dict
