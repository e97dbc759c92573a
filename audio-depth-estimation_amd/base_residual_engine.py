"""Base + Residual model on libadn: one op tape (shared encoder, two decoders, two 1x1 heads) and the fused step.

Replaces /root/reference/models/base_residual_model.py:151-202 and the training step of
/root/reference/train_base_residual.py:375-388 with utils_base_residual_loss.BaseResidualLoss (:72-160): the
reconstruction term (masked L1 or SIlog of the final depth) reuses the masked-loss kernels of the U-Net path
(adn_loss_stats / adn_loss_finish), the structural target is adn_lowpass(gt), the rest is csrc/baseres.hip.
The encoder activations x1..x4 receive gradients from both decoders (accumulated in their gradient buffers).
"""
from __future__ import annotations

import torch

from . import kernels as K
from .dc_engine import DCEngine, Head1x1, flag_solo, mark_tail_writers
from .engine import GraphedStep


class BaseResidualEngine(DCEngine):
    def __init__(self, module, compute_dtype=torch.bfloat16):
        super().__init__(module, None, compute_dtype, 'BaseResidualDepthNet')

    def _prepare_net(self, x):
        if not self._bound():
            self.bind_parameters()
        B, Cin, H, W = x.shape
        key = (B, Cin, H, W, x.device)
        if self._shape_enter(key):
            return
        m = self.module
        if Cin != m.input_channels:
            raise RuntimeError(f'expected input[{B}, {Cin}, {H}, {W}] to have {m.input_channels} channels, but got {Cin} '
                               'channels instead')
        # input size != output_size: both heads resize their ACTIVATED map (bilinear, align_corners=False) before the sum
        # and the clamp (base_residual_model.py:185-211)
        S = m.output_size
        osz = S if (H != S or W != S) else None
        self.B, self.dev = B, x.device
        self._scratch = {}
        self.epc = 8 if self.dtype == torch.bfloat16 else 4
        self.pairs = []
        inp = self.thin_input('x', Cin, H, W)
        ops, f = m.inc.adn_ops([inp], 'x1', H, W)
        feats = [f]
        for i, down in enumerate((m.down1, m.down2, m.down3, m.down4)):
            o, f = down.adn_ops(feats[-1], f'x{i + 2}')
            ops += o
            feats.append(f)
        ends = {}
        for tag, ups in (('base', (m.base_up1, m.base_up2, m.base_up3, m.base_up4)),
                         ('res', (m.res_up1, m.res_up2, m.res_up3, m.res_up4))):
            d = feats[4]
            for i, up in enumerate(ups):
                o, d = up.adn_ops(d, feats[3 - i], f'{tag}.d{4 - i}')
                ops += o
            ends[tag] = d
        self.inputs, self.ops = [(inp, 0, Cin)], ops
        self.head_base = Head1x1(ends['base'], m.base_head, 1, m.max_depth, osz, clamp_after_resize=False)   # sigmoid * max_depth
        self.head_res = Head1x1(ends['res'], m.res_head, 2, 0.3 * m.max_depth, osz, clamp_after_resize=False)  # tanh * 0.3 max_depth
        acts = {}
        for op in ops:
            for a in list(getattr(op, 'srcs', [])) + [getattr(op, 'src', None), getattr(op, 'out', None)]:
                if a is not None:
                    acts[id(a)] = a
        self.acts = list(acts.values())
        for a in self.acts:
            flag_solo(a)
            a.alloc(B, self.dtype, x.device)
        mark_tail_writers(ops + [self.head_base, self.head_res])
        ws = 1 << 16
        for op in ops + [self.head_base, self.head_res]:
            op.prepare(self)
            ws = max(ws, op.workspace_bytes(self))
        f32 = dict(dtype=torch.float32, device=x.device)
        self.final = torch.empty(B, 1, S, S, **f32) if osz else torch.empty(B, 1, H, W, **f32)
        self.workspace = torch.empty(max(ws, K.lowpass_workspace_bytes(B, H, W, 64)) // 4 + 4, **f32)
        self.weights_dirty = True
        self._shape_key = key

    def forward_net(self, x, training):
        if not x.is_cuda:
            raise RuntimeError('BaseResidualDepthNet needs a HIP device tensor (libadn has no CPU path)')
        x = x.contiguous().float()
        self._prepare_net(x)
        if self.weights_dirty or self._packed_version != self._version_sum():
            self._pack_weights()
        self.load_input(x)
        for op in self.ops:
            op.fwd(self, training)
        self.head_base.fwd(self, training)
        self.head_res.fwd(self, training)
        K.clamp_add(self.head_base.result, self.head_res.result, self.module.max_depth, self.final)
        return self.head_base.result, self.head_res.result, self.final

    def run(self, x, training):
        b, r, f = self.forward_net(x, training)
        return b.clone(), r.clone(), f.clone()

    def backward_net(self, dbase, dres):
        for a in self.acts:
            a.written = False
        self._final = set(id(p) for p, _, _ in self.param_meta if not p.requires_grad)
        self._wm = len(self.param_meta)
        self.head_res.bwd_head(self, dres)
        self.head_base.bwd_head(self, dbase)
        for op in reversed(self.ops):
            if op.out.needs_grad:
                op.bwd(self)


class _BaseResidualFunction(torch.autograd.Function):
    """torch.autograd bridge: parameters are inputs, the outputs are (base_depth, residual, final_depth), so
    ``criterion(base, res, final, gt, mask)[0].backward()`` of train_base_residual.py reaches the parameters.
    final = clamp(base + residual, 0, max_depth) (base_residual_model.py:150-152) routes its gradient into both heads."""

    @staticmethod
    def forward(ctx, x, engine, *params):
        b, r, f = engine.forward_net(x, True)
        engine.autograd_pass = getattr(engine, 'autograd_pass', 0) + 1
        ctx.engine, ctx.stamp = engine, engine.autograd_pass
        ctx.set_materialize_grads(False)
        return b.clone(), r.clone(), f.clone()

    @staticmethod
    def backward(ctx, g_base, g_res, g_final):
        eng = ctx.engine
        if ctx.stamp != eng.autograd_pass:
            raise RuntimeError('BaseResidualDepthNet: backward through a forward whose activations were overwritten by a '
                               'later training forward of the same module')
        base, resid = eng.head_base.result, eng.head_res.result
        add_ = lambda dst, src: K.bcast_add(dst.view(-1, 1, 1, 1), src.contiguous().float().view(-1, 1), 1.0, accumulate=True)
        dbase, dres = torch.zeros_like(base), torch.zeros_like(resid)
        if g_final is not None:
            s = base.clone()
            add_(s, resid)
            masked = torch.empty_like(s)
            K.clamp_range(s, eng.module.max_depth, masked, g=g_final.contiguous().float())
            add_(dbase, masked)
            add_(dres, masked)
        if g_base is not None:
            add_(dbase, g_base)
        if g_res is not None:
            add_(dres, g_res)
        eng.backward_net(dbase, dres)
        return (None, None) + tuple(eng.grad_view(p) if p.requires_grad else None for p, _, _ in eng.param_meta)


def run_base_residual(engine, x, training):
    if not engine._bound():
        engine.bind_parameters()
    if training and torch.is_grad_enabled() and any(p.requires_grad for p, _, _ in engine.param_meta):
        return _BaseResidualFunction.apply(x, engine, *[p for p, _, _ in engine.param_meta])
    with torch.no_grad():
        return engine.run(x, training)


class BaseResidualTrainer(GraphedStep):
    """One fused step of train_base_residual.py:375-388: forward, BaseResidualLoss (valid = gt > 0), backward,
    clip_grad_norm_(1.0), optimizer."""

    def __init__(self, engine, lambda_recon=1.0, lambda_base=1.2, lambda_sparse=0.05, lowpass_kernel=16, use_l1=True,
                 use_silog=False, silog_lambda=0.5, optimizer='AdamW', lr=1e-4, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=None, clip_norm=1.0, ddp=None):
        self.engine = engine
        self.use_l1 = use_l1
        self.lambda_recon, self.lambda_base, self.lambda_sparse = lambda_recon, lambda_base, lambda_sparse
        self.k, self.use_silog, self.silog_lambda = lowpass_kernel, use_silog, silog_lambda
        self.opt_kind = {'AdamW': 0, 'Adam': 1, 'SGD': 2}[optimizer]
        self.lr, self.betas, self.eps = float(lr), betas, float(eps)
        self.weight_decay = float((0.01 if optimizer == 'AdamW' else 0.0) if weight_decay is None else weight_decay)
        self.clip_norm = clip_norm
        self.ddp = ddp                    # ddp.GradientAllReducer: one process per GPU, DataParallel semantics
        self._ready = False

    def enable_graph(self, after_steps=3):
        if self.ddp is not None:
            raise RuntimeError('the hipGraph step is not combined with the data-parallel reducer (host-side collectives)')
        super().enable_graph(after_steps)

    @classmethod
    def from_criterion(cls, engine, criterion, **kw):
        """Build from a utils_base_residual_loss.BaseResidualLoss / AdaptiveBaseResidualLoss instance."""
        c = getattr(criterion, 'base_loss', criterion)
        return cls(engine, c.lambda_recon, c.lambda_base, c.lambda_sparse, c.lowpass_kernel, c.use_l1, c.use_silog,
                   c.silog_lambda, **kw)

    def set_criterion(self, criterion):
        c = getattr(criterion, 'base_loss', criterion)
        self.set_weights(c.lambda_recon, c.lambda_base)

    def load_state_dict(self, sd, device):
        """Restore a torch.optim state dict (optim_state.py; the 'optimizer_state_dict' entry of the checkpoints written
        by train_dc._run or by the reference's torch optimizer); round 1's flat layout is re-sliced per parameter or rejected."""
        from . import optim_state
        if not self.engine._bound():
            self.engine.bind_parameters()
        self._setup_optimizer(torch.device(device))
        if optim_state.is_torch_format(sd):
            step, group = optim_state.import_state(sd, self.engine.param_meta, self.engine._view, self.exp_avg,
                                                   self.exp_avg_sq)
            optim_state.adopt_group(self, group)
            self.state[0] = float(step)
        elif 'exp_avg' in sd:
            step = optim_state.import_legacy_flat(sd, self.engine.param_meta, self.exp_avg, self.exp_avg_sq)
            self.state[0] = float(step)

    def _setup_optimizer(self, dev):
        if getattr(self, '_opt_ready', False):
            return
        eng = self.engine
        self.state = torch.zeros(8, dtype=torch.float64, device=dev)
        self.exp_avg, self.exp_avg_sq = torch.zeros_like(eng.flat_p), torch.zeros_like(eng.flat_p)
        self._opt_ready = True

    def set_weights(self, lambda_recon, lambda_base):
        """AdaptiveBaseResidualLoss.set_epoch (utils_base_residual_loss.py:210-229) result."""
        self.lambda_recon, self.lambda_base = lambda_recon, lambda_base

    _SCRATCH = ('lstats', 'bstats', 'loss_ws', 'norm_ws', 'recon', 'terms', 'struct', 'gfinal', 'dbase', 'dres')

    def _setup(self, pred):
        """Loss / gradient scratch of one batch shape.  Every shape keeps its own set for the trainer's lifetime: a captured
        hipGraph holds raw pointers into the set it was captured with, and an eager step on another shape in between (a
        ragged last batch) must not free or rebind those buffers (round-2 advisor finding)."""
        eng, dev = self.engine, pred.device
        f64, f32 = dict(dtype=torch.float64, device=dev), dict(dtype=torch.float32, device=dev)
        self._setup_optimizer(dev)
        sets = self.__dict__.setdefault('_scratch_sets', {})
        key = tuple(pred.shape)
        if key not in sets:
            sets[key] = dict(lstats=torch.zeros(4, **f64), bstats=torch.zeros(4, **f64),
                             loss_ws=torch.empty(4096 + 8, **f64), norm_ws=torch.empty(1024 + 8, **f64),
                             recon=torch.zeros(1, **f32), terms=torch.zeros(4, **f32),
                             struct=torch.empty_like(pred), gfinal=torch.empty_like(pred),
                             dbase=torch.empty_like(pred), dres=torch.empty_like(pred))
        for name in self._SCRATCH:
            setattr(self, name, sets[key][name])
        if not self._ready:
            self.bucket_norm = None
            if self.ddp is not None:
                self.ddp.attach(eng)
                if self.clip_norm is not None and eng.flat_g.is_cuda:
                    self.bucket_norm = self.ddp.enable_bucket_norm()
        self._ready = True

    def state_dict(self):
        from . import optim_state
        eng = self.engine
        if not eng._bound():
            eng.bind_parameters()
        ready = getattr(self, '_opt_ready', False)
        step = int(self.state[0].item()) if ready else 0
        return optim_state.export_state(eng.param_meta, eng._view, self.exp_avg if ready else None,
                                        self.exp_avg_sq if ready else None, step, self.opt_kind, self.lr, self.betas,
                                        self.eps, self.weight_decay)

    def step(self, x, gt):
        """Returns (total loss 0-dim device tensor, terms f32[4] = weighted recon, mean|base-struct|, mean|res|, total)."""
        return self._graphed(x, gt)

    def _step_impl(self, x, gt):
        eng = self.engine
        base, resid, final = eng.forward_net(x, True)
        gt = gt.contiguous().float()
        if not self._ready or self.struct.shape != final.shape:
            self._setup(final)
        K.lowpass(gt, self.k, self.struct, eng.workspace)
        from .utils_base_residual_loss import recon_criterion
        crit, mm = recon_criterion(self.use_l1, self.use_silog)   # weighted SIlog / L1 (Combined, one weight) or masked MSE
        l1w, sw = (0.0, self.lambda_recon) if self.use_silog else (self.lambda_recon, 0.0)
        K.loss_stats(final, gt, 1.0, mm, 1e-6, self.lstats, self.loss_ws)
        if self.ddp is not None:          # one global-batch loss, as under DataParallel (base_residual_model.py:266-269)
            self.ddp.all_reduce_loss_stats(self.lstats)
        K.loss_finish(final, gt, 1.0, mm, 1e-6, self.lstats, crit, l1w, sw, self.silog_lambda, self.recon, self.gfinal)
        K.baseres_stats(base, resid, self.struct, gt, self.recon, self.lambda_recon, self.lambda_base, self.lambda_sparse,
                        self.bstats, self.terms, eng.workspace)
        if self.ddp is not None:
            self.ddp.all_reduce_loss_stats(self.bstats)
            n = self.bstats[0].clamp_min(1.0)
            self.terms[1], self.terms[2] = self.bstats[1] / n, self.bstats[2] / n
            self.terms[3] = self.terms[0] + self.lambda_base * self.terms[1] + self.lambda_sparse * self.terms[2]
            self.ddp.begin_backward()
        K.baseres_grad(base, resid, self.struct, gt, self.gfinal, eng.module.max_depth, self.bstats, self.lambda_base,
                       self.lambda_sparse, self.dbase, self.dres)
        eng.backward_net(self.dbase, self.dres)
        if self.ddp is not None:
            self.ddp.finish()
        if self.clip_norm is not None and self.bucket_norm is not None:
            K.grad_norm_ranges(eng.flat_g, None, self.bucket_norm, float(self.clip_norm), self.state, self.norm_ws)
        elif self.clip_norm is not None:
            K.grad_norm(eng.flat_g, float(self.clip_norm), self.state, self.norm_ws)
        K.optimizer_step(eng.flat_p, eng.flat_g, self.exp_avg, self.exp_avg_sq, self.opt_kind, self.lr, self.betas[0],
                         self.betas[1], self.eps, self.weight_decay, self.clip_norm is not None, self.state,
                         bf16_copy=eng.flat_w16)
        eng.weights_dirty = True
        eng.s2_fresh = eng.flat_w16 is not None
        return self.terms[3], self.terms
