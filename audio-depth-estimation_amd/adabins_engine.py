"""AdaBins distillation model on libadn: two DoubleConv op tapes (RGB teacher, audio student), the bin predictor,
the soft-binning head, the residual head and the fused distillation step.

Replaces what PyTorch dispatches for /root/reference/models/adabins_distillation_model.py:301-426 and the training
step of /root/reference/train_adabins_distillation.py:445-456 with utils_distillation_loss.DistillationLoss
(:147-238).  Data layout as in dc_engine.py; additionally per branch: pooled bottleneck g [B,512] f32, hidden
activations / bin widths / centres f32, bin logits [B,H,W,n_bins] in the compute dtype (never expanded to
f32 NCHW unless the caller asks for ``bin_logits``), base depth / residual / final depth f32 [B,1,H,W].
The teacher's parameters are a contiguous prefix of the flat parameter buffer (parameters() order); it never
receives gradients (the reference runs it under no_grad), so the clip norm and the optimizer act on the suffix.
"""
from __future__ import annotations

import torch

from . import kernels as K
from .dc_engine import Act, ConvBNReLU, DCEngine, Head1x1, Op, flag_solo, mark_tail_writers
from .engine import GraphedStep
from ._lib import EPI_ACT, EPI_ADD, GEMM_S1


class ConvLinear(Op):
    """Conv2d 1x1 with bias, no norm / activation (the class head, adabins_distillation_model.py:192)."""

    def __init__(self, src, conv, out):
        self.src, self.conv, self.out = src, conv, out
        out.producer = self
        src.consumers.append(self)

    def prepare(self, eng):
        T, dev, B = eng.dtype, eng.dev, eng.B
        s, o = self.src, self.out
        self.w_fwd = torch.empty(o.C, K.s1_row_stride(T, 1, s.C), dtype=T, device=dev)
        self.w_dg = torch.empty(s.C, K.s1_row_stride(T, 1, o.C), dtype=T, device=dev) if s.needs_grad else None
        q = lambda n, segs, cin: K.igemm_query(T, GEMM_S1, B, o.H, o.W, cin, 0, n, segs, ks=1)[1]
        self._ws = max(q(o.C, [o.C], s.C), q(s.C, [s.C], o.C) if s.needs_grad else 0,
                       K.wgrad_workspace_bytes(T, B, o.H, o.W, o.C, 0, s.C, 0, ks=1),
                       K.channel_sum_workspace_bytes(B * o.H * o.W, o.C))

    def workspace_bytes(self, eng):
        return self._ws

    def pack(self, eng):
        master = eng._flat_slice(eng.flat_p, self.conv.weight)
        K.pack_rows(master, self.out.C, 1, self.src.C, self.w_fwd)
        if self.w_dg is not None:
            K.pack_transpose_taps(master, self.out.C, 1, self.src.C, self.w_dg, flip=False)

    def fwd(self, eng, training):
        o = self.out
        K.igemm(eng.dtype, GEMM_S1, eng.B, o.H, o.W, self.src.data, None, self.w_fwd, o.C, EPI_ACT,
                [K.Seg(o.C, out0=o.data, bias=self.conv.bias, slope=1.0)], eng.workspace, ks=1)

    def bwd(self, eng):
        o, s = self.out, self.src
        fg = lambda p: eng._flat_slice(eng.flat_g, p)
        K.wgrad(eng.dtype, eng.B, o.H, o.W, o.grad, None, s.data, None, fg(self.conv.weight), eng.workspace, ks=1)
        K.channel_sum(o.grad, eng.B * o.H * o.W, o.C, o.C, fg(self.conv.bias), eng.workspace)
        K.igemm(eng.dtype, GEMM_S1, eng.B, o.H, o.W, o.grad, None, self.w_dg, s.C, EPI_ADD,
                [K.Seg(s.C, out0=s.grad, accumulate=s.written)], eng.workspace, ks=1)
        s.written = True


class BinPredictor(Op):
    """AdaBinsBinPredictor.forward (:127-149) placed right behind the producer of x5 in the tape, so that in the
    reversed tape its backward runs after every other consumer of x5 and before x5's own backward."""

    def __init__(self, x5, mod, branch):
        self.src, self.mod, self.branch = x5, mod, branch
        self.out = x5                       # (tape bookkeeping: backward runs iff x5 needs a gradient)
        x5.consumers.append(self)

    def prepare(self, eng):
        B, dev = eng.B, eng.dev
        x = self.src
        lin1, lin2 = self.mod.predictor[0], self.mod.predictor[3]
        self.Cb, self.Hd, self.nb = lin1.in_features, lin1.out_features, lin2.out_features
        self.p = float(self.mod.predictor[2].p)
        f32 = dict(dtype=torch.float32, device=dev)
        br = self.branch
        br.g = torch.empty(B, self.Cb, **f32)
        br.h1 = torch.empty(B, self.Hd, **f32)
        br.widths, br.centers = torch.empty(B, self.nb, **f32), torch.empty(B, self.nb, **f32)
        self.mask = torch.empty(B, self.Hd, dtype=torch.uint8, device=dev)
        if x.needs_grad:
            self.dW2p = torch.empty(B, self.nb * self.Hd, **f32)
            self.db2p = torch.empty(B, self.nb, **f32)
            self.dW1p = torch.empty(B, self.Hd * self.Cb, **f32)
            self.db1p = torch.empty(B, self.Hd, **f32)
            self.dg = torch.empty(B, self.Cb, **f32)
            br.dcent = torch.empty(B, self.nb, **f32)
        self._ws = max(K.pool_workspace_bytes(B, x.H * x.W, x.C, 1), K.channel_sum_workspace_bytes(B, self.Hd * self.Cb))
        self.draws = 0

    def workspace_bytes(self, eng):
        return self._ws

    def _params(self, eng):
        lin1, lin2 = self.mod.predictor[0], self.mod.predictor[3]
        fp = lambda p: eng._flat_slice(eng.flat_p, p)
        return fp(lin1.weight).view(self.Hd, self.Cb), fp(lin1.bias), fp(lin2.weight).view(self.nb, self.Hd), fp(lin2.bias)

    def fwd(self, eng, training):
        x, br = self.src, self.branch
        K.pool(x.data, None, eng.B, x.H * x.W, x.C, 1, 1.0 / (x.H * x.W), br.g, eng.workspace)
        self.use_mask = bool(training and self.p > 0.0)
        if self.use_mask:
            self.draws += 1
            K.dropout_mask(self.mask, self.p, eng.dropout_seed * 1000003 + self.draws * 2 + br.index,
                           eng.step_counter)
        W1, b1, W2, b2 = self._params(eng)
        K.binpred_fwd(br.g, W1, b1, W2, b2, self.mask if self.use_mask else None, self.p, self.mod.max_depth, br.h1,
                      br.widths, br.centers)

    def bwd(self, eng):
        x, br = self.src, self.branch
        W1, _, W2, _ = self._params(eng)
        lin1, lin2 = self.mod.predictor[0], self.mod.predictor[3]
        K.binpred_bwd(br.dcent, br.widths, br.h1, br.g, W1, W2, self.use_mask, self.p, self.mod.max_depth, self.dW2p,
                      self.db2p, self.dW1p, self.db1p, self.dg)
        fg = lambda p: eng._flat_slice(eng.flat_g, p)
        B = eng.B
        for part, prm in ((self.dW2p, lin2.weight), (self.db2p, lin2.bias), (self.dW1p, lin1.weight), (self.db1p, lin1.bias)):
            n = part.shape[1]
            K.channel_sum(part, B, n, n, fg(prm), eng.workspace)
        K.bcast_add(x.grad, self.dg, 1.0 / (x.H * x.W), accumulate=x.written)
        x.written = True


class FeatureLoss(Op):
    """Gradient of the feature-distillation term (utils_distillation_loss.py:72-98) into an encoder activation; sits
    right behind that activation's producer in the tape (see BinPredictor)."""

    def __init__(self, act, level, eng):
        self.src, self.out, self.level = act, act, level
        act.consumers.append(self)

    def fwd(self, eng, training):
        pass

    def bwd(self, eng):
        a = self.src
        extra = eng.feat_extra[self.level] if eng.feat_extra is not None else None
        if extra is not None:                   # autograd path: upstream gradient of the returned feature, NHWC f32
            K.bcast_add(a.grad.view(-1, 1, 1, a.C), extra.view(-1, a.C), 1.0, accumulate=a.written)
            a.written = True
        if eng.feat_coef is None:
            return
        assert a.written
        t = eng.branches['rgb'].feats[self.level]
        K.featcos_grad(a.data, t.data, eng.feat_stats[self.level], eng.feat_coef[self.level], a.grad)


class _Branch:
    pass


class AdaBinsEngine(DCEngine):
    """Runs AdaBinsDistillationModel through libadn (one engine per module instance)."""

    def __init__(self, module, compute_dtype=torch.bfloat16):
        super().__init__(module, None, compute_dtype, 'AdaBinsDistillationModel')
        self.dropout_seed = 0
        self.step_counter = None     # device f64[1] step count (the trainer's optimizer state): fresh dropout per replay
        self.feat_coef = None
        self.feat_stats = None
        self.feat_extra = None       # autograd path: upstream gradients of x1..x5 (see backward_leaves)
        self.autograd_pass = 0       # forward count of the student branch: a stale backward is refused
        self.branches = {}

    def bind_parameters(self):
        super().bind_parameters()
        # the teacher's parameters are a prefix of parameters(): offset of the first trainable (student) parameter
        first = next(iter(self.module.audio_encoder.parameters()))
        self.train_offset = self.offset[id(first)]

    # ------------------------------------------------------------------ build
    def _build_branch(self, name, index, Cin, enc, pred, dec, B, H, W, needs_grad):
        m = self.module
        br = _Branch()
        br.name, br.index, br.needs_grad = name, index, needs_grad
        br.inp = self.thin_input(name + '.in', Cin, H, W)
        ops_e, feats = enc.adn_ops(br.inp, name, H, W)
        ops = []
        for op in ops_e:                                # feature-loss / bin-predictor hooks right behind the producers
            ops.append(op)
            for lv, f in enumerate(feats):
                if getattr(op, 'out', None) is f and isinstance(op, ConvBNReLU):
                    if needs_grad:
                        ops.append(FeatureLoss(f, lv, self))
                    if lv == 4:
                        ops.append(BinPredictor(f, pred, br))
        ops_d, d1 = dec.adn_ops(feats, name)
        br.dec_bn_ops = [op for op in ops_d if isinstance(op, ConvBNReLU)]
        ops += ops_d
        br.logits = Act(name + '.logits', dec.n_bins, H, W)
        br.class_op = ConvLinear(d1, dec.class_head, br.logits)
        br.head = Head1x1(d1, m.residual_head, 2, 0.05 * m.max_depth)
        br.ops, br.feats, br.d1 = ops, feats, d1
        if not needs_grad:
            for op in ops + [br.class_op]:
                for a in list(getattr(op, 'srcs', [])) + [getattr(op, 'src', None), getattr(op, 'out', None)]:
                    if a is not None:
                        a.needs_grad = False
        return br

    def _prepare_branches(self, B, H, W, dev):
        if not self._bound():
            self.bind_parameters()
        key = (B, H, W, dev)
        if self._shape_enter(key):
            return
        m = self.module
        # output_size != input size: the reference resizes logits and residual with mode='nearest' before the per-pixel
        # softmax expectation / tanh / clamp (:196-198, 334-337, 383-386); per-pixel maps commute with a nearest resize, so
        # the forward outputs are computed at the input resolution and resized at the end (_outputs).  The fused TRAINING
        # step exists for output_size == input size only (what train_adabins_distillation.py uses); it raises otherwise.
        self.resize_to = m.output_size if (H != m.output_size or W != m.output_size) else None
        self.B, self.dev = B, dev
        self._scratch = {}
        self.epc = 8 if self.dtype == torch.bfloat16 else 4
        self.pairs = []
        self.branches = {
            'rgb': self._build_branch('rgb', 0, 3, m.rgb_encoder, m.rgb_bin_predictor, m.rgb_decoder, B, H, W, False),
            'audio': self._build_branch('audio', 1, 2, m.audio_encoder, m.audio_bin_predictor, m.audio_decoder, B, H, W,
                                        True),
        }
        f32 = dict(dtype=torch.float32, device=dev)
        ws = 1 << 16
        self.ops = []
        for br in self.branches.values():
            allops = br.ops + [br.class_op]
            acts = {}
            for op in allops:
                for a in list(getattr(op, 'srcs', [])) + [getattr(op, 'src', None), getattr(op, 'out', None)]:
                    if a is not None:
                        acts[id(a)] = a
            br.acts = list(acts.values())
            for a in br.acts:
                flag_solo(a)
                a.alloc(B, self.dtype, dev)
            mark_tail_writers(allops + [br.head])
            for op in allops + [br.head]:
                op.prepare(self)
                ws = max(ws, op.workspace_bytes(self))
            self.ops += allops                           # (weight packing walks self.ops)
            pix = B * H * W
            br.base = torch.empty(pix, **f32)
            br.final = torch.empty(B, 1, H, W, **f32)
            br.mean_logits = torch.empty(B, m.n_bins, **f32)
            ws = max(ws, K.pool_workspace_bytes(B, H * W, m.n_bins, 1), K.bins_bwd_workspace_bytes(B, H * W, m.n_bins))
            for f in br.feats:
                ws = max(ws, K.pool_workspace_bytes(B, f.H * f.W, f.C, 3))
        st = self.branches['audio']
        st.dbase, st.dres = torch.empty(B * H * W, **f32), torch.empty(B * H * W, **f32)
        st.dmean, st.dcent_extra = torch.empty(B, m.n_bins, **f32), torch.empty(B, m.n_bins, **f32)
        self.feat_stats_buf = [torch.empty(B, 3, f.C, **f32) for f in st.feats]
        self.pix_stats = torch.zeros(4, dtype=torch.float64, device=dev)
        self.terms = torch.zeros(8, **f32)
        self.workspace = torch.empty(ws // 4 + 4, **f32)
        self.weights_dirty = True
        self._shape_key = key

    # ------------------------------------------------------------------ forward of one branch
    def _forward_branch(self, br, x, training):
        if not x.is_cuda:
            raise RuntimeError('AdaBinsDistillationModel needs HIP device tensors (libadn has no CPU path)')
        x = x.contiguous().float()
        want = 3 if br.name == 'rgb' else 2
        if x.shape[1] != want:
            raise RuntimeError(f'expected input[{list(x.shape)}] to have {want} channels, but got {x.shape[1]} channels instead')
        self._prepare_branches(x.shape[0], x.shape[2], x.shape[3], x.device)
        if self.weights_dirty or self._packed_version != self._version_sum():
            self._pack_weights()
        K.nchw_slice_to_nhwc(x, 0, want, br.inp.data)
        for op in br.ops:
            op.fwd(self, training)
        if training:                                    # the reference's second decoder pass: same activations, but
            for op in br.dec_bn_ops:                    # every decoder BatchNorm updates its running stats once more
                op.update_running_stats_again(self)
        br.class_op.fwd(self, training)
        K.bins_fwd(br.logits.data, br.centers, br.base)
        br.head.fwd(self, training)

    def _finalize_plain(self, br):
        """final = clamp(base + residual) without loss terms (gt := base only feeds statistics nobody reads)."""
        K.distill_pix_stats(br.base, br.head.result, br.base, None, self.module.max_depth, br.final, self.pix_stats,
                            self.workspace)

    def run_branch(self, which, x, training):
        br_name = 'rgb' if which == 'rgb' else 'audio'
        with torch.no_grad():
            self._prepare_branches(x.shape[0], x.shape[2], x.shape[3], x.device)
            br = self.branches[br_name]
            self._forward_branch(br, x, training)
            self._finalize_plain(br)
            return self._outputs(br)

    def _outputs(self, br):
        B, m = self.B, self.module
        feats = {}
        for i, f in enumerate(br.feats):
            t = torch.empty(B, f.C, f.H, f.W, dtype=torch.float32, device=self.dev)
            K.nhwc_to_nchw(f.data, t)
            feats[f'x{i + 1}'] = t
        lg = br.logits
        logits = torch.empty(B, lg.C, lg.H, lg.W, dtype=torch.float32, device=self.dev)
        K.nhwc_to_nchw(lg.data, logits)
        shp = (B, 1, lg.H, lg.W)
        base, resid, final = br.base.view(shp).clone(), br.head.result.view(shp).clone(), br.final.clone()
        if self.resize_to is not None:
            logits, base, resid, final = [K.resize_nearest(t, self.resize_to) for t in (logits, base, resid, final)]
        return {'features': feats, 'bin_centers': br.centers.clone(), 'bin_widths': br.widths.clone(), 'bin_logits': logits,
                'base_depth': base, 'residual': resid, 'final_depth': final}

    # ------------------------------------------------------------------ backward of the student
    def backward_student(self, dbase, dres, dmean, dcent_extra, logits_extra=None):
        br = self.branches['audio']
        for a in br.acts:
            a.written = False
        self._final = set(id(p) for p, _, _ in self.param_meta if not p.requires_grad)
        self._wm = len(self.param_meta)
        br.head.bwd_head(self, dres)
        K.bins_bwd(br.logits.data, br.centers, br.base, dbase, dmean, br.logits.grad, br.dcent, self.workspace)
        if dcent_extra is not None:
            K.bcast_add(br.dcent.view(self.B, 1, 1, -1), dcent_extra, 1.0, accumulate=True)
        if logits_extra is not None:                     # upstream gradient of the returned bin_logits, NHWC f32
            lg = br.logits
            K.bcast_add(lg.grad.view(-1, 1, 1, lg.C), logits_extra.view(-1, lg.C), 1.0, accumulate=True)
        br.class_op.bwd(self)
        for op in reversed(br.ops):
            if op.out.needs_grad:
                op.bwd(self)


def _f32_add_(dst, src):
    """dst += src for flat f32 device tensors (adn_bcast_add with one pixel per row)."""
    K.bcast_add(dst.view(-1, 1, 1, 1), src.contiguous().view(-1, 1), 1.0, accumulate=True)


def _backward_leaves(eng, g_feats, g_centers, g_logits, g_base, g_res, g_final):
    """Backward of the student branch from upstream gradients of the tensors forward() returned (any may be None):
    g_feats 5 x [B,C,h,w], g_centers [B,nb], g_logits [B,nb,H,W], g_base / g_res / g_final [B,1,H,W], all f32 NCHW.
    final = clamp(base + residual, 0, max_depth) (adabins_distillation_model.py:389-391) routes g_final into both."""
    br, m = eng.branches['audio'], eng.module
    B, H, W = eng.B, br.logits.H, br.logits.W
    f32 = dict(dtype=torch.float32, device=eng.dev)
    dbase = torch.zeros(B * H * W, **f32)
    dres = torch.zeros(B * H * W, **f32)
    if g_final is not None:
        s = br.base.clone()
        _f32_add_(s, br.head.result)
        masked = torch.empty_like(s)
        K.clamp_range(s, m.max_depth, masked, g=g_final.contiguous().float().view(-1))
        _f32_add_(dbase, masked)
        _f32_add_(dres, masked)
    if g_base is not None:
        _f32_add_(dbase, g_base.float())
    if g_res is not None:
        _f32_add_(dres, g_res.float())
    eng.feat_extra = None
    if any(g is not None for g in g_feats):
        eng.feat_extra = []
        for g, a in zip(g_feats, br.feats):
            if g is None:
                eng.feat_extra.append(None)
                continue
            t = torch.empty(B, a.H, a.W, a.C, **f32)
            K.nchw_to_nhwc(g.contiguous().float(), t)
            eng.feat_extra.append(t)
    saved = eng.feat_coef, eng.feat_stats
    eng.feat_coef, eng.feat_stats = None, None
    glog = None
    if g_logits is not None:
        glog = torch.empty(B, H, W, m.n_bins, **f32)
        K.nchw_to_nhwc(g_logits.contiguous().float(), glog)
    br.dmean.zero_()
    try:
        eng.backward_student(dbase, dres, br.dmean, g_centers.contiguous().float() if g_centers is not None else None,
                             logits_extra=glog)
    finally:
        eng.feat_extra = None
        eng.feat_coef, eng.feat_stats = saved


class _StudentFunction(torch.autograd.Function):
    """torch.autograd bridge of the student branch: the parameters are inputs, the outputs are the leaves of the
    reference's output dict (x1..x5, bin_centers, bin_widths, bin_logits, base_depth, residual, final_depth), so the
    reference's own loop -- ``loss, _ = criterion(model(audio, rgb), gt, mask); loss.backward(); clip; optimizer.step()``
    (train_adabins_distillation.py:445-456) -- runs unchanged.  The fused AdaBinsTrainer stays the fast path."""

    @staticmethod
    def forward(ctx, audio, engine, *params):
        eng = engine
        eng._prepare_branches(audio.shape[0], audio.shape[2], audio.shape[3], audio.device)
        br = eng.branches['audio']
        eng._forward_branch(br, audio, True)
        eng._finalize_plain(br)
        eng.autograd_pass += 1
        ctx.engine, ctx.stamp = eng, eng.autograd_pass
        ctx.set_materialize_grads(False)
        o = eng._outputs(br)
        leaves = tuple(o['features'][f'x{i}'] for i in range(1, 6)) + (o['bin_centers'], o['bin_widths'], o['bin_logits'],
                                                                       o['base_depth'], o['residual'], o['final_depth'])
        ctx.mark_non_differentiable(o['bin_widths'])      # the centres carry the predictor's gradient (:143-149)
        return leaves

    @staticmethod
    def backward(ctx, *g):
        eng = ctx.engine
        if ctx.stamp != eng.autograd_pass:
            raise RuntimeError('AdaBinsDistillationModel: backward through a forward whose activations were overwritten '
                               'by a later training forward of the same module')
        g_logits, g_base, g_res, g_final = g[7], g[8], g[9], g[10]
        if eng.resize_to is not None:          # the returned maps are nearest-resized: gradients back through the gather
            H, W = eng.branches['audio'].logits.H, eng.branches['audio'].logits.W
            back = lambda t: None if t is None else K.resize_nearest_bwd(t.contiguous().float(), H, W)
            g_logits, g_base, g_res, g_final = back(g_logits), back(g_base), back(g_res), back(g_final)
        _backward_leaves(eng, g[0:5], g[5], g_logits, g_base, g_res, g_final)
        off = eng.train_offset
        return (None, None) + tuple(eng.grad_view(p) if (o >= off and p.requires_grad) else None
                                    for p, o, _ in eng.param_meta)


def run_student(engine, audio, training):
    """forward_audio of the module: differentiable outputs in training mode under grad, plain values otherwise."""
    if not engine._bound():
        engine.bind_parameters()
    off = engine.train_offset
    if training and torch.is_grad_enabled() and any(p.requires_grad for p, o, _ in engine.param_meta if o >= off):
        v = _StudentFunction.apply(audio, engine, *[p for p, _, _ in engine.param_meta])
        feats = {f'x{i + 1}': v[i] for i in range(5)}
        return {'features': feats, 'bin_centers': v[5], 'bin_widths': v[6], 'bin_logits': v[7], 'base_depth': v[8],
                'residual': v[9], 'final_depth': v[10]}
    return engine.run_branch('audio', audio, training)


def _update_again(self, eng):
    """Second running-statistics update of a decoder BatchNorm with the same batch statistics."""
    bn, o = self.bn, self.out
    gamma, beta, rmean, rvar, _ = self._bn_vectors()
    if rmean is None:
        return
    K.bn_fwd_finalize(self.part, self.P, self.N, eng.B * o.H * o.W, gamma, beta, bn.eps,
                      0.1 if bn.momentum is None else bn.momentum, rmean, rvar, bn.num_batches_tracked, o.mean, o.istd,
                      self.scale, self.shift, scratch=eng.bn_scratch)


ConvBNReLU.update_running_stats_again = _update_again


class _GradView:
    """What ddp.GradientAllReducer.attach binds to: a flat gradient range and the grad-ready hook slot."""

    def __init__(self, flat_g):
        self.flat_g = flat_g
        self.on_grad_ready = None


class AdaBinsTrainer(GraphedStep):
    """One fused distillation step: teacher forward, student forward, DistillationLoss, student backward,
    clip_grad_norm_(1.0), AdamW -- train_adabins_distillation.py:445-456 with the loss weights of :179-188."""

    def __init__(self, engine, lambda_task=1.0, lambda_response=0.5, lambda_feature=0.3, lambda_bin=0.2,
                 lambda_sparse=0.1, temperature=4.0, optimizer='AdamW', lr=1e-4, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=None, clip_norm=1.0, ddp=None):
        self.engine = engine
        self.ddp = ddp                    # ddp.GradientAllReducer: one process per GPU, DataParallel semantics
        self.lambdas = (lambda_task, lambda_response, lambda_feature, lambda_bin, lambda_sparse)
        self.temperature = temperature
        self.opt_kind = {'AdamW': 0, 'Adam': 1, 'SGD': 2}[optimizer]
        self.lr, self.betas, self.eps = float(lr), betas, float(eps)
        self.weight_decay = float((0.01 if optimizer == 'AdamW' else 0.0) if weight_decay is None else weight_decay)
        self.clip_norm = clip_norm
        self._ready = False

    @classmethod
    def from_criterion(cls, engine, criterion, **kw):
        """Build from a utils_distillation_loss.DistillationLoss / AdaptiveDistillationLoss instance."""
        t = cls(engine, **kw)
        t.set_criterion(criterion)
        return t

    def set_criterion(self, criterion):
        """Take the loss weights / temperature of ``criterion`` (call again after AdaptiveDistillationLoss.set_epoch)."""
        c = criterion.criterion() if hasattr(criterion, 'get_adaptive_weights') else criterion
        self.lambdas, self.temperature = c.weights(), c.temperature

    def _setup(self, dev):
        eng = self.engine
        f64 = dict(dtype=torch.float64, device=dev)
        self.state = torch.zeros(8, **f64)
        self.norm_ws = torch.empty(1024 + 8, **f64)
        self.exp_avg = torch.zeros_like(eng.flat_p)
        self.exp_avg_sq = torch.zeros_like(eng.flat_p)
        eng.step_counter = self.state
        self.bucket_norm = None
        if self.ddp is not None:
            # only the student's gradients (the suffix of the flat buffer) are exchanged: the teacher never trains
            off = eng.train_offset
            self._ddp_view = _GradView(eng.flat_g[off:])
            self.ddp.attach(self._ddp_view)
            eng.on_grad_ready = lambda lo: self._ddp_view.on_grad_ready(max(0, lo - off))
            if self.clip_norm is not None and eng.flat_g.is_cuda:
                self.bucket_norm = self.ddp.enable_bucket_norm()
        self._ready = True

    def _mean_logits_resized(self, br, S):
        """Spatial mean of the nearest-resized bin logits (the KL term's input when output_size != input size)."""
        eng, lg = self.engine, br.logits
        B = eng.B
        f32 = dict(dtype=torch.float32, device=lg.data.device)
        t = torch.empty(B, lg.C, lg.H, lg.W, **f32)
        K.nhwc_to_nchw(lg.data, t)
        r = K.resize_nearest(t, S)
        n = torch.empty(B, S, S, lg.C, **f32)
        K.nchw_to_nhwc(r, n)
        wsp = torch.empty(max(K.pool_workspace_bytes(B, S * S, lg.C, 1), 16) // 4, **f32)
        K.pool(n, None, B, S * S, lg.C, 1, 1.0 / (S * S), br.mean_logits, wsp)

    def enable_graph(self, after_steps=3):
        if self.ddp is not None:
            raise RuntimeError('the hipGraph step is not combined with the data-parallel reducer (host-side collectives)')
        super().enable_graph(after_steps)

    def _optim_meta(self):
        """Parameters the reference's optimizer holds: filter(requires_grad, model.parameters())
        (train_adabins_distillation.py:371-386); the teacher's (never given a gradient) carry no state."""
        return [(p, off, n) for p, off, n in self.engine.param_meta if p.requires_grad]

    def state_dict(self):
        """'optimizer_state_dict' in torch.optim format (optim_state.py)."""
        from . import optim_state
        eng = self.engine
        if not eng._bound():
            eng.bind_parameters()
        step = int(self.state[0].item()) if self._ready else 0
        sd = optim_state.export_state(self._optim_meta(), eng._view, self.exp_avg if self._ready else None,
                                      self.exp_avg_sq if self._ready else None, step, self.opt_kind, self.lr, self.betas,
                                      self.eps, self.weight_decay)
        for i, (p, off, _) in enumerate(self._optim_meta()):          # teacher parameters: no gradient, no state
            if off < eng.train_offset:
                sd['state'].pop(i, None)
        return sd

    def load_state_dict(self, sd, device):
        from . import optim_state
        if not self.engine._bound():
            self.engine.bind_parameters()
        self._setup(device)
        if optim_state.is_torch_format(sd):
            step, group = optim_state.import_state(sd, self._optim_meta(), self.engine._view, self.exp_avg, self.exp_avg_sq)
            optim_state.adopt_group(self, group)
            self.state[0] = float(step)
        elif 'exp_avg' in sd:
            step = optim_state.import_legacy_flat(sd, self.engine.param_meta, self.exp_avg, self.exp_avg_sq)
            self.state[0] = float(step)

    def step(self, audio, rgb, gt):
        """audio [B,2,H,W], rgb [B,3,H,W] or None, gt [B,1,H,W] -> (total loss 0-dim device tensor, terms f32[8]).
        With ``enable_graph()`` the step replays as one hipGraph (the dropout draw mixes in the device-side step count)."""
        return self._graphed(audio, rgb, gt)

    def _step_impl(self, audio, rgb, gt):
        eng = self.engine
        m = eng.module
        eng._prepare_branches(audio.shape[0], audio.shape[2], audio.shape[3], audio.device)
        if not self._ready:
            self._setup(audio.device)
        st, te = eng.branches['audio'], eng.branches['rgb']
        gt = gt.contiguous().float()
        has_t = rgb is not None
        if has_t:
            eng._forward_branch(te, rgb, True)
            eng._finalize_plain(te)
        eng._forward_branch(st, audio, True)
        lt, lr_, lf, lb, ls = self.lambdas
        # output_size != input size: the reference resizes logits and residual with mode='nearest' before the per-pixel maps
        # (:196-198, 334-337, 383-386).  Those maps commute with a nearest resize, so the outputs at S x S are gathers of the
        # input-resolution ones: the pixel terms are evaluated on the gathered base / residual against the S x S target, the
        # mean logits over the gathered logits, and the gradients come back through the gather's transpose
        # (adn_resize_nearest_bwd) before the student's backward.  Plain tensors per step: the rare path is not tuned.
        S = eng.resize_to
        B, H, W = eng.B, st.logits.H, st.logits.W
        f32 = dict(dtype=torch.float32, device=gt.device)
        if S is None:
            base_o, res_o, tfin_o, final_o = st.base, st.head.result, (te.final if has_t else None), st.final
        else:
            shp = (B, 1, H, W)
            base_o = K.resize_nearest(st.base.view(shp), S).view(-1)
            res_o = K.resize_nearest(st.head.result.view(shp), S).view(-1)
            tfin_o = K.resize_nearest(te.final, S).view(-1) if has_t else None
            final_o = torch.empty(B, 1, S, S, **f32)
            if gt.shape[-2:] != (S, S):
                raise RuntimeError(f'AdaBinsTrainer: gt is {tuple(gt.shape)}, the model output is {S}x{S}')
        K.distill_pix_stats(base_o, res_o, gt, tfin_o, m.max_depth, final_o, eng.pix_stats, eng.workspace)
        world = 1
        if self.ddp is not None:
            # DataParallel computes ONE loss on the gathered outputs (adabins_distillation_model.py:493-496): the pixel
            # terms normalise by the global valid count (sum the statistics), the per-sample means become means over
            # world x B samples (their weights / world here, gradients SUM-reduced below)
            world = self.ddp.world_size
            self.ddp.all_reduce_loss_stats(eng.pix_stats)
        HW = H * W
        if S is None:
            K.pool(st.logits.data, None, B, HW, m.n_bins, 1, 1.0 / HW, st.mean_logits, eng.workspace)
        else:
            self._mean_logits_resized(st, S)
        if has_t:
            if S is None:
                K.pool(te.logits.data, None, B, HW, m.n_bins, 1, 1.0 / HW, te.mean_logits, eng.workspace)
            else:
                self._mean_logits_resized(te, S)
            for i, (a, r) in enumerate(zip(st.feats, te.feats)):
                K.pool(a.data, r.data, B, a.H * a.W, a.C, 3, 1.0, eng.feat_stats_buf[i], eng.workspace)
            eng.feat_stats = eng.feat_stats_buf
            eng.feat_coef = [-lf / (5.0 * B * a.C * world) for a in st.feats]
        else:
            eng.feat_stats, eng.feat_coef = None, None
        K.distill_small(st.mean_logits, te.mean_logits if has_t else None, st.centers, te.centers if has_t else None,
                        eng.feat_stats_buf, [a.C for a in st.feats], eng.pix_stats, self.temperature,
                        (lt, lr_, lf / world, lb / world, ls), eng.terms, st.dmean, st.dcent_extra)
        if self.ddp is not None:
            # report the global terms: feature / KL / centre terms are means of the per-rank means
            self.ddp.all_reduce_loss_stats(eng.terms[2:5])
            eng.terms[2:5] /= world
            eng.terms[6] = (lt * eng.terms[0] + lr_ * eng.terms[1] + lf * eng.terms[2] + lb * (eng.terms[3] + eng.terms[4])
                            + ls * eng.terms[5])
            self.ddp.begin_backward()
        if S is None:
            K.distill_pix_grad(st.base, st.head.result, gt, tfin_o, m.max_depth, eng.pix_stats, lt,
                               lr_ if has_t else 0.0, ls, st.dbase, st.dres)
            eng.backward_student(st.dbase, st.dres, st.dmean, st.dcent_extra)
        else:
            db_o, dr_o = torch.empty(B * S * S, **f32), torch.empty(B * S * S, **f32)
            K.distill_pix_grad(base_o, res_o, gt, tfin_o, m.max_depth, eng.pix_stats, lt, lr_ if has_t else 0.0, ls, db_o,
                               dr_o)
            dbase = K.resize_nearest_bwd(db_o.view(B, 1, S, S), H, W).view(-1)
            dres = K.resize_nearest_bwd(dr_o.view(B, 1, S, S), H, W).view(-1)
            # d KL / d logits: dmean / S^2 on every output pixel, back through the gather
            gl = torch.empty(B, S, S, m.n_bins, **f32)
            K.bcast_add(gl, st.dmean, 1.0 / (S * S), accumulate=False)
            gn = torch.empty(B, m.n_bins, S, S, **f32)
            K.nhwc_to_nchw(gl, gn)
            gh = K.resize_nearest_bwd(gn, H, W)
            glog = torch.empty(B, H, W, m.n_bins, **f32)
            K.nchw_to_nhwc(gh, glog)
            st.dmean.zero_()
            eng.backward_student(dbase, dres, st.dmean, st.dcent_extra, logits_extra=glog)
        if self.ddp is not None:
            self.ddp.finish()
        off = eng.train_offset
        p, g = eng.flat_p[off:], eng.flat_g[off:]
        if self.clip_norm is not None and self.bucket_norm is not None:
            K.grad_norm_ranges(g, None, self.bucket_norm, float(self.clip_norm), self.state, self.norm_ws)
        elif self.clip_norm is not None:
            K.grad_norm(g, float(self.clip_norm), self.state, self.norm_ws)
        K.optimizer_step(p, g, self.exp_avg[off:], self.exp_avg_sq[off:], self.opt_kind, self.lr, self.betas[0],
                         self.betas[1], self.eps, self.weight_decay, self.clip_norm is not None, self.state,
                         bf16_copy=eng.flat_w16[off:] if eng.flat_w16 is not None else None)
        eng.weights_dirty = True
        eng.s2_fresh = False            # the teacher half of the bf16 mirror is still valid, but keep the re-cast simple
        return eng.terms[6], eng.terms
