"""DoubleConv U-Net family on libadn kernels: an op tape over NHWC activations.

Replaces what PyTorch dispatches for ``model(x)`` / ``loss.backward()`` of the reference's DoubleConv-based
networks (/root/reference/models/rgb_depth_model.py:80-218, binaural_attention_model.py:155-340): a model
mirror describes its forward as a list of ops over named activation records; the engine runs the list forward
and in reverse for the backward pass.  Every op is a handful of libadn launches (C ABI, include/adn.h):

  ConvBNReLU   (Conv2d k3/k1 -> BatchNorm2d -> ReLU)   S1 implicit GEMM with the Z+stats epilogue, BN finalize,
               one BN+ReLU materialisation pass; backward = [ReLU+BN stats pass |fused into the consumer's dgrad
               epilogue] -> finalize -> apply -> wgrad -> dgrad (S1 GEMM with the flipped/transposed operand)
  MaxPool2 / Upsample2x / Head1x1        memory-bound kernels of csrc/dcnet.hip

Data layout in HBM: activations NHWC in the compute dtype (bf16 throughput path, f32 exact path); per
ConvBNReLU the raw conv output z and the activated y (y feeds the next GEMM through LDS-DMA, z is needed by the
BN backward); one gradient buffer per activation, written by its consumers in backward order (first writer
overwrites, later writers accumulate); skip concats are virtual (two base pointers into the consumer GEMM).
Parameters/gradients: flat f32 buffers (flat.py); gradients become final from the END of the flat buffer
towards its start because backward visits the ops in reverse parameters() order.
"""
from __future__ import annotations

import os

import torch

from . import _lib
from . import kernels as K
from .flat import FlatParamEngine
from ._lib import EPI_ACT, EPI_ADD, EPI_BWD, EPI_Z_STATS, GEMM_S1

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


class Act:
    """One NHWC activation record: data (+ raw conv output z and BN statistics when produced by ConvBNReLU)."""

    def __init__(self, name, C, H, W, needs_grad=True):
        self.name, self.C, self.H, self.W = name, C, H, W
        self.needs_grad = needs_grad
        self.data = None
        self.grad = None
        self.producer = None
        self.consumers = []
        self.fused_bwd = False       # sole consumer is a conv: its dgrad epilogue applies the ReLU mask + BN stats
        self.written = False         # a consumer already wrote .grad in the current backward pass
        self.z = self.mean = self.istd = self.bpart = None
        self.bpart_rows = 0
        self.alias_of = None         # this record's gradient IS that record's gradient buffer (gated residual)
        self.pair_data = self.pair_grad = None      # [2B,...] buffers when stacked with a sibling ([left; right])
        self.q8 = self.q8s = None    # MX-fp8 copy of .data (e4m3 bytes + E8M0 scales) for the fp8 conv path
        self.q8_serial = -1          # forward pass the copy belongs to
        self.want_q8 = False         # some consumer is an MX-fp8 conv: the producer writes the copy with its output
        self.tail_writer = None      # the op that writes this record's gradient LAST in the backward pass, when it is a max-pool /
        self.tail_fused = False      # upsample: that kernel then applies the ReLU mask + BN-backward sums (no relu_bwd_stats pass)

    def target(self):
        return self.alias_of if self.alias_of is not None else self

    def alloc(self, B, dtype, dev):
        if self.data is None:
            self.data = torch.empty(B, self.H, self.W, self.C, dtype=dtype, device=dev)
        if self.needs_grad and self.grad is None:
            self.grad = self.alias_of.grad if self.alias_of is not None else torch.empty_like(self.data)


def stack_pair(a, b, B, dtype, dev):
    """Allocate two same-shaped records as the halves of one [2B,H,W,C] buffer (data and gradient), so that ops
    over [left; right] run as ONE launch with batch 2B."""
    assert (a.C, a.H, a.W) == (b.C, b.H, b.W)
    data = torch.empty(2 * B, a.H, a.W, a.C, dtype=dtype, device=dev)
    a.data, b.data = data[:B], data[B:]
    a.pair_data = b.pair_data = data
    if a.alias_of is not None:
        grad = a.alias_of.pair_grad
    else:
        grad = torch.empty_like(data)
    a.grad, b.grad = grad[:B], grad[B:]
    a.pair_grad = b.pair_grad = grad


class Op:
    params = ()

    def prepare(self, eng):
        pass

    def workspace_bytes(self, eng):
        return 0


class ConvBNReLU(Op):
    """Conv2d(k in {1,3}, padding k//2) [+bias] -> BatchNorm2d -> ReLU over the virtual concat of ``srcs``."""

    def __init__(self, srcs, conv, bn, out):
        self.srcs, self.conv, self.bn, self.out = list(srcs), conv, bn, out
        self.ks = conv.kernel_size[0]
        assert conv.kernel_size in ((1, 1), (3, 3)) and conv.stride == (1, 1) and conv.padding == (self.ks // 2,) * 2
        assert len(self.srcs) in (1, 2)
        out.producer = self
        for s in self.srcs:
            s.consumers.append(self)

    def prepare(self, eng):
        T, dev, B = eng.dtype, eng.dev, eng.B
        o = self.out
        N = o.C                                       # as the kernels see it (a solo mid tensor may be zero padded)
        self.N = N
        self.n_real = getattr(o, 'C_real', N)
        self.padded = self.n_real != N
        self.cin_real = sum(s.C_real if hasattr(s, 'C_real') else s.C for s in self.srcs)
        self.c0 = self.srcs[0].C
        self.c1 = self.srcs[1].C if len(self.srcs) > 1 else 0
        cin = self.c0 + self.c1                       # as the kernels see it (thin / padded inputs carry zero channels)
        assert all(not hasattr(s, 'C_real') for s in self.srcs[:-1]), 'a padded source must be the last one'
        taps = self.ks * self.ks
        w = self.conv.weight
        assert w.shape[1] == self.cin_real and w.shape[0] == self.n_real
        f32 = dict(dtype=torch.float32, device=dev)
        # forward operand [N][row stride]: the parameter memory itself when rows need no padding
        rs = K.s1_row_stride(T, taps, cin)
        self.fwd_is_view = (rs == taps * self.cin_real) and not self.padded
        if self.fwd_is_view:
            src = eng.flat_p if T == torch.float32 else eng.flat_w16
            self.w_fwd = eng._flat_slice(src, w).view(N, rs)
        else:
            self.w_fwd = torch.zeros(N, rs, dtype=T, device=dev)
        # input-gradient operand [Cin][row stride'] (flipped taps, transposed channels); rows of padded inputs stay 0
        self.need_dgrad = any(s.needs_grad for s in self.srcs)
        self.w_dg = torch.zeros(cin, K.s1_row_stride(T, taps, N), dtype=T, device=dev) if self.need_dgrad else None
        if self.padded:
            # Output channels zero-padded to a multiple of 64 so that the layer runs on the MFMA kernels: padded f32
            # master [N][taps][cin_real] (extra rows 0), padded BN vectors (gamma = beta = 0 -> the extra channels are
            # exactly 0 forward and backward), running statistics re-homed into padded buffers of which the module's
            # buffers become views, temporaries for the gradients.
            bn = self.bn
            self.wm_p = torch.zeros(N, taps * self.cin_real, **f32)
            self.gamma_p, self.beta_p = torch.zeros(N, **f32), torch.zeros(N, **f32)
            self.bias_p = torch.zeros(N, **f32) if self.conv.bias is not None else None
            self.dgamma_p, self.dbeta_p = torch.empty(N, **f32), torch.empty(N, **f32)
            self.dw_p = torch.empty(N, taps * self.cin_real, **f32) if o.needs_grad else None
            if bn.track_running_stats and bn.running_mean is not None:
                rm, rv = torch.zeros(N, **f32), torch.ones(N, **f32)
                rm[:self.n_real].copy_(bn.running_mean)
                rv[:self.n_real].copy_(bn.running_var)
                bn.running_mean, bn.running_var = rm[:self.n_real], rv[:self.n_real]
                self.rm_p, self.rv_p = rm, rv
            else:
                self.rm_p = self.rv_p = None
        o.z = torch.empty_like(o.data)
        o.mean, o.istd = torch.empty(N, **f32), torch.empty(N, **f32)
        self.scale, self.shift = torch.empty(N, **f32), torch.empty(N, **f32)
        self.coef = torch.empty(2 * N, **f32)
        H, W = o.H, o.W
        # MX-fp8 variant of this layer (engine in fp8 mode): 3 x 3, every channel count a multiple of 64, image tileable
        # by 8 x 16 pixels; other layers (the thin first conv, 1 x 1 convs) stay on the bf16 kernels
        self.mx8 = bool(eng.mx8 and self.ks == 3 and not self.padded and self.cin_real == cin and self.c0 % 64 == 0 and
                        self.c1 % 64 == 0 and N % 64 == 0 and H % 8 == 0 and W % 16 == 0 and
                        o.pair_data is None and all(s_.pair_data is None for s_ in self.srcs) and
                        o.data.shape[0] == B and all(s_.data.shape[0] == B for s_ in self.srcs))
        u8 = dict(dtype=torch.uint8, device=dev)
        if self.mx8:
            s8, ssc = K.mx8_pack_shapes(N, cin, False)
            self.w8_fwd, self.wsc_fwd = torch.empty(s8, **u8), torch.empty(ssc, **u8)
            if self.need_dgrad:
                s8, ssc = K.mx8_pack_shapes(N, cin, True)
                self.w8_dg, self.wsc_dg = torch.empty(s8, **u8), torch.empty(ssc, **u8)
                self.g8, self.g8s = torch.empty(B, H, W, N, **u8), torch.empty(B, H, W, N // 32, **u8)
            for s_ in self.srcs:
                s_.want_q8 = True
                if s_.q8 is None:
                    s_.q8, s_.q8s = torch.empty(B, s_.H, s_.W, s_.C, **u8), torch.empty(B, s_.H, s_.W, s_.C // 32, **u8)
        self.P, ws1 = K.igemm_query(T, GEMM_S1, B, H, W, self.c0, self.c1, N, [N], ks=self.ks)
        if self.mx8:
            self.P, ws1 = K.conv3x3_mx8_num_partials(B, H, W, N, self.c0, self.c1), 0
        self.part = torch.empty(self.P * 2 * N, **f32)
        ws2 = 0
        if self.need_dgrad:
            segc = [self.c0, self.c1] if self.c1 else [self.c0]
            pg, ws2 = K.igemm_query(T, GEMM_S1, B, H, W, N, 0, cin, segc, ks=self.ks)
            if self.mx8:
                pg, ws2 = K.conv3x3_mx8_num_partials(B, H, W, cin, N, 0), 0
            for s in self.srcs:
                if s.fused_bwd:
                    s.bpart_rows = pg
                    s.bpart = torch.empty(pg * 2 * s.C, **f32)
        if not o.fused_bwd and o.needs_grad:
            o.bpart_rows = K.relu_bwd_stats_num_partials(B * H * W, N)
            o.bpart = torch.empty(o.bpart_rows * 2 * N, **f32)
        cv = self.cin_real if self.cin_real != cin else 0
        ws3 = K.wgrad_workspace_bytes(T, B, H, W, N, 0, self.c0, self.c1, c_valid=cv, ks=self.ks)
        self.c_valid = cv
        self._ws = max(ws1, ws2, ws3)
        # shared scratch of the BatchNorm partial-row pre-reduction (kernels._bn_prereduce): 64 rows x 2 x the widest layer
        need = K.BN_REDUCE_SLICES * 2 * N
        if getattr(eng, 'bn_scratch', None) is None or eng.bn_scratch.numel() < need or eng.bn_scratch.device != dev:
            eng.bn_scratch = torch.empty(max(need, K.BN_REDUCE_SLICES * 2 * 1024), **f32)

    def workspace_bytes(self, eng):
        return self._ws

    def pack(self, eng):
        w = self.conv.weight
        taps = self.ks * self.ks
        master = eng._flat_slice(eng.flat_p, w)
        if self.padded:
            nr = self.n_real
            K.pack_rows(master, nr, 1, taps * self.cin_real, self.wm_p[:nr])          # rows nr..N stay zero
            K.pack_rows(eng._flat_slice(eng.flat_p, self.bn.weight), nr, 1, 1, self.gamma_p[:nr])
            K.pack_rows(eng._flat_slice(eng.flat_p, self.bn.bias), nr, 1, 1, self.beta_p[:nr])
            if self.bias_p is not None:
                K.pack_rows(eng._flat_slice(eng.flat_p, self.conv.bias), nr, 1, 1, self.bias_p[:nr])
            master = self.wm_p
        if self.mx8:
            K.mx8_pack(master, self.N, self.cin_real, False, self.w8_fwd, self.wsc_fwd)
            if self.need_dgrad:
                K.mx8_pack(master, self.N, self.cin_real, True, self.w8_dg, self.wsc_dg)
            return
        if not self.fwd_is_view:
            K.pack_rows(master, self.N, taps, self.cin_real, self.w_fwd, y_pad=self.c0 + self.c1)
        if self.w_dg is not None:
            K.pack_transpose_taps(master, self.N, taps, self.cin_real, self.w_dg, flip=True)

    def _bn_vectors(self):
        """(gamma, beta, running_mean, running_var, conv bias) as the kernels see them (padded copies when padded)."""
        bn = self.bn
        if self.padded:
            return self.gamma_p, self.beta_p, self.rm_p, self.rv_p, self.bias_p
        track = bn.track_running_stats and bn.running_mean is not None
        return bn.weight, bn.bias, bn.running_mean if track else None, bn.running_var if track else None, self.conv.bias

    def fwd(self, eng, training):
        T, B, o = eng.dtype, eng.B, self.out
        in0 = self.srcs[0].data
        in1 = self.srcs[1].data if self.c1 else None
        bn, N = self.bn, self.N
        pixels = B * o.H * o.W
        gamma, beta, rmean, rvar, cbias = self._bn_vectors()
        if self.mx8:
            q0, s0 = eng.q8_of(self.srcs[0])
            q1, s1 = eng.q8_of(self.srcs[1]) if self.c1 else (None, None)
        if training:
            if self.mx8:
                K.conv3x3_mx8(B, o.H, o.W, q0, s0, q1, s1, self.w8_fwd, self.wsc_fwd, N, EPI_Z_STATS,
                              [K.Seg(N, out0=o.z, partials=self.part, bias=cbias)])
            else:
                K.igemm(T, GEMM_S1, B, o.H, o.W, in0, in1, self.w_fwd, N, EPI_Z_STATS,
                        [K.Seg(N, out0=o.z, partials=self.part, bias=cbias)], eng.workspace,
                        algo_c=self.cin_real, ks=self.ks)
            K.bn_fwd_finalize(self.part, self.P, N, pixels, gamma, beta, bn.eps,
                              BN_MOMENTUM if bn.momentum is None else bn.momentum, rmean, rvar,
                              bn.num_batches_tracked if rmean is not None else None, o.mean, o.istd, self.scale,
                              self.shift, scratch=eng.bn_scratch)
            if o.want_q8 and N % 32 == 0:             # a consumer is an fp8 conv: its operand copy comes with the output
                K.bn_act_mx8(o.z, pixels, N, self.scale, self.shift, o.data, o.q8, o.q8s)
                o.q8_serial = eng.fwd_serial
            else:
                K.bn_act(o.z, pixels, N, self.scale, self.shift, 0.0, None, o.data)
        elif self.mx8:
            K.bn_eval_affine(gamma, beta, rmean, rvar, bn.eps, self.scale, self.shift)
            if cbias is not None:
                self.shift.addcmul_(cbias.detach(), self.scale)
            K.conv3x3_mx8(B, o.H, o.W, q0, s0, q1, s1, self.w8_fwd, self.wsc_fwd, N, EPI_ACT,
                          [K.Seg(N, out1=o.data, scale=self.scale, shift=self.shift)])
        else:
            K.bn_eval_affine(gamma, beta, rmean, rvar, bn.eps, self.scale, self.shift)
            if cbias is not None:                             # BN(conv + b) = conv * scale + (shift + b * scale)
                self.shift.addcmul_(cbias.detach(), self.scale)
            K.igemm(T, GEMM_S1, B, o.H, o.W, in0, in1, self.w_fwd, N, EPI_ACT,
                    [K.Seg(N, out1=o.data, scale=self.scale, shift=self.shift)], eng.workspace,
                    algo_c=self.cin_real, ks=self.ks)

    def bwd(self, eng):
        T, B, o = eng.dtype, eng.B, self.out
        N, bn = self.N, self.bn
        pixels = B * o.H * o.W
        G = o.grad
        if not o.fused_bwd and not o.tail_fused:
            K.relu_bwd_stats(G, o.data, o.z, o.mean, o.istd, pixels, N, o.bpart)
        fg = lambda p: eng._flat_slice(eng.flat_g, p)
        if self.padded:
            K.bn_bwd_finalize(o.bpart, o.bpart_rows, N, pixels, self.dgamma_p, self.dbeta_p, self.coef, scratch=eng.bn_scratch)
            K.pack_rows(self.dgamma_p[:self.n_real], self.n_real, 1, 1, fg(bn.weight))
            K.pack_rows(self.dbeta_p[:self.n_real], self.n_real, 1, 1, fg(bn.bias))
        else:
            K.bn_bwd_finalize(o.bpart, o.bpart_rows, N, pixels, fg(bn.weight), fg(bn.bias), self.coef, scratch=eng.bn_scratch)
        if self.mx8 and self.need_dgrad:                 # G is now d loss / d z (+ its fp8 copy for the input-gradient GEMM)
            K.bn_bwd_apply_mx8(G, o.z, pixels, N, self.scale, o.mean, o.istd, self.coef, self.g8, self.g8s)
        else:
            K.bn_bwd_apply(G, o.z, pixels, N, self.scale, o.mean, o.istd, self.coef)      # G is now d loss / d z
        eng._mark(bn.weight, bn.bias)
        if self.conv.bias is not None:
            # a bias in front of BatchNorm has an identically zero gradient (BN subtracts the batch mean); the
            # reference accumulates float noise here, we write the exact value
            eng._flat_slice(eng.flat_g, self.conv.bias).zero_()
            eng._mark(self.conv.bias)
        in0 = self.srcs[0].data
        in1 = self.srcs[1].data if self.c1 else None
        if self.padded:
            K.wgrad(T, B, o.H, o.W, G, None, in0, in1, self.dw_p, eng.workspace, c_valid=self.c_valid, ks=self.ks)
            nr, row = self.n_real, self.dw_p.shape[1]
            K.pack_rows(self.dw_p[:nr], nr, 1, row, fg(self.conv.weight).view(nr, row))
        else:
            K.wgrad(T, B, o.H, o.W, G, None, in0, in1, fg(self.conv.weight), eng.workspace, c_valid=self.c_valid,
                    ks=self.ks)
        eng._ready(self.conv.weight)
        if not self.need_dgrad:
            return
        segs, epi = [], EPI_ADD
        for s in self.srcs:
            tgt = s.target()
            if s.fused_bwd:
                epi = EPI_BWD
                segs.append(K.Seg(s.C, out0=s.grad, ref=s.data, slope=0.0, z=s.z, mean=s.mean, istd=s.istd,
                                  partials=s.bpart, accumulate=tgt.written))
            else:
                segs.append(K.Seg(s.C, out0=s.grad if s.needs_grad else eng.scratch_like(s), accumulate=tgt.written))
            tgt.written = True
        if self.mx8:
            K.conv3x3_mx8(B, o.H, o.W, self.g8, self.g8s, None, None, self.w8_dg, self.wsc_dg, self.c0 + self.c1, epi, segs)
        else:
            K.igemm(T, GEMM_S1, B, o.H, o.W, G, None, self.w_dg, self.c0 + self.c1, epi, segs, eng.workspace, ks=self.ks)


class MaxPool2(Op):
    def __init__(self, src, out):
        self.src, self.out = src, out
        out.producer = self
        src.consumers.append(self)

    def fwd(self, eng, training):
        o = self.out
        if o.want_q8 and o.q8 is not None and o.data.dtype == torch.bfloat16 and o.C % 32 == 0:
            K.maxpool2_fwd_mx8(self.src.data, o.data, o.q8, o.q8s)       # the fp8 conv's operand copy comes with the output
            o.q8_serial = eng.fwd_serial
        else:
            K.maxpool2_fwd(self.src.data, o.data)

    def prepare(self, eng):
        s_ = self.src
        _tail_setup(self, eng, eng.B * ((s_.H + 1) // 2) * ((s_.W + 1) // 2) * (s_.C // 8))

    def bwd(self, eng):
        s_ = self.src
        if self.tail:
            K.maxpool2_bwd_tail(self.out.grad, s_.data, s_.grad, s_.written, s_.z, s_.mean, s_.istd, s_.bpart)
        else:
            K.maxpool2_bwd(self.out.grad, s_.data, s_.grad, accumulate=s_.written)
        s_.written = True


class Upsample2x(Op):
    """nn.Upsample(scale_factor=2, bilinear, align_corners=True) + F.pad to out's H x W."""

    def __init__(self, src, out):
        self.src, self.out = src, out
        out.producer = self
        src.consumers.append(self)

    def fwd(self, eng, training):
        o = self.out
        if o.want_q8 and o.q8 is not None and o.data.dtype == torch.bfloat16 and o.C % 32 == 0:
            K.upsample2x_fwd_mx8(self.src.data, o.data, o.q8, o.q8s)
            o.q8_serial = eng.fwd_serial
        else:
            K.upsample2x_fwd(self.src.data, o.data)

    def prepare(self, eng):
        s_ = self.src
        _tail_setup(self, eng, eng.B * s_.H * s_.W * (s_.C // 8))

    def bwd(self, eng):
        s_ = self.src
        if self.tail:
            K.upsample2x_bwd_tail(self.out.grad, s_.grad, s_.written, s_.data, s_.z, s_.mean, s_.istd, s_.bpart)
        else:
            K.upsample2x_bwd(self.out.grad, s_.grad, accumulate=s_.written)
        s_.written = True


class ConvT2x2(Op):
    """nn.ConvTranspose2d(Cin, Cin // 2, kernel_size=2, stride=2) of ``Up(bilinear=False)``: non-overlapping taps, so
    it is a 1x1 GEMM [pixels, Cin] x [Cin, 4 Cout] into a tap-major buffer + a pixel shuffle; the backward is the
    inverse shuffle, a 1x1 input-gradient GEMM and a 1x1 weight-gradient GEMM.  The flat parameter storage of a 4-D
    weight is [dim0][kh][kw][dim1] (flat.py), i.e. [Cin][tap][Cout] here: it IS the GEMM's [K = Cin][N = 4 Cout]
    operand in the buffer's tap-major column order, and the weight-gradient GEMM writes the gradient slice in place."""

    def __init__(self, src, convt, out):
        assert convt.kernel_size == (2, 2) and convt.stride == (2, 2) and convt.padding == (0, 0)
        assert convt.output_padding == (0, 0) and convt.groups == 1 and convt.dilation == (1, 1)
        self.src, self.convt, self.out = src, convt, out
        if (out.H, out.W) != (2 * src.H, 2 * src.W):
            raise NotImplementedError('ConvTranspose2d upsampling followed by F.pad (odd skip sizes) is not implemented')
        out.producer = self
        src.consumers.append(self)

    def prepare(self, eng):
        T, dev, B = eng.dtype, eng.dev, eng.B
        s, o = self.src, self.out
        cin, cout = self.convt.weight.shape[0], self.convt.weight.shape[1]
        assert s.C == cin and o.C == cout and not hasattr(s, 'C_real')
        self.cin, self.cout, self.n4 = cin, cout, 4 * cout
        f32 = dict(dtype=torch.float32, device=dev)
        self.tmp = torch.empty(B, s.H, s.W, self.n4, dtype=T, device=dev)            # forward GEMM output / its gradient
        self.w_fwd = torch.zeros(self.n4, K.s1_row_stride(T, 1, cin), dtype=T, device=dev)
        self.w_dg = torch.zeros(cin, K.s1_row_stride(T, 1, self.n4), dtype=T, device=dev) if s.needs_grad else None
        self.bias4 = torch.zeros(self.n4, **f32) if self.convt.bias is not None else None
        _, ws1 = K.igemm_query(T, GEMM_S1, B, s.H, s.W, cin, 0, self.n4, [self.n4], ks=1)
        ws2 = K.igemm_query(T, GEMM_S1, B, s.H, s.W, self.n4, 0, cin, [cin], ks=1)[1] if s.needs_grad else 0
        ws3 = K.wgrad_workspace_bytes(T, B, s.H, s.W, cin, 0, self.n4, 0, ks=1)
        ws4 = K.channel_sum_workspace_bytes(B * o.H * o.W, cout) if self.bias4 is not None else 0
        self._ws = max(ws1, ws2, ws3, ws4)

    def workspace_bytes(self, eng):
        return self._ws

    def pack(self, eng):
        master = eng._flat_slice(eng.flat_p, self.convt.weight)                         # [Cin][tap][Cout]
        K.pack_transpose_taps(master, self.cin, 1, self.n4, self.w_fwd, flip=False)
        if self.w_dg is not None:
            K.pack_rows(master, self.cin, 1, self.n4, self.w_dg)
        if self.bias4 is not None:
            self.bias4.view(4, self.cout).copy_(eng._flat_slice(eng.flat_p, self.convt.bias).view(1, self.cout).expand(4, -1))

    def fwd(self, eng, training):
        s = self.src
        K.igemm(eng.dtype, GEMM_S1, eng.B, s.H, s.W, s.data, None, self.w_fwd, self.n4, EPI_ADD,
                [K.Seg(self.n4, out0=self.tmp, bias=self.bias4)], eng.workspace, ks=1)
        K.pixel_shuffle2(self.tmp, self.out.data)

    def bwd(self, eng):
        T, B, s, o = eng.dtype, eng.B, self.src, self.out
        fg = lambda p: eng._flat_slice(eng.flat_g, p)
        if self.bias4 is not None:
            K.channel_sum(o.grad, B * o.H * o.W, self.cout, self.cout, fg(self.convt.bias), eng.workspace)
            eng._mark(self.convt.bias)
        K.pixel_shuffle2(self.tmp, o.grad, inverse=True)                              # tmp = d loss / d (GEMM output)
        K.wgrad(T, B, s.H, s.W, s.data, None, self.tmp, None, fg(self.convt.weight), eng.workspace, ks=1)
        eng._ready(self.convt.weight)
        if self.w_dg is not None:
            tgt = s.target()
            K.igemm(T, GEMM_S1, B, s.H, s.W, self.tmp, None, self.w_dg, self.cin, EPI_ADD,
                    [K.Seg(self.cin, out0=s.grad, accumulate=tgt.written)], eng.workspace, ks=1)
            tgt.written = True


class Head1x1(Op):
    """Conv2d(C, 1, 1) + clamp(0, max_depth) (act 0) or sigmoid * max_depth + clamp (act 1); f32 [B,1,H,W] out.
    With ``out_size`` != the source size the reference resizes BEFORE the clamp (F.interpolate(bilinear,
    align_corners=False), rgb_depth_model.py:200-209, binaural_attention_model.py:326-337): the head then runs
    unclamped (act 0 -> identity), adn_resize_bilinear and adn_clamp_range follow; backward in reverse."""

    def __init__(self, src, conv, act, max_depth, out_size=None, clamp_after_resize=True):
        self.src, self.conv, self.act, self.max_depth = src, conv, act, float(max_depth)
        assert conv.kernel_size == (1, 1) and conv.out_channels == 1
        self.out_size = out_size                  # the caller applies the reference's rule (width != output_size)
        # clamp_after_resize False: the Base+Residual heads resize the ACTIVATED map and clamp only base + residual
        # (base_residual_model.py:185-211): head with its activation -> adn_resize_bilinear, nothing else
        self.clamp_after_resize = clamp_after_resize
        if self.out_size is not None and clamp_after_resize:
            self.act = {0: 3, 1: 1}[act]          # sigmoid * max_depth already lies inside the clamp range
        src.consumers.append(self)

    def prepare(self, eng):
        s = self.src
        pixels = eng.B * s.H * s.W
        f32 = dict(dtype=torch.float32, device=eng.dev)
        self.zpre = torch.empty(pixels, **f32)
        self.result = torch.empty(eng.B, 1, s.H, s.W, **f32)
        if self.out_size is not None:
            S = self.out_size
            self.pre, self.gpre = self.result, torch.empty(eng.B, 1, s.H, s.W, **f32)
            self.resized, self.gres = torch.empty(eng.B, S, S, **f32), torch.empty(eng.B, S, S, **f32)
            self.result = torch.empty(eng.B, 1, S, S, **f32)
        self._ws = K.head1x1_bwd_workspace_bytes(pixels, s.C)
        self.c_real = getattr(s, 'C_real', s.C)
        if self.c_real != s.C:                          # zero-padded source: padded weight copy / gradient temporary
            self.w_p, self.dw_p = torch.zeros(s.C, **f32), torch.empty(s.C, **f32)

    def workspace_bytes(self, eng):
        return self._ws

    def _weight(self, eng):
        w = eng._flat_slice(eng.flat_p, self.conv.weight)
        if self.c_real != self.src.C:
            K.pack_rows(w, self.c_real, 1, 1, self.w_p[:self.c_real])
            return self.w_p
        return w

    def fwd(self, eng, training):
        if self.out_size is None:
            K.head1x1_fwd(self.src.data, self._weight(eng), self.conv.bias, self.act, self.max_depth, self.zpre,
                          self.result)
            return
        s = self.src
        K.head1x1_fwd(s.data, self._weight(eng), self.conv.bias, self.act, self.max_depth, self.zpre, self.pre)
        if not self.clamp_after_resize:
            K.resize_bilinear(self.pre.view(eng.B, s.H, s.W), self.out_size, False, self.result.view(eng.B, self.out_size,
                                                                                                     self.out_size))
            return
        K.resize_bilinear(self.pre.view(eng.B, s.H, s.W), self.out_size, False, self.resized)
        K.clamp_range(self.resized, self.max_depth, self.result)

    def bwd_head(self, eng, gout):
        s = self.src
        assert not s.written
        if self.out_size is not None:
            if self.clamp_after_resize:
                K.clamp_range(self.resized, self.max_depth, self.gres, g=gout.contiguous())
                K.resize_bilinear_bwd(self.gres, s.H, s.W, self.gpre.view(eng.B, s.H, s.W))
            else:
                K.resize_bilinear_bwd(gout.contiguous().view(eng.B, self.out_size, self.out_size), s.H, s.W,
                                      self.gpre.view(eng.B, s.H, s.W))
            gout = self.gpre
        padded = self.c_real != s.C
        w = self.w_p if padded else eng._flat_slice(eng.flat_p, self.conv.weight)
        db = eng._flat_slice(eng.flat_g, self.conv.bias) if self.conv.bias is not None else None
        dw = eng._flat_slice(eng.flat_g, self.conv.weight)
        K.head1x1_bwd(gout, self.zpre, s.data, w, self.act, self.max_depth, s.grad, self.dw_p if padded else dw, db,
                      eng.workspace)
        if padded:
            K.pack_rows(self.dw_p[:self.c_real], self.c_real, 1, 1, dw)
        s.written = True
        eng._mark(self.conv.bias)
        eng._ready(self.conv.weight)


class CrossAttention(Op):
    """BinauralCrossAttention.forward (binaural_attention_model.py:106-153) over the stacked pair [left; right]:
    fused q|k|v 1x1 projection (one GEMM, batch 2B), streaming-softmax attention with kv_shift = B (left attends
    right and right attends left in one launch), out projection + gated residual x + gamma * out(att) in the
    GEMM epilogue.  ``xl/xr`` and ``ol/or_`` are stacked pairs; the outputs' gradient buffers alias the inputs'."""

    def __init__(self, xl, xr, mod, ol, or_):
        self.xl, self.xr, self.mod, self.ol, self.or_ = xl, xr, mod, ol, or_
        self.out = ol
        for o in (ol, or_):
            o.producer = self
        ol.alias_of, or_.alias_of = xl, xr
        for x in (xl, xr):
            x.consumers.append(self)

    def prepare(self, eng):
        T, dev, B = eng.dtype, eng.dev, eng.B
        x = self.xl
        C, H, W = x.C, x.H, x.W
        m = self.mod
        self.C, self.dqk = C, m.query.out_channels
        n_qkv = 2 * self.dqk + C
        self.ldq = (n_qkv + 63) // 64 * 64 if n_qkv >= 64 else (n_qkv + 7) // 8 * 8
        f32 = dict(dtype=torch.float32, device=dev)
        act = lambda c: torch.empty(2 * B, H, W, c, dtype=T, device=dev)
        self.qkv, self.att, self.tbuf = act(self.ldq), act(C), act(C)
        self.dqkv = torch.zeros(2 * B, H, W, self.ldq, dtype=T, device=dev)       # padding columns stay zero
        self.lse = torch.empty(2 * B, H * W, **f32)
        self.wm_qkv = torch.zeros(self.ldq, C, **f32)                             # fused [q; k; v; 0] master
        self.b_qkv = torch.zeros(self.ldq, **f32)
        self.w_qkv = torch.empty(self.ldq, K.s1_row_stride(T, 1, C), dtype=T, device=dev)
        self.w_qkv_dg = torch.empty(C, K.s1_row_stride(T, 1, self.ldq), dtype=T, device=dev)
        self.w_out = torch.empty(C, K.s1_row_stride(T, 1, C), dtype=T, device=dev)
        self.w_out_dg = torch.empty(C, K.s1_row_stride(T, 1, C), dtype=T, device=dev)
        self.dwm_qkv = torch.empty(self.ldq, C, **f32)
        self.db_qkv = torch.empty(self.ldq, **f32)
        self.gsum = torch.empty(C, **f32)
        q = lambda n, segs, cin: K.igemm_query(T, GEMM_S1, 2 * B, H, W, cin, 0, n, segs, ks=1)[1]
        rows = 2 * B * H * W
        self._ws = max(q(self.ldq, [self.ldq], C), q(C, [C], C), q(C, [C], self.ldq),
                       K.wgrad_workspace_bytes(T, 2 * B, H, W, C, 0, C, 0, ks=1),
                       K.wgrad_workspace_bytes(T, 2 * B, H, W, self.ldq, 0, C, 0, ks=1),
                       K.channel_sum_workspace_bytes(rows, max(C, self.ldq)), 8192, rows * 4)
        self.scale = 1.0 / (C ** 0.5)

    def workspace_bytes(self, eng):
        return self._ws

    def _slices(self):
        d, C = self.dqk, self.C
        return (0, d), (d, 2 * d), (2 * d, 2 * d + C)

    def pack(self, eng):
        m, C = self.mod, self.C
        for conv, (lo, hi) in zip((m.query, m.key, m.value), self._slices()):
            K.pack_rows(eng._flat_slice(eng.flat_p, conv.weight), hi - lo, 1, C, self.wm_qkv[lo:hi])
            K.pack_rows(eng._flat_slice(eng.flat_p, conv.bias), hi - lo, 1, 1, self.b_qkv[lo:hi])
        K.pack_rows(self.wm_qkv, self.ldq, 1, C, self.w_qkv)
        K.pack_transpose_taps(self.wm_qkv, self.ldq, 1, C, self.w_qkv_dg, flip=False)
        wo = eng._flat_slice(eng.flat_p, m.out.weight)
        K.pack_rows(wo, C, 1, C, self.w_out)
        K.pack_transpose_taps(wo, C, 1, C, self.w_out_dg, flip=False)

    def _views(self, buf):
        N = self.xl.H * self.xl.W
        flat = buf.view(buf.shape[0], N, buf.shape[-1])
        return [flat[:, :, lo:hi] for lo, hi in self._slices()]

    def fwd(self, eng, training):
        T, B, x, m = eng.dtype, eng.B, self.xl, self.mod
        C, H, W, N = self.C, x.H, x.W, x.H * x.W
        xin, xout = x.pair_data, self.ol.pair_data
        K.igemm(T, GEMM_S1, 2 * B, H, W, xin, None, self.w_qkv, self.ldq, EPI_ACT,
                [K.Seg(self.ldq, out0=self.qkv, bias=self.b_qkv, slope=1.0)], eng.workspace, ks=1)
        q, k, v = self._views(self.qkv)
        K.attn_fwd(q, k, v, self.att.view(2 * B, N, C), self.lse, self.dqk, C, B, self.scale)
        K.igemm(T, GEMM_S1, 2 * B, H, W, self.att, None, self.w_out, C, EPI_ADD,
                [K.Seg(C, out0=xout, bias=m.out.bias, scale=m.gamma, final_act=1, ref=xin)], eng.workspace, ks=1)

    def bwd(self, eng):
        T, B, x, m = eng.dtype, eng.B, self.xl, self.mod
        C, H, W, N = self.C, x.H, x.W, x.H * x.W
        G = x.pair_grad                      # d loss / d [left_out; right_out] == residual part of d loss / d x
        assert x.written and self.xr.written
        fg = lambda p: eng._flat_slice(eng.flat_g, p)
        ws = eng.workspace
        K.igemm(T, GEMM_S1, 2 * B, H, W, G, None, self.w_out_dg, C, EPI_ADD, [K.Seg(C, out0=self.tbuf)], ws, ks=1)
        K.channel_sum(G, 2 * B * N, C, C, self.gsum, ws)
        K.wgrad(T, 2 * B, H, W, G, None, self.att, None, fg(m.out.weight), ws, ks=1)
        K.gate_bwd(self.tbuf, self.att, m.gamma, self.gsum, m.out.bias, C, fg(m.gamma), fg(m.out.bias),
                   fg(m.out.weight), ws)
        q, k, v = self._views(self.qkv)
        dq, dk, dv = self._views(self.dqkv)
        K.attn_bwd(q, k, v, self.att.view(2 * B, N, C), self.lse, self.dqk, C, B, self.scale,
                   self.tbuf.view(2 * B, N, C), dq, dk, dv, ws)
        K.wgrad(T, 2 * B, H, W, self.dqkv, None, x.pair_data, None, self.dwm_qkv, ws, ks=1)
        K.channel_sum(self.dqkv, 2 * B * N, self.ldq, self.ldq, self.db_qkv, ws)
        for conv, (lo, hi) in zip((m.query, m.key, m.value), self._slices()):
            K.pack_rows(self.dwm_qkv[lo:hi], hi - lo, 1, C, fg(conv.weight).view(hi - lo, C))
            K.pack_rows(self.db_qkv[lo:hi], hi - lo, 1, 1, fg(conv.bias))
        K.igemm(T, GEMM_S1, 2 * B, H, W, self.dqkv, None, self.w_qkv_dg, C, EPI_ADD,
                [K.Seg(C, out0=G, accumulate=True)], ws, ks=1)
        eng._mark(m.query.bias, m.key.weight, m.key.bias, m.value.weight, m.value.bias, m.out.weight, m.out.bias,
                  m.gamma)
        eng._ready(m.query.weight)


def _pad_ok(a):
    """Every consumer can take extra zero channels at the END of this record: a conv that reads it as its last
    source, the 1x1 head, or an upsample whose own output qualifies."""
    for c in a.consumers:
        if isinstance(c, ConvBNReLU):
            if c.srcs[-1] is not a:
                return False
        elif isinstance(c, Upsample2x):
            if not _pad_ok(c.out):
                return False
        elif not isinstance(c, Head1x1):
            return False
    return True


def flag_solo(a):
    """Per-record decisions taken once the tape is known (records are visited in forward order):
      * fused_bwd: the only consumer is a single-source conv (the mid tensor of a DoubleConv) -- that conv's dgrad
        epilogue applies the ReLU mask + BN statistics;
      * zero padding of the channel count to a multiple of 64 so that narrow layers reach the MFMA kernels (96 -> 128
        in the AdaBins decoder, 32 / 16 -> 64 in the Base+Residual base decoder and at narrow test widths), allowed
        when every consumer tolerates trailing zero channels (_pad_ok); an upsample output inherits its source's
        padding."""
    prod = a.producer
    if isinstance(prod, Upsample2x) and hasattr(prod.src, 'C_real') and a.data is None:
        a.C_real, a.C = prod.src.C_real, prod.src.C
        return
    if not isinstance(prod, ConvBNReLU):
        return
    solo = len(a.consumers) == 1 and isinstance(a.consumers[0], ConvBNReLU) and len(a.consumers[0].srcs) == 1
    a.fused_bwd = solo and a.needs_grad
    if a.C % 64 != 0 and a.C >= 16 and a.data is None and not hasattr(a, 'C_real') and a.consumers and _pad_ok(a):
        a.C_real = a.C
        a.C = (a.C + 63) // 64 * 64


def mark_tail_writers(ops):
    """``ops``: the tape in forward order (heads last).  The backward pass walks it in reverse, so the consumer that sits
    FIRST in the tape writes a record's gradient last; remember it when it is a max-pool or upsample op."""
    pos = {id(op): i for i, op in enumerate(ops)}
    seen = {}
    for op in ops:
        for a in list(getattr(op, 'srcs', [])) + [getattr(op, 'src', None)]:
            if a is not None:
                seen[id(a)] = a
    for a in seen.values():
        a.tail_writer = None
        cons = [c for c in a.consumers]
        if not cons or not isinstance(a.producer, ConvBNReLU):
            continue
        first = min(cons, key=lambda c: pos.get(id(c), 1 << 30))
        if isinstance(first, (MaxPool2, Upsample2x)) and id(first) in pos:
            a.tail_writer = first


def _tail_setup(op, eng, work):
    """Shared by MaxPool2 / Upsample2x.prepare: fuse the source record's ReLU mask + BN-backward statistics into this op's
    backward kernel when this op is the record's last gradient writer."""
    s_ = op.src
    op.tail = False
    if (s_.tail_writer is op and isinstance(s_.producer, ConvBNReLU) and not s_.fused_bwd and s_.needs_grad and
            s_.alias_of is None and s_.pair_data is None and not hasattr(s_, 'C_real') and s_.z is not None and
            os.environ.get('ADN_NO_TAIL_FUSION') is None):
        P = K.tail_stats_blocks(work, s_.C)
        if P > 0:
            s_.tail_fused, s_.bpart_rows = True, P
            s_.bpart = torch.empty(P * 2 * s_.C, dtype=torch.float32, device=eng.dev)
            op.tail = True


class DCEngine(FlatParamEngine):
    """Runs a DoubleConv-family module through libadn.  ``build(engine, B, C, H, W)`` (supplied by the model
    mirror) returns (inputs, ops, head): ``inputs`` = [(Act, first channel, channels)] slices of the NCHW network
    input, the forward op list in execution order, and the output head."""

    def __init__(self, module, build, compute_dtype=torch.bfloat16, model_name='model'):
        self.module = module
        self._build = build
        # compute_dtype torch.float8_e4m3fn = BASELINE config 5's precision: the 3 x 3 convolutions' forward and
        # input-gradient GEMMs run block-scaled fp8 (MX e4m3, csrc/mx8.hip); storage, BatchNorm, the weight-gradient
        # GEMMs and every other op stay bf16
        self.requested_dtype = compute_dtype
        self.mx8 = compute_dtype == torch.float8_e4m3fn
        self.dtype = torch.bfloat16 if self.mx8 else compute_dtype
        self.fwd_serial = 0
        self.model_name = model_name
        self.depth_norm = False
        self._init_flat()
        self.ops = []
        self._scratch = {}

    def scratch_like(self, act):
        """Throw-away gradient target for a source that needs no gradient but shares a dgrad GEMM."""
        key = (act.C, act.H, act.W)
        if key not in self._scratch:
            self._scratch[key] = torch.empty(self.B, act.H, act.W, act.C, dtype=self.dtype, device=self.dev)
        return self._scratch[key]

    def q8_of(self, act):
        """(e4m3 bytes, E8M0 scales) of ``act.data`` for the current forward pass; quantised here unless the producer
        already wrote the copy together with its output."""
        if act.q8_serial != self.fwd_serial:
            K.mx8_quantize(act.data, act.q8, act.q8s)
            act.q8_serial = self.fwd_serial
        return act.q8, act.q8s

    def _prepare(self, x):
        if not self._bound():
            self.bind_parameters()
        B, Cin, H, W = x.shape
        key = (B, Cin, H, W, x.device)
        if self._shape_enter(key):
            return
        self.B, self.dev = B, x.device
        self._scratch = {}
        epc = 8 if self.dtype == torch.bfloat16 else 4
        self.epc = epc
        self.pairs = []
        self.inputs, self.ops, self.head = self._build(self, B, Cin, H, W)
        acts = {}
        for op in self.ops:
            for a in (list(getattr(op, 'srcs', [])) +
                      [getattr(op, k, None) for k in ('src', 'out', 'xl', 'xr', 'ol', 'or_')]):
                if a is not None:
                    acts[id(a)] = a
        self.acts = list(acts.values())
        for a, b in self.pairs:                       # inputs of an aliasing pair are stacked before its outputs
            stack_pair(a, b, B, self.dtype, x.device)
        for a in self.acts:
            flag_solo(a)
            a.alloc(B, self.dtype, x.device)
        mark_tail_writers(self.ops + [self.head])
        ws = 16
        for op in self.ops + [self.head]:
            op.prepare(self)
            ws = max(ws, op.workspace_bytes(self))
        self.workspace = torch.empty(ws // 4 + 4, dtype=torch.float32, device=x.device)
        self.weights_dirty = True
        self._shape_key = key

    def thin_input(self, name, C, H, W):
        """Network input record: C real channels zero-padded to one 16-byte chunk for the MFMA loader."""
        cp = (C + self.epc - 1) // self.epc * self.epc
        a = Act(name, cp, H, W, needs_grad=False)
        a.C_real = C
        return a

    def _pack_weights(self):
        if self.flat_w16 is not None and not self._mirror_fresh():
            _lib.record_py(lambda: self.flat_w16.copy_(self.flat_p))
        for op in self.ops:
            if hasattr(op, 'pack'):
                op.pack(self)
        self.weights_dirty = False
        self.s2_fresh = False
        self._packed_version = self._version_sum()

    def pair(self, a, b):
        """Stack two records along the batch ([left; right]); call in build() order: inputs before aliases."""
        self.pairs.append((a, b))

    def load_input(self, x):
        for act, c_lo, c in self.inputs:
            K.nchw_slice_to_nhwc(x, c_lo, c, act.data)

    def forward(self, x, training):
        if not x.is_cuda:
            raise RuntimeError(f'{self.model_name}.forward needs a HIP device tensor (libadn has no CPU path)')
        x = x.contiguous().float()
        self._prepare(x)
        if self.weights_dirty or self._packed_version != self._version_sum():
            self._pack_weights()
        self.fwd_serial += 1
        self.load_input(x)
        for op in self.ops:
            op.fwd(self, training)
        self.head.fwd(self, training)
        return self.head.result

    def _mark(self, *params):
        for p in params:
            if p is not None:
                self._final.add(id(p))

    def _ready(self, param):
        """``param``'s gradient is final: advance the watermark below which flat_g may still change and hand
        the data-parallel reducer every bucket above it (ops may finish in any order)."""
        self._final.add(id(param))
        idx = self._wm
        while idx > 0 and id(self.param_meta[idx - 1][0]) in self._final:
            idx -= 1
        if idx != self._wm:
            self._wm = idx
            if self.on_grad_ready is not None:
                off = self.param_meta[idx][1]
                _lib.record_py(lambda: self.on_grad_ready(off))

    def backward(self, gout):
        """gout: d loss / d output, f32 [B,1,H,W].  Fills flat_g (all parameters)."""
        for a in self.acts:
            a.written = False
        self._final = set(id(p) for p, _, _ in self.param_meta if not p.requires_grad)
        self._wm = len(self.param_meta)
        gout = gout.contiguous().float()
        self.head.bwd_head(self, gout)
        for op in reversed(self.ops):
            if op.out.needs_grad:
                op.bwd(self)
        if self.on_grad_ready is not None:
            _lib.record_py(lambda: self.on_grad_ready(0))

    def features(self, names):
        """NCHW f32 copies of named activations (return_features=True of the reference forward)."""
        res = {}
        by_name = {a.name: a for a in self.acts}
        for n in names:
            a = by_name[n]
            t = torch.empty(self.B, a.C, a.H, a.W, dtype=torch.float32, device=self.dev)
            K.nhwc_to_nchw(a.data, t)
            res[n] = t[:, :getattr(a, 'C_real', a.C)]           # drop the zero channels of a padded record
        return res


class _DCFunction(torch.autograd.Function):
    """torch.autograd bridge: parameters are inputs so that loss.backward() reaches them."""

    @staticmethod
    def forward(ctx, x, engine, training, *params):
        ctx.engine = engine
        return engine.forward(x, training).clone()

    @staticmethod
    def backward(ctx, gout):
        eng = ctx.engine
        eng.backward(gout)
        return (None, None, None) + tuple(eng.grad_view(p) for p, _, _ in eng.param_meta)


def run_dcnet(engine, x, training):
    if not engine._bound():
        engine.bind_parameters()
    needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p, _, _ in engine.param_meta)
    if needs_grad and training:
        return _DCFunction.apply(x, engine, training, *[p for p, _, _ in engine.param_meta])
    with torch.no_grad():
        return engine.forward(x, training).clone()
