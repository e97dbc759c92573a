"""Tensor-level wrappers over the libadn C ABI (one Python function per entry point).

Everything here only *launches* HIP kernels on torch's current stream; tensors are
allocated by the caller (PyTorch's caching allocator).  No CPU fallbacks.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import (ADN_BF16, ADN_F32, EPI_ACT, EPI_BWD, EPI_FINAL, EPI_RAW, EPI_Z_STATS, GEMM_S2, GEMM_T2,
                   AdnEpiSeg, AdnIgemmDesc, AdnWgradDesc, ptr)

__all__ = ['dtype_code', 'Seg', 'igemm', 'igemm_query', 'wgrad', 'wgrad_workspace_bytes', 'pack_weights',
           'nchw_to_nhwc', 'nhwc_to_nchw', 'bn_fwd_finalize', 'bn_eval_affine', 'bn_act', 'bn_bwd_finalize',
           'bn_bwd_apply', 'loss_stats', 'loss_finish', 'final_act_bwd', 'sum_to_scalar', 'grad_norm',
           'optimizer_step', 'compute_errors', 'frontend', 'convt_n1_forward', 'convt_n1_workspace_bytes']


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# Optional per-launch timing of the GEMM kernels (bench.py): when PROFILE is a list, every igemm/wgrad
# launch appends (label, algorithmic_flops, start_event, end_event); events are recorded on the stream the
# kernel is launched on (torch's current stream), nothing synchronises.
PROFILE = None


def _prof_begin():
    if PROFILE is None:
        return None
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    return ev


def _prof_end(ev0, label, flops):
    if ev0 is None:
        return
    ev1 = torch.cuda.Event(enable_timing=True)
    ev1.record()
    PROFILE.append((label, flops, ev0, ev1))


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return ADN_F32
    if dt == torch.bfloat16:
        return ADN_BF16
    raise TypeError(f'libadn supports float32 and bfloat16 activations, got {dt}')


def _dev(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError('libadn kernels need CUDA/HIP device tensors (no CPU fallback)')


class Seg:
    """One output channel segment of an implicit GEMM (mirrors AdnEpiSeg)."""

    def __init__(self, channels, out0=None, out1=None, ref=None, z=None, mean=None, istd=None, scale=None,
                 shift=None, bias=None, partials=None, slope=0.0, accumulate=False, final_act=0):
        self.channels = channels
        self.tensors = (out0, out1, ref, z, mean, istd, scale, shift, bias, partials)
        self.slope, self.accumulate, self.final_act = slope, accumulate, final_act

    def fill(self, s: AdnEpiSeg):
        _dev(*self.tensors)
        (s.out0, s.out1, s.ref, s.z, s.mean, s.istd, s.scale, s.shift, s.bias, s.partials) = [ptr(t) for t in
                                                                                                 self.tensors]
        s.channels = self.channels
        s.slope = float(self.slope)
        s.accumulate = int(bool(self.accumulate))
        s.final_act = int(self.final_act)


def _igemm_desc(dtype, geom, B, Hs, Ws, in0, in1, w, N, epi, segs, workspace=None):
    d = AdnIgemmDesc()
    d.dtype, d.geom, d.B, d.Hs, d.Ws = dtype_code(dtype), geom, B, Hs, Ws
    d.C0 = in0.shape[-1]
    d.C1 = in1.shape[-1] if in1 is not None else 0
    d.N = N
    _dev(in0, in1, w, workspace)
    d.in0, d.in1, d.w = ptr(in0), ptr(in1), ptr(w)
    d.epi = epi
    segs[0].fill(d.seg[0])
    if len(segs) > 1:
        segs[1].fill(d.seg[1])
    d.workspace = ptr(workspace)
    d.workspace_bytes = workspace.numel() * workspace.element_size() if workspace is not None else 0
    return d


def igemm_query(dtype, geom, B, Hs, Ws, C0, C1, N, seg_channels):
    """(num stats partial rows, workspace bytes) for a shape, without touching the GPU."""
    d = AdnIgemmDesc()
    d.dtype, d.geom, d.B, d.Hs, d.Ws, d.C0, d.C1, d.N = dtype_code(dtype), geom, B, Hs, Ws, C0, C1, N
    d.in0 = d.w = 1
    d.in1 = 1 if C1 else None
    d.epi = EPI_RAW
    d.seg[0].channels = seg_channels[0]
    d.seg[0].out0 = 1
    d.seg[1].channels = seg_channels[1] if len(seg_channels) > 1 else 0
    d.seg[1].out0 = 1
    lib = _lib.load()
    p = lib.adn_igemm_num_partials(C.byref(d))
    wsb = lib.adn_igemm_workspace_bytes(C.byref(d))
    if p < 0 or wsb < 0:
        raise RuntimeError('adn_igemm query failed: ' + lib.adn_last_error().decode())
    return p, wsb


def igemm(dtype, geom, B, Hs, Ws, in0, in1, w, N, epi, segs, workspace=None, algo_c=None):
    """algo_c: real (unpadded) gathered channel count, only used for the algorithmic FLOP count."""
    d = _igemm_desc(dtype, geom, B, Hs, Ws, in0, in1, w, N, epi, segs, workspace)
    ev = _prof_begin()
    _lib.call('adn_igemm', C.byref(d), _stream())
    # algorithmic FLOPs 2*M_out*N*K, K = taps*Cin  (S2: 16 taps on B*Hs*Ws pixels; T2: 4 taps on 4x the pixels)
    flops = 2.0 * B * Hs * Ws * N * 16 * (algo_c if algo_c else d.C0 + d.C1)
    _lib.annotate(label='igemm', flops=flops)
    if ev is not None:
        _prof_end(ev, 'igemm', flops)


def _wgrad_desc(dtype, B, Hs, Ws, plain0, plain1, gath0, gath1, dw, workspace, c_valid=0):
    d = AdnWgradDesc()
    d.c_valid = c_valid
    d.dtype, d.B, d.Hs, d.Ws = dtype_code(dtype), B, Hs, Ws
    _dev(plain0, plain1, gath0, gath1, dw, workspace)
    d.plain0, d.plain1 = ptr(plain0), ptr(plain1)
    d.R0 = plain0.shape[-1]
    d.R1 = plain1.shape[-1] if plain1 is not None else 0
    d.gath0, d.gath1 = ptr(gath0), ptr(gath1)
    d.C0 = gath0.shape[-1]
    d.C1 = gath1.shape[-1] if gath1 is not None else 0
    d.dw = ptr(dw)
    d.workspace = ptr(workspace)
    d.workspace_bytes = workspace.numel() * workspace.element_size() if workspace is not None else 0
    return d


def wgrad_workspace_bytes(dtype, B, Hs, Ws, R0, R1, C0, C1, c_valid=0):
    d = AdnWgradDesc()
    d.c_valid = c_valid
    d.dtype, d.B, d.Hs, d.Ws, d.R0, d.R1, d.C0, d.C1 = dtype_code(dtype), B, Hs, Ws, R0, R1, C0, C1
    d.plain0 = d.gath0 = d.dw = 1
    d.plain1 = 1 if R1 else None
    d.gath1 = 1 if C1 else None
    lib = _lib.load()
    n = lib.adn_wgrad_workspace_bytes(C.byref(d))
    if n < 0:
        raise RuntimeError('adn_wgrad query failed: ' + lib.adn_last_error().decode())
    return n


def wgrad(dtype, B, Hs, Ws, plain0, plain1, gath0, gath1, dw, workspace=None, c_valid=0):
    d = _wgrad_desc(dtype, B, Hs, Ws, plain0, plain1, gath0, gath1, dw, workspace, c_valid)
    ev = _prof_begin()
    _lib.call('adn_wgrad', C.byref(d), _stream())
    flops = 2.0 * B * Hs * Ws * (d.R0 + d.R1) * 16 * (c_valid if c_valid else d.C0 + d.C1)
    _lib.annotate(label='wgrad', flops=flops)
    if ev is not None:
        _prof_end(ev, 'wgrad', flops)


def pack_weights(master, X, Y, dtype, s2_out=None, t2_out=None, y_pad=None):
    """master f32 [X][16][Y] -> s2 [X][16][y_pad] (zero padded) and/or t2 [4][Y][4][X]."""
    _dev(master, s2_out, t2_out)
    _lib.call('adn_pack_weights', ptr(master), X, Y, Y if y_pad is None else y_pad, dtype_code(dtype), ptr(s2_out),
              ptr(t2_out), _stream())


def nchw_to_nhwc(src, dst):
    """src f32 [B,C,H,W] -> dst [B,H,W,Cpad] (Cpad = dst.shape[-1] >= C, extra channels zero)."""
    B, Cc, H, W = src.shape
    _dev(src, dst)
    _lib.call('adn_nchw_to_nhwc', ptr(src), ptr(dst), B, Cc, dst.shape[-1], H, W, dtype_code(dst.dtype), _stream())


def nhwc_to_nchw(src, dst):
    B, Cc, H, W = dst.shape
    _dev(src, dst)
    _lib.call('adn_nhwc_to_nchw', ptr(src), ptr(dst), B, Cc, H, W, dtype_code(src.dtype), _stream())


def bn_fwd_finalize(partials, P, Cc, count, gamma, beta, eps, momentum, running_mean, running_var, nbt, mean,
                    istd, scale, shift):
    _dev(partials, mean, istd, scale, shift)
    _lib.call('adn_bn_fwd_finalize', ptr(partials), P, Cc, count, ptr(gamma), ptr(beta), eps, momentum,
              ptr(running_mean), ptr(running_var), ptr(nbt), ptr(mean), ptr(istd), ptr(scale), ptr(shift),
              _stream())


def bn_eval_affine(gamma, beta, running_mean, running_var, eps, scale, shift):
    _dev(running_mean, scale)
    _lib.call('adn_bn_eval_affine', ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), eps,
              running_mean.numel(), ptr(scale), ptr(shift), _stream())


def bn_act(z, pixels, Cc, scale, shift, slope, out_leaky=None, out_relu=None):
    _dev(z, scale, shift, out_leaky, out_relu)
    _lib.call('adn_bn_act', ptr(z), pixels, Cc, dtype_code(z.dtype), ptr(scale), ptr(shift), slope, ptr(out_leaky),
              ptr(out_relu), _stream())


def bn_bwd_finalize(partials, P, Cc, count, dgamma, dbeta, coef):
    _dev(partials, coef)
    _lib.call('adn_bn_bwd_finalize', ptr(partials), P, Cc, count, ptr(dgamma), ptr(dbeta), ptr(coef), _stream())


def bn_bwd_apply(g, z, pixels, Cc, scale, mean, istd, coef):
    _dev(g, z)
    _lib.call('adn_bn_bwd_apply', ptr(g), ptr(z), pixels, Cc, dtype_code(g.dtype), ptr(scale), ptr(mean), ptr(istd),
              ptr(coef), _stream())


def loss_stats(pred, gt, scale, mask_mode, eps, stats, workspace):
    _dev(pred, gt, stats, workspace)
    _lib.call('adn_loss_stats', ptr(pred), ptr(gt), pred.numel(), scale, mask_mode, eps, ptr(stats), ptr(workspace),
              workspace.numel() * workspace.element_size(), _stream())


def loss_finish(pred, gt, scale, mask_mode, eps, stats, criterion, l1_weight, silog_weight, silog_lambda, loss_out,
                grad):
    _dev(pred, gt, stats, loss_out, grad)
    _lib.call('adn_loss_finish', ptr(pred), ptr(gt), pred.numel(), scale, mask_mode, eps, ptr(stats), criterion,
              l1_weight, silog_weight, silog_lambda, ptr(loss_out), ptr(grad), _stream())


def final_act_bwd(gout, out, final_act, dz):
    """dz [pixels, Cpad]: channel 0 = gout * act'(out), padded channels zero."""
    _dev(gout, out, dz)
    _lib.call('adn_final_act_bwd', ptr(gout), ptr(out), out.numel(), final_act, dtype_code(dz.dtype),
              dz.numel() // out.numel(), ptr(dz), _stream())


def convt_n1_workspace_bytes(B, Hs, Ws):
    return _lib.load().adn_convt_n1_workspace_bytes(B, Hs, Ws)


def convt_n1_forward(dtype, B, Hs, Ws, in0, in1, w_master, bias, final_act, out, workspace):
    _dev(in0, in1, w_master, bias, out, workspace)
    ev = _prof_begin()
    _lib.call('adn_convt_n1_forward', dtype_code(dtype), B, Hs, Ws, ptr(in0), in0.shape[-1], ptr(in1),
              in1.shape[-1] if in1 is not None else 0, ptr(w_master), ptr(bias), final_act, ptr(out), ptr(workspace),
              workspace.numel() * workspace.element_size(), _stream())
    if ev is not None:
        cin = in0.shape[-1] + (in1.shape[-1] if in1 is not None else 0)
        _prof_end(ev, 'convt_n1', 2.0 * B * Hs * Ws * 16 * cin)


def sum_to_scalar(x, out, workspace):
    _dev(x, out, workspace)
    _lib.call('adn_sum_to_scalar', ptr(x), x.numel(), dtype_code(x.dtype), ptr(out), ptr(workspace),
              workspace.numel() * workspace.element_size(), _stream())


def grad_norm(grads, max_norm, state, workspace):
    _dev(grads, state, workspace)
    _lib.call('adn_grad_norm', ptr(grads), grads.numel(), max_norm, ptr(state), ptr(workspace),
              workspace.numel() * workspace.element_size(), _stream())


def optimizer_step(params, grads, exp_avg, exp_avg_sq, kind, lr, beta1, beta2, eps, weight_decay, use_clip, state,
                   bf16_copy=None):
    _dev(params, grads, state, bf16_copy)
    _lib.call('adn_optimizer_step', ptr(params), ptr(grads), ptr(exp_avg), ptr(exp_avg_sq), params.numel(), kind, lr,
              beta1, beta2, eps, weight_decay, int(use_clip), ptr(state), ptr(bf16_copy), _stream())


def pack_t2_multi(flat_master, table, layers, total_blocks, dtype, t2_base):
    _dev(flat_master, table, t2_base)
    _lib.call('adn_pack_t2_multi', ptr(flat_master), ptr(table), layers, total_blocks, dtype_code(dtype), ptr(t2_base),
              _stream())


def compute_errors(gt, pred, out7):
    """gt/pred [samples, pixels] f32 -> out7 [samples, 7]."""
    _dev(gt, pred, out7)
    _lib.call('adn_compute_errors', ptr(gt), ptr(pred), gt.shape[0], gt.shape[1], ptr(out7), None, 0, _stream())


def frontend(wave, mode, S, antialias, out, workspace):
    _dev(wave, out, workspace)
    B, _, T = wave.shape
    _lib.call('adn_frontend', ptr(wave), B, T, mode, S, int(antialias), ptr(out), ptr(workspace),
              workspace.numel() * workspace.element_size(), _stream())


def frontend_workspace_bytes(B, T, mode):
    return _lib.load().adn_frontend_workspace_bytes(B, T, mode)


def resize_bilinear(src, S, antialias, out):
    """src f32 [planes, H, W] -> out f32 [planes, S, S] (align_corners=False, optional antialias)."""
    _dev(src, out)
    planes, H, W = src.shape
    _lib.call('adn_resize_bilinear', ptr(src), planes, H, W, S, int(antialias), ptr(out), _stream())
