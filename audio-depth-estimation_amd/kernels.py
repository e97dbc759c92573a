"""Tensor-level wrappers over the libadn C ABI (one Python function per entry point).

Everything here only *launches* HIP kernels on torch's current stream; tensors are
allocated by the caller (PyTorch's caching allocator).  No CPU fallbacks.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import AdnAttnDesc, AdnDistillSmall  # noqa: E402
from ._lib import (ADN_BF16, ADN_F32, EPI_ACT, EPI_ADD, EPI_BWD, EPI_FINAL, EPI_RAW, EPI_Z_STATS, GEMM_S1, GEMM_S2, GEMM_T2,
                   AdnEpiSeg, AdnIgemmDesc, AdnMx8ConvDesc, AdnWgradDesc, ptr)

__all__ = ['dtype_code', 'Seg', 'igemm', 'igemm_query', 'wgrad', 'wgrad_workspace_bytes', 'pack_weights',
           'nchw_to_nhwc', 'nhwc_to_nchw', 'bn_fwd_finalize', 'bn_eval_affine', 'bn_act', 'bn_bwd_finalize',
           'bn_bwd_apply', 'loss_stats', 'loss_finish', 'final_act_bwd', 'sum_to_scalar', 'grad_norm', 'grad_norm_ranges', 'wgrad_sq_count',
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


def _igemm_desc(dtype, geom, B, Hs, Ws, in0, in1, w, N, epi, segs, workspace=None, ks=0):
    d = AdnIgemmDesc()
    d.ks = ks
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


def igemm_query(dtype, geom, B, Hs, Ws, C0, C1, N, seg_channels, ks=0, epi=EPI_RAW):
    """(num stats partial rows, workspace bytes) for a shape, without touching the GPU.  ``epi``: the epilogue the launch
    will use -- the kernel choice (and with it the number of partial rows of the Z_STATS / BWD epilogues) depends on it."""
    d = AdnIgemmDesc()
    d.ks = ks
    d.dtype, d.geom, d.B, d.Hs, d.Ws, d.C0, d.C1, d.N = dtype_code(dtype), geom, B, Hs, Ws, C0, C1, N
    d.in0 = d.w = 1
    d.in1 = 1 if C1 else None
    d.epi = epi
    d.seg[0].channels = seg_channels[0]
    d.seg[0].out0 = d.seg[0].ref = 1                    # (dummy non-null operands: the query only validates and plans)
    d.seg[1].channels = seg_channels[1] if len(seg_channels) > 1 else 0
    d.seg[1].out0 = d.seg[1].ref = 1
    lib = _lib.load()
    p = lib.adn_igemm_num_partials(C.byref(d))
    wsb = lib.adn_igemm_workspace_bytes(C.byref(d))
    if p < 0 or wsb < 0:
        raise RuntimeError('adn_igemm query failed: ' + lib.adn_last_error().decode())
    return p, wsb


def igemm(dtype, geom, B, Hs, Ws, in0, in1, w, N, epi, segs, workspace=None, algo_c=None, ks=0):
    """algo_c: real (unpadded) gathered channel count, only used for the algorithmic FLOP count."""
    d = _igemm_desc(dtype, geom, B, Hs, Ws, in0, in1, w, N, epi, segs, workspace, ks)
    ev = _prof_begin()
    _lib.call('adn_igemm', C.byref(d), _stream())
    # algorithmic FLOPs 2*M_out*N*K, K = taps*Cin  (S2: 16 taps on B*Hs*Ws pixels; T2: 4 taps on 4x the pixels;
    # S1: ks*ks taps on B*Hs*Ws pixels)
    flops = 2.0 * B * Hs * Ws * N * (ks * ks if geom == GEMM_S1 else 16) * (algo_c if algo_c else d.C0 + d.C1)
    _lib.annotate(label='igemm', flops=flops)
    if ev is not None:
        _prof_end(ev, 'igemm', flops)


def _wgrad_desc(dtype, B, Hs, Ws, plain0, plain1, gath0, gath1, dw, workspace, c_valid=0, ks=0):
    d = AdnWgradDesc()
    d.c_valid = c_valid
    d.geom, d.ks = (GEMM_S1, ks) if ks else (0, 0)
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


def wgrad_workspace_bytes(dtype, B, Hs, Ws, R0, R1, C0, C1, c_valid=0, ks=0):
    d = AdnWgradDesc()
    d.c_valid = c_valid
    d.geom, d.ks = (GEMM_S1, ks) if ks else (0, 0)
    d.dtype, d.B, d.Hs, d.Ws, d.R0, d.R1, d.C0, d.C1 = dtype_code(dtype), B, Hs, Ws, R0, R1, C0, C1
    d.plain0 = d.gath0 = d.dw = 1
    d.plain1 = 1 if R1 else None
    d.gath1 = 1 if C1 else None
    lib = _lib.load()
    n = lib.adn_wgrad_workspace_bytes(C.byref(d))
    if n < 0:
        raise RuntimeError('adn_wgrad query failed: ' + lib.adn_last_error().decode())
    return n


def wgrad_batchable(dtype, B, Hs, Ws, R0, R1, C0, C1, c_valid=0):
    """(class, norm partials) of this k4 weight gradient as a problem of a wgrad_batch launch; class 0 = not batchable,
    problems of one launch share their class (host-only query)."""
    d = AdnWgradDesc()
    d.c_valid = c_valid
    d.geom, d.ks = 0, 0
    d.dtype, d.B, d.Hs, d.Ws, d.R0, d.R1, d.C0, d.C1 = dtype_code(dtype), B, Hs, Ws, R0, R1, C0, C1
    d.plain0 = d.gath0 = d.dw = 1
    d.plain1 = 1 if R1 else None
    d.gath1 = 1 if C1 else None
    lib = _lib.load()
    return int(lib.adn_wgrad_batchable(C.byref(d))), int(lib.adn_wgrad_batch_sq_count(C.byref(d)))


def wgrad_batch(dtype, B, problems):
    """problems: list of (Hs, Ws, plain0, plain1, gath0, gath1, dw, sq): up to 8 weight gradients in one launch."""
    arr = (AdnWgradDesc * len(problems))()
    flops = 0.0
    for slot, (Hs, Ws, p0, p1, g0, g1, dw, sq) in zip(arr, problems):
        d = _wgrad_desc(dtype, B, Hs, Ws, p0, p1, g0, g1, dw, None)
        d.sq_partials = ptr(sq)
        C.memmove(C.byref(slot), C.byref(d), C.sizeof(AdnWgradDesc))
        flops += 2.0 * B * Hs * Ws * (d.R0 + d.R1) * 16 * (d.C0 + d.C1)
    ev = _prof_begin()
    _lib.call('adn_wgrad_batch', arr, len(problems), _stream())
    _lib.annotate(label='wgrad', flops=flops)
    if ev is not None:
        _prof_end(ev, 'wgrad', flops)


def wgrad_patch_batch_workspace_bytes(dtype, B, shapes):
    """shapes: list of (Hs, Ws, R0, R1, C0, C1).  Bytes of slab scratch of one wgrad_patch_batch launch over them, or -1 when
    some layer is not a patch-staged one (host-only query)."""
    arr = (AdnWgradDesc * len(shapes))()
    for d, (Hs, Ws, R0, R1, C0, C1) in zip(arr, shapes):
        d.dtype, d.B, d.Hs, d.Ws, d.R0, d.R1, d.C0, d.C1 = dtype_code(dtype), B, Hs, Ws, R0, R1, C0, C1
        d.plain0 = d.gath0 = d.dw = 1
        d.plain1 = 1 if R1 else None
        d.gath1 = 1 if C1 else None
    return int(_lib.load().adn_wgrad_patch_batch_workspace_bytes(arr, len(shapes)))


def wgrad_patch_batch(dtype, B, problems, workspace):
    """problems: list of (Hs, Ws, plain0, plain1, gath0, gath1, dw, sq): 2 .. 4 patch-staged weight gradients in one launch
    with 1/n of the pixel splits each (+ one slab sum per problem)."""
    arr = (AdnWgradDesc * len(problems))()
    flops = 0.0
    for slot, (Hs, Ws, p0, p1, g0, g1, dw, sq) in zip(arr, problems):
        d = _wgrad_desc(dtype, B, Hs, Ws, p0, p1, g0, g1, dw, workspace)
        d.sq_partials = ptr(sq)
        C.memmove(C.byref(slot), C.byref(d), C.sizeof(AdnWgradDesc))
        flops += 2.0 * B * Hs * Ws * (d.R0 + d.R1) * 16 * (d.C0 + d.C1)
    ev = _prof_begin()
    _lib.call('adn_wgrad_patch_batch', arr, len(problems), _stream())
    _lib.annotate(label='wgrad', flops=flops)
    if ev is not None:
        _prof_end(ev, 'wgrad', flops)


def wgrad_sq_count(dtype, B, Hs, Ws, R0, R1, C0, C1, c_valid=0, ks=0):
    """Partial sums of dW^2 adn_wgrad leaves behind when asked to (``sq=``); 0 = this layer's kernel has no fused form."""
    d = AdnWgradDesc()
    d.c_valid = c_valid
    d.geom, d.ks = (GEMM_S1, ks) if ks else (0, 0)
    d.dtype, d.B, d.Hs, d.Ws, d.R0, d.R1, d.C0, d.C1 = dtype_code(dtype), B, Hs, Ws, R0, R1, C0, C1
    d.plain0 = d.gath0 = d.dw = 1
    d.plain1 = 1 if R1 else None
    d.gath1 = 1 if C1 else None
    return int(_lib.load().adn_wgrad_sq_count(C.byref(d)))


def wgrad(dtype, B, Hs, Ws, plain0, plain1, gath0, gath1, dw, workspace=None, c_valid=0, ks=0, sq=None):
    """ks=0: the k4 s2 p1 pair (gathered tensor on the 2x grid); ks in {1,3}: stride-1 ks x ks conv (same grid).
    sq (optional, f64, wgrad_sq_count(...) elements): receives the partial sums of dW^2 for the fused gradient norm."""
    d = _wgrad_desc(dtype, B, Hs, Ws, plain0, plain1, gath0, gath1, dw, workspace, c_valid, ks)
    if sq is not None:
        _dev(sq)
        if sq.dtype != torch.float64:
            raise TypeError('wgrad: sq must be float64')
        d.sq_partials = ptr(sq)
    ev = _prof_begin()
    _lib.call('adn_wgrad', C.byref(d), _stream())
    flops = 2.0 * B * Hs * Ws * (d.R0 + d.R1) * (ks * ks if ks else 16) * (c_valid if c_valid else d.C0 + d.C1)
    _lib.annotate(label='wgrad', flops=flops)
    if ev is not None:
        _prof_end(ev, 'wgrad', flops)


def pack_weights(master, X, Y, dtype, s2_out=None, t2_out=None, y_pad=None):
    """master f32 [X][16][Y] -> s2 [X][16][y_pad] (zero padded) and/or t2 [4][Y][4][X]."""
    _dev(master, s2_out, t2_out)
    _lib.call('adn_pack_weights', ptr(master), X, Y, Y if y_pad is None else y_pad, dtype_code(dtype), ptr(s2_out),
              ptr(t2_out), _stream())


def s1_row_stride(dtype, taps, cin):
    """Elements per packed S1 weight row: taps*cin rounded up to the 128-byte K-step of the GEMM."""
    bk = 128 // (2 if dtype == torch.bfloat16 else 4)
    return -(-taps * cin // bk) * bk


def pack_rows(master, X, taps, Y, out, y_pad=None):
    """master f32 [X][taps][Y] -> out [X][row_stride] = [X][taps][y_pad] + zero tail (S1 forward operand)."""
    _dev(master, out)
    _lib.call('adn_pack_rows', ptr(master), X, taps, Y, Y if y_pad is None else y_pad,
              out.stride(0) if out.dim() > 1 else 1, dtype_code(out.dtype), ptr(out), _stream())


def pack_transpose_taps(master, X, taps, Y, out, flip=True):
    """master f32 [X][taps][Y] -> out [Y][row_stride], out[y][t'*X+x] = master[x][t][y] (S1 dgrad operand)."""
    _dev(master, out)
    _lib.call('adn_pack_transpose_taps', ptr(master), X, taps, Y, int(flip), out.stride(0), dtype_code(out.dtype),
              ptr(out), _stream())


# ---- block-scaled fp8 (MX e4m3) 3 x 3 convolution path (csrc/mx8.hip) ------------------------------------------------------
def mx8_quantize(src, dst8, scales):
    """src bf16 [..., C] -> dst8 uint8 (e4m3 bits, same shape) + scales uint8 (E8M0) [..., C/32]."""
    _dev(src, dst8, scales)
    if src.dtype != torch.bfloat16 or dst8.dtype != torch.uint8 or scales.dtype != torch.uint8:
        raise TypeError('mx8_quantize: src bf16, dst / scales uint8')
    Cc = src.shape[-1]
    _lib.call('adn_mx8_quantize', ptr(src), src.numel() // Cc, Cc, ptr(dst8), ptr(scales), _stream())


def bn_act_mx8(z, pixels, Cc, scale, shift, out_relu, out8, out_scales):
    """bn_act (BN affine + ReLU, bf16) that also writes the MX-fp8 copy of its output."""
    _dev(z, scale, shift, out_relu, out8, out_scales)
    _lib.call('adn_bn_act_mx8', ptr(z), pixels, Cc, ptr(scale), ptr(shift), ptr(out_relu), ptr(out8), ptr(out_scales),
              _stream())


def bn_bwd_apply_mx8(g, z, pixels, Cc, scale, mean, istd, coef, out8, out_scales):
    """bn_bwd_apply (in place on g, bf16) that also writes the MX-fp8 copy of the resulting d loss / d z."""
    _dev(g, z, scale, mean, istd, coef, out8, out_scales)
    _lib.call('adn_bn_bwd_apply_mx8', ptr(g), ptr(z), pixels, Cc, ptr(scale), ptr(mean), ptr(istd), ptr(coef), ptr(out8),
              ptr(out_scales), _stream())


def mx8_pack_shapes(X, Y, transpose):
    """(w8 shape, wsc shape) of the packed MX operand of a [X,Y,3,3] conv weight."""
    rows, kc = (Y, X) if transpose else (X, Y)
    return (rows, 10, kc), (rows, kc // 64, 5, 4)


def mx8_pack(master, X, Y, transpose, w8, wsc):
    """master f32 [X][9][Y] -> w8 / wsc (see include/adn.h adn_mx8_pack)."""
    _dev(master, w8, wsc)
    _lib.call('adn_mx8_pack', ptr(master), X, Y, int(transpose), ptr(w8), ptr(wsc), _stream())


def conv3x3_mx8_num_partials(B, H, W, N=128, C0=64, C1=0):
    """Stats partial rows the kernel writes for this shape (queried from the library: 128 or 256 pixels per workgroup)."""
    d = AdnMx8ConvDesc()
    d.B, d.H, d.W, d.C0, d.C1, d.N = B, H, W, C0, C1, N
    d.in0 = d.sc0 = d.w = d.wsc = 1
    d.in1 = d.sc1 = 1 if C1 else None
    d.epi = EPI_Z_STATS
    d.seg[0].channels, d.seg[0].out0 = N, 1
    lib = _lib.load()
    n = lib.adn_conv3x3_mx8_num_partials(C.byref(d))
    if n < 0:
        raise RuntimeError('adn_conv3x3_mx8 query failed: ' + lib.adn_last_error().decode())
    return n


def conv3x3_mx8(B, H, W, in0, sc0, in1, sc1, w8, wsc, N, epi, segs):
    """3 x 3 stride-1 conv on MX-fp8 operands; outputs / epilogue operands of ``segs`` are bf16 (as igemm)."""
    d = AdnMx8ConvDesc()
    d.B, d.H, d.W = B, H, W
    d.C0 = in0.shape[-1]
    d.C1 = in1.shape[-1] if in1 is not None else 0
    d.N = N
    _dev(in0, sc0, in1, sc1, w8, wsc)
    d.in0, d.sc0, d.in1, d.sc1, d.w, d.wsc = ptr(in0), ptr(sc0), ptr(in1), ptr(sc1), ptr(w8), ptr(wsc)
    d.epi = epi
    segs[0].fill(d.seg[0])
    if len(segs) > 1:
        segs[1].fill(d.seg[1])
    ev = _prof_begin()
    _lib.call('adn_conv3x3_mx8', C.byref(d), _stream())
    flops = 2.0 * B * H * W * N * 9 * (d.C0 + d.C1)
    _lib.annotate(label='conv3x3_mx8', flops=flops)
    if ev is not None:
        _prof_end(ev, 'conv3x3_mx8', flops)


def nchw_to_nhwc(src, dst):
    """src f32 [B,C,H,W] -> dst [B,H,W,Cpad] (Cpad = dst.shape[-1] >= C, extra channels zero)."""
    B, Cc, H, W = src.shape
    _dev(src, dst)
    _lib.call('adn_nchw_to_nhwc', ptr(src), ptr(dst), B, Cc, dst.shape[-1], H, W, dtype_code(dst.dtype), _stream())


def nhwc_to_nchw(src, dst):
    B, Cc, H, W = dst.shape
    _dev(src, dst)
    _lib.call('adn_nhwc_to_nchw', ptr(src), ptr(dst), B, Cc, H, W, dtype_code(src.dtype), _stream())


BN_REDUCE_ROWS, BN_REDUCE_SLICES = 2048, 64


def _bn_prereduce(partials, P, Cc, scratch):
    """Layers with thousands of partial rows: coalesced pre-reduction to 64 rows in ``scratch`` (f32 [>= 64 * 2 * C])."""
    if scratch is None or P < BN_REDUCE_ROWS or Cc % 32 != 0 or scratch.numel() < BN_REDUCE_SLICES * 2 * Cc:
        return partials, P
    _dev(partials, scratch)
    _lib.call('adn_bn_partials_reduce', ptr(partials), P, Cc, BN_REDUCE_SLICES, ptr(scratch), _stream())
    return scratch, BN_REDUCE_SLICES


def bn_fwd_finalize(partials, P, Cc, count, gamma, beta, eps, momentum, running_mean, running_var, nbt, mean,
                    istd, scale, shift, scratch=None):
    partials, P = _bn_prereduce(partials, P, Cc, scratch)
    _dev(partials, mean, istd, scale, shift)
    _lib.call('adn_bn_fwd_finalize', ptr(partials), P, Cc, count, ptr(gamma), ptr(beta), eps, momentum,
              ptr(running_mean), ptr(running_var), ptr(nbt), ptr(mean), ptr(istd), ptr(scale), ptr(shift),
              _stream())


def bn_fwd_fused(partials, P, Cc, count, gamma, beta, eps, momentum, running_mean, running_var, nbt, mean, istd, scale,
                 shift, z, pixels, slope, out_leaky, out_relu):
    """bn_fwd_finalize + bn_act in one launch (small tensors, C % 32 == 0)."""
    _dev(partials, mean, istd, scale, shift, z, out_leaky, out_relu)
    _lib.call('adn_bn_fwd_fused', ptr(partials), P, Cc, count, ptr(gamma), ptr(beta), eps, momentum, ptr(running_mean),
              ptr(running_var), ptr(nbt), ptr(mean), ptr(istd), ptr(scale), ptr(shift), ptr(z), pixels,
              dtype_code(z.dtype), float(slope), ptr(out_leaky), ptr(out_relu), _stream())


def bn_bwd_fused(partials, P, Cc, count, dgamma, dbeta, g, z, pixels, scale, mean, istd):
    """bn_bwd_finalize + bn_bwd_apply (g in place) in one launch (small tensors, C % 32 == 0)."""
    _dev(partials, g, z, scale, mean, istd)
    _lib.call('adn_bn_bwd_fused', ptr(partials), P, Cc, count, ptr(dgamma), ptr(dbeta), ptr(g), ptr(z), pixels,
              dtype_code(g.dtype), ptr(scale), ptr(mean), ptr(istd), _stream())


def bn_eval_affine(gamma, beta, running_mean, running_var, eps, scale, shift):
    _dev(running_mean, scale)
    _lib.call('adn_bn_eval_affine', ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), eps,
              running_mean.numel(), ptr(scale), ptr(shift), _stream())


def bn_act(z, pixels, Cc, scale, shift, slope, out_leaky=None, out_relu=None):
    _dev(z, scale, shift, out_leaky, out_relu)
    _lib.call('adn_bn_act', ptr(z), pixels, Cc, dtype_code(z.dtype), ptr(scale), ptr(shift), slope, ptr(out_leaky),
              ptr(out_relu), _stream())


def bn_bwd_finalize(partials, P, Cc, count, dgamma, dbeta, coef, scratch=None):
    partials, P = _bn_prereduce(partials, P, Cc, scratch)
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


def loss_finish_dz(pred, gt, scale, mask_mode, eps, stats, criterion, l1_weight, silog_weight, silog_lambda, loss_out, dz,
                   final_act, bias_grad, workspace):
    """loss_finish + final_act_bwd + the last layer's bias gradient in one pass (1-channel prediction = activation output)."""
    _dev(pred, gt, stats, loss_out, dz, bias_grad, workspace)
    _lib.call('adn_loss_finish_dz', ptr(pred), ptr(gt), pred.numel(), scale, mask_mode, eps, ptr(stats), criterion,
              l1_weight, silog_weight, silog_lambda, ptr(loss_out), ptr(dz), final_act, ptr(bias_grad), ptr(workspace),
              workspace.numel() * workspace.element_size(), _stream())


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


def l0_forward(x, w_master, B, Hs, Ws, slope, out_leaky, out_relu):
    """First conv (2 -> 64 channels) of the bf16 path: x f32 NCHW [B,2,2Hs,2Ws], w f32 [64][16][2] (parameter memory)."""
    _dev(x, w_master, out_leaky, out_relu)
    _lib.call('adn_l0_forward', ptr(x), ptr(w_master), B, Hs, Ws, x.shape[1], w_master.numel() // (16 * x.shape[1]),
              float(slope), ptr(out_leaky), ptr(out_relu), _stream())
    _lib.annotate(label='edge', flops=2.0 * B * Hs * Ws * 64 * 16 * x.shape[1])


def d0_dgrad_num_partials(B, Hs, Ws):
    return _lib.load().adn_d0_dgrad_num_partials(B, Hs, Ws)


def d0_dgrad(dz, w_master, B, Hs, Ws, seg0, seg1):
    """Input gradient of the last transposed conv (128 -> 1): dz f32 [B,1,2Hs,2Ws], segments = skip / up half."""
    _dev(dz, w_master)
    s0, s1 = AdnEpiSeg(), AdnEpiSeg()
    seg0.fill(s0)
    seg1.fill(s1)
    _lib.call('adn_d0_dgrad', ptr(dz), ptr(w_master), B, Hs, Ws, C.byref(s0), C.byref(s1), _stream())
    _lib.annotate(label='edge', flops=2.0 * B * Hs * Ws * 128 * 16)


def thin_wgrad_workspace_bytes(B, Hs, Ws, ct, c0, c1):
    return _lib.load().adn_thin_wgrad_workspace_bytes(B, Hs, Ws, ct, c0, c1)


def thin_wgrad(thin, plain0, plain1, B, Hs, Ws, dw, workspace):
    """dw[c][tap*ct + t] = sum plain[.., c] * thin window; thin f32 [B,ct,2Hs,2Ws] planar, plain bf16 NHWC."""
    _dev(thin, plain0, plain1, dw, workspace)
    ct = thin.shape[1]
    c0 = plain0.shape[-1]
    c1 = plain1.shape[-1] if plain1 is not None else 0
    _lib.call('adn_thin_wgrad', ptr(thin), ct, ptr(plain0), c0, ptr(plain1), c1, B, Hs, Ws, ptr(dw), ptr(workspace),
              workspace.numel() * workspace.element_size(), _stream())
    _lib.annotate(label='edge', flops=2.0 * B * Hs * Ws * (c0 + c1) * 16 * ct)


def sum_to_scalar(x, out, workspace):
    _dev(x, out, workspace)
    _lib.call('adn_sum_to_scalar', ptr(x), x.numel(), dtype_code(x.dtype), ptr(out), ptr(workspace),
              workspace.numel() * workspace.element_size(), _stream())


def grad_norm(grads, max_norm, state, workspace):
    _dev(grads, state, workspace)
    _lib.call('adn_grad_norm', ptr(grads), grads.numel(), max_norm, ptr(state), ptr(workspace),
              workspace.numel() * workspace.element_size(), _stream())


def grad_sqsum_count(n):
    """Number of f64 partial sums adn_grad_sqsum_partials writes for n elements."""
    return _lib.load().adn_grad_norm_workspace_bytes(n) // 8


def grad_sqsum_partials(grads, partials):
    """partials f64[grad_sqsum_count(n)] <- sums of squares of the f32 slice ``grads`` (one all-reduce bucket)."""
    _dev(grads, partials)
    _lib.call('adn_grad_sqsum_partials', ptr(grads), grads.numel(), ptr(partials), _nbytes(partials), _stream())


def grad_norm_ranges(grads, ranges, extra, max_norm, state, workspace):
    """Total gradient norm + clip coefficient from (a) ``ranges`` (int64 [n, 2] rows (offset, length <= 8192), both
    multiples of 4) of ``grads`` and (b) ``extra``: partial sums of squares the gradient kernels already wrote."""
    _dev(grads, ranges, extra, state, workspace)
    nr = 0 if ranges is None else ranges.shape[0]
    ne = 0 if extra is None else extra.numel()
    _lib.call('adn_grad_norm_ranges', ptr(grads), ptr(ranges), nr, ptr(extra), ne, max_norm, ptr(state), ptr(workspace),
              workspace.numel() * workspace.element_size(), _stream())


def optimizer_step(params, grads, exp_avg, exp_avg_sq, kind, lr, beta1, beta2, eps, weight_decay, use_clip, state,
                   bf16_copy=None):
    _dev(params, grads, state, bf16_copy)
    _lib.call('adn_optimizer_step', ptr(params), ptr(grads), ptr(exp_avg), ptr(exp_avg_sq), params.numel(), kind, lr,
              beta1, beta2, eps, weight_decay, int(use_clip), ptr(state), ptr(bf16_copy), _stream())


def pack_t2_multi(flat_master, table, layers, total_blocks, dtype, t2_base):
    """flat_master: the f32 masters or their bf16 mirror (same offsets)."""
    _dev(flat_master, table, t2_base)
    _lib.call('adn_pack_t2_multi', ptr(flat_master), dtype_code(flat_master.dtype), ptr(table), layers, total_blocks,
              dtype_code(dtype), ptr(t2_base), _stream())


def edge_loss(pred, gt, lambdas, stats, terms, grad, workspace):
    """Edge-aware / smoothness loss (adn_edge_loss): pred, gt f32 [B,1,H,W]; lambdas = (recon, edge, smooth)."""
    _dev(pred, gt, stats, terms, grad, workspace)
    B, H, W = pred.shape[0], pred.shape[-2], pred.shape[-1]
    _lib.call('adn_edge_loss', ptr(pred), ptr(gt), B, H, W, float(lambdas[0]), float(lambdas[1]), float(lambdas[2]),
              ptr(stats), ptr(terms), ptr(grad), ptr(workspace), workspace.numel() * workspace.element_size(), _stream())


def edge_loss_workspace_bytes(B, H, W):
    return _lib.load().adn_edge_loss_workspace_bytes(B, H, W)


def compute_errors(gt, pred, out7):
    """gt/pred [samples, pixels] f32 -> out7 [samples, 7]."""
    _dev(gt, pred, out7)
    _lib.call('adn_compute_errors', ptr(gt), ptr(pred), gt.shape[0], gt.shape[1], ptr(out7), None, 0, _stream())


def frontend(wave, mode, S, antialias, out, workspace):
    _dev(wave, out, workspace)
    B, _, T = wave.shape
    _lib.call('adn_frontend', ptr(wave), B, T, mode, S, int(antialias), ptr(out), ptr(workspace),
              workspace.numel() * workspace.element_size(), _stream())


def resize_bilinear_bwd(gout, H, W, gin):
    """gout f32 [planes, S, S] -> gin f32 [planes, H, W]: adjoint of resize_bilinear(antialias=False)."""
    _dev(gout, gin)
    planes, S, _ = gout.shape
    _lib.call('adn_resize_bilinear_bwd', ptr(gout), planes, H, W, S, ptr(gin), _stream())


def clamp_range(x, max_depth, out, g=None):
    """out = clamp(x, 0, max_depth), or with ``g``: g masked to 0 <= x <= max_depth (the clamp's backward)."""
    _dev(x, out, g)
    _lib.call('adn_clamp_range', ptr(x), ptr(g), x.numel(), float(max_depth), ptr(out), _stream())


def frontend_workspace_bytes(B, T, mode):
    return _lib.load().adn_frontend_workspace_bytes(B, T, mode)


def resize_bilinear(src, S, antialias, out):
    """src f32 [planes, H, W] -> out f32 [planes, S, S] (align_corners=False, optional antialias)."""
    _dev(src, out)
    planes, H, W = src.shape
    _lib.call('adn_resize_bilinear', ptr(src), planes, H, W, S, int(antialias), ptr(out), _stream())


# ---- DoubleConv U-Net family (csrc/dcnet.hip) -------------------------------------------------------
def maxpool2_fwd(src, dst):
    """src [B,H,W,C] -> dst [B,H//2,W//2,C] (nn.MaxPool2d(2))."""
    B, H, W, Cc = src.shape
    _dev(src, dst)
    _lib.call('adn_maxpool2_fwd', ptr(src), ptr(dst), B, H, W, Cc, dtype_code(src.dtype), _stream())


def maxpool2_fwd_mx8(src, dst, out8, out_scales):
    """maxpool2_fwd (bf16) that also writes the MX-fp8 copy of dst."""
    B, H, W, Cc = src.shape
    _dev(src, dst, out8, out_scales)
    _lib.call('adn_maxpool2_fwd_mx8', ptr(src), ptr(dst), B, H, W, Cc, ptr(out8), ptr(out_scales), _stream())


def upsample2x_fwd_mx8(src, dst, out8, out_scales):
    """upsample2x_fwd (bf16) that also writes the MX-fp8 copy of dst."""
    B, Hi, Wi, Cc = src.shape
    _dev(src, dst, out8, out_scales)
    _lib.call('adn_upsample2x_fwd_mx8', ptr(src), ptr(dst), B, Hi, Wi, dst.shape[1], dst.shape[2], Cc, ptr(out8),
              ptr(out_scales), _stream())


def maxpool2_bwd(gdst, y, gsrc, accumulate):
    B, H, W, Cc = y.shape
    _dev(gdst, y, gsrc)
    _lib.call('adn_maxpool2_bwd', ptr(gdst), ptr(y), ptr(gsrc), B, H, W, Cc, int(bool(accumulate)),
              dtype_code(y.dtype), _stream())


def tail_stats_blocks(work, Cc):
    """Partial rows the tail-fused backward kernels write for ``work`` thread items (0: this channel count is not fusable)."""
    return _lib.load().adn_tail_stats_blocks(work, Cc)


def maxpool2_bwd_tail(gdst, y, gsrc, accumulate, z, mean, istd, partials):
    """maxpool2_bwd that is the last writer of gsrc: + ReLU mask of y and BN-backward partial sums."""
    B, H, W, Cc = y.shape
    _dev(gdst, y, gsrc, z, mean, istd, partials)
    _lib.call('adn_maxpool2_bwd_tail', ptr(gdst), ptr(y), ptr(gsrc), B, H, W, Cc, int(bool(accumulate)), dtype_code(y.dtype),
              ptr(z), ptr(mean), ptr(istd), ptr(partials), _stream())


def upsample2x_bwd_tail(gdst, gsrc, accumulate, y, z, mean, istd, partials):
    B, Hi, Wi, Cc = gsrc.shape
    _dev(gdst, gsrc, y, z, mean, istd, partials)
    _lib.call('adn_upsample2x_bwd_tail', ptr(gdst), ptr(gsrc), B, Hi, Wi, gdst.shape[1], gdst.shape[2], Cc,
              int(bool(accumulate)), dtype_code(gsrc.dtype), ptr(y), ptr(z), ptr(mean), ptr(istd), ptr(partials), _stream())


def upsample2x_fwd(src, dst):
    """Bilinear x2 (align_corners=True) of src [B,Hi,Wi,C], zero padded into dst [B,Ho,Wo,C]."""
    B, Hi, Wi, Cc = src.shape
    _dev(src, dst)
    _lib.call('adn_upsample2x_fwd', ptr(src), ptr(dst), B, Hi, Wi, dst.shape[1], dst.shape[2], Cc,
              dtype_code(src.dtype), _stream())


def upsample2x_bwd(gdst, gsrc, accumulate=False):
    B, Hi, Wi, Cc = gsrc.shape
    _dev(gdst, gsrc)
    _lib.call('adn_upsample2x_bwd', ptr(gdst), ptr(gsrc), B, Hi, Wi, gdst.shape[1], gdst.shape[2], Cc,
              int(bool(accumulate)), dtype_code(gsrc.dtype), _stream())


def pixel_shuffle2(packed, spatial, inverse=False):
    """packed [B,H,W,4*C] (tap-major) <-> spatial [B,2H,2W,C]: the scatter of ConvTranspose2d(k 2, s 2) / the gather of
    its gradient (inverse)."""
    B, H, W, C4 = packed.shape
    assert spatial.shape == (B, 2 * H, 2 * W, C4 // 4) and packed.dtype == spatial.dtype
    _dev(packed, spatial)
    src, dst = (spatial, packed) if inverse else (packed, spatial)
    _lib.call('adn_pixel_shuffle2', ptr(src), ptr(dst), B, H, W, C4 // 4, int(bool(inverse)), dtype_code(packed.dtype),
              _stream())


def relu_bwd_stats_num_partials(pixels, Cc):
    return _lib.load().adn_relu_bwd_stats_num_partials(pixels, Cc)


def relu_bwd_stats(g, y, z, mean, istd, pixels, Cc, partials):
    _dev(g, y, z, mean, istd, partials)
    _lib.call('adn_relu_bwd_stats', ptr(g), ptr(y), ptr(z), ptr(mean), ptr(istd), pixels, Cc, dtype_code(g.dtype),
              ptr(partials), _stream())


def head1x1_fwd(x, w, bias, act, max_depth, zpre, out):
    """x [..., C] NHWC, w f32 [C], bias f32 [1] or None -> zpre/out f32 [pixels]."""
    Cc = x.shape[-1]
    _dev(x, w, bias, zpre, out)
    _lib.call('adn_head1x1_fwd', ptr(x), ptr(w), ptr(bias), x.numel() // Cc, Cc, dtype_code(x.dtype), act,
              float(max_depth), ptr(zpre), ptr(out), _stream())


def head1x1_bwd_workspace_bytes(pixels, Cc):
    return _lib.load().adn_head1x1_bwd_workspace_bytes(pixels, Cc)


def head1x1_bwd(gout, zpre, x, w, act, max_depth, gx, dw, db, workspace):
    Cc = x.shape[-1]
    _dev(gout, zpre, x, w, gx, dw, db, workspace)
    _lib.call('adn_head1x1_bwd', ptr(gout), ptr(zpre), ptr(x), ptr(w), x.numel() // Cc, Cc, dtype_code(x.dtype), act,
              float(max_depth), ptr(gx), ptr(dw), ptr(db), ptr(workspace),
              workspace.numel() * workspace.element_size(), _stream())


def l1tv_workspace_bytes(n):
    return _lib.load().adn_l1tv_workspace_bytes(n)


def l1tv_stats(pred, gt, stats, workspace):
    B, H, W = pred.shape[0], pred.shape[-2], pred.shape[-1]
    _dev(pred, gt, stats, workspace)
    _lib.call('adn_l1tv_stats', ptr(pred), ptr(gt), B, H, W, ptr(stats), ptr(workspace),
              workspace.numel() * workspace.element_size(), _stream())


def l1tv_finish(pred, gt, stats, replicas, lambda_l1, lambda_smooth, loss_out, grad):
    B, H, W = pred.shape[0], pred.shape[-2], pred.shape[-1]
    _dev(pred, gt, stats, loss_out, grad)
    _lib.call('adn_l1tv_finish', ptr(pred), ptr(gt), B, H, W, ptr(stats), int(replicas), float(lambda_l1),
              float(lambda_smooth), ptr(loss_out), ptr(grad), _stream())


# ---- binaural cross-attention (csrc/attn.hip, attn_mfma.hip) -----------------------------------------------
def nchw_slice_to_nhwc(src, c_lo, Cc, dst):
    """Channels [c_lo, c_lo + Cc) of src f32 [B,Ct,H,W] -> dst [B,H,W,Cpad] (extra channels zero)."""
    B, Ct, H, W = src.shape
    _dev(src, dst)
    _lib.call('adn_nchw_slice_to_nhwc', ptr(src), ptr(dst), B, Ct, c_lo, Cc, dst.shape[-1], H, W, dtype_code(dst.dtype),
              _stream())


def _esz(t):
    return t.element_size()


def _attn_desc(q, k, v, o, lse, dqk, dv, kv_shift, scale, dout=None, dq=None, dk=None, dvg=None, workspace=None):
    """q/k/v/o (and gradients) are [B2, N, *] views whose last-dim stride is 1 and row stride = stride(1)."""
    d = AdnAttnDesc()
    B2, N = q.shape[0], q.shape[1]
    d.dtype, d.B2, d.N, d.dqk, d.dv, d.kv_shift = dtype_code(q.dtype), B2, N, dqk, dv, kv_shift
    for t in (q, k, v, o, dout, dq, dk, dvg):
        if t is not None:
            assert t.stride(-1) == 1 and t.stride(0) == N * t.stride(1), 'attention operands must be [B2,N,ld] row views'
    _dev(q, k, v, o, lse, dout, dq, dk, dvg, workspace)
    d.q, d.k, d.v, d.o, d.lse = ptr(q), ptr(k), ptr(v), ptr(o), ptr(lse)
    d.ld_q, d.ld_k, d.ld_v, d.ld_o = q.stride(1), k.stride(1), v.stride(1), o.stride(1)
    d.scale = float(scale)
    if dout is not None:
        d.dout, d.dq, d.dk, d.dvp = ptr(dout), ptr(dq), ptr(dk), ptr(dvg)
        d.ld_do, d.ld_dq, d.ld_dk, d.ld_dv = dout.stride(1), dq.stride(1), dk.stride(1), dvg.stride(1)
        d.workspace = ptr(workspace)
        d.workspace_bytes = workspace.numel() * workspace.element_size()
    return d


def attn_fwd(q, k, v, o, lse, dqk, dv, kv_shift, scale):
    d = _attn_desc(q, k, v, o, lse, dqk, dv, kv_shift, scale)
    _lib.call('adn_attn_fwd', C.byref(d), _stream())
    _lib.annotate(label='attn_fwd', flops=2.0 * d.B2 * d.N * d.N * (dqk + dv))


def attn_bwd(q, k, v, o, lse, dqk, dv, kv_shift, scale, dout, dq, dk, dvg, workspace):
    d = _attn_desc(q, k, v, o, lse, dqk, dv, kv_shift, scale, dout, dq, dk, dvg, workspace)
    _lib.call('adn_attn_bwd', C.byref(d), _stream())
    _lib.annotate(label='attn_bwd', flops=2.0 * d.B2 * d.N * d.N * (3 * dqk + 2 * dv))


def channel_sum_workspace_bytes(rows, Cc):
    return _lib.load().adn_channel_sum_workspace_bytes(rows, Cc)


def channel_sum(x, rows, Cc, ld, out, workspace):
    _dev(x, out, workspace)
    _lib.call('adn_channel_sum', ptr(x), rows, Cc, ld, dtype_code(x.dtype), ptr(out), ptr(workspace),
              workspace.numel() * workspace.element_size(), _stream())


def gate_bwd(t, att, gamma, gsum, bias, Cc, dgamma, dbias, dw, workspace):
    _dev(t, att, gamma, gsum, bias, dgamma, dbias, dw, workspace)
    _lib.call('adn_gate_bwd', ptr(t), ptr(att), t.numel(), dtype_code(t.dtype), ptr(gamma), ptr(gsum), ptr(bias), Cc,
              ptr(dgamma), ptr(dbias), ptr(dw), dw.numel() if dw is not None else 0, ptr(workspace),
              workspace.numel() * workspace.element_size(), _stream())


# ---- AdaBins distillation model (csrc/adabins.hip) -----------------------------------------------------------
def _nbytes(t):
    return t.numel() * t.element_size()


def pool_workspace_bytes(B, HW, Cc, nq=1):
    return _lib.load().adn_pool_workspace_bytes(B, HW, Cc, nq)


def pool(x, y, B, HW, Cc, nq, scale, out, workspace):
    """x (and y) [B,HW,Cc] NHWC views -> out f32 [B,Cc] (nq=1: scale*sum x) or [B,3,Cc] (sum x^2, y^2, x*y)."""
    _dev(x, y, out, workspace)
    _lib.call('adn_pool', ptr(x), ptr(y), B, HW, Cc, x.shape[-1], nq, dtype_code(x.dtype), float(scale), ptr(out),
              ptr(workspace), _nbytes(workspace), _stream())


def binpred_fwd(g, W1, b1, W2, b2, mask, drop_p, max_depth, h1, widths, centers):
    B, Cb = g.shape
    Hd, nb = W1.shape[0], W2.shape[0]
    _dev(g, W1, b1, W2, b2, mask, h1, widths, centers)
    _lib.call('adn_binpred_fwd', ptr(g), ptr(W1), ptr(b1), ptr(W2), ptr(b2), ptr(mask), float(drop_p), float(max_depth),
              B, Cb, Hd, nb, ptr(h1), ptr(widths), ptr(centers), _stream())


def binpred_bwd(dcent, widths, h1, g, W1, W2, has_mask, drop_p, max_depth, dW2p, db2p, dW1p, db1p, dg):
    B, Cb = g.shape
    Hd, nb = W1.shape[0], W2.shape[0]
    _dev(dcent, widths, h1, g, W1, W2, dW2p, db2p, dW1p, db1p, dg)
    _lib.call('adn_binpred_bwd', ptr(dcent), ptr(widths), ptr(h1), ptr(g), ptr(W1), ptr(W2), int(bool(has_mask)),
              float(drop_p), float(max_depth), B, Cb, Hd, nb, ptr(dW2p), ptr(db2p), ptr(dW1p), ptr(db1p), ptr(dg), _stream())


def dropout_mask(mask, p, seed, counter=None):
    _dev(mask, counter)
    _lib.call('adn_dropout_mask', ptr(mask), mask.numel(), float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, ptr(counter),
              _stream())


def bcast_add(gx, dg, scale, accumulate):
    B, H, W, Cc = gx.shape
    _dev(gx, dg)
    _lib.call('adn_bcast_add', ptr(gx), ptr(dg), B, H * W, Cc, float(scale), int(bool(accumulate)), dtype_code(gx.dtype),
              _stream())


def bins_fwd(logits, centers, base):
    B, H, W, nb = logits.shape
    _dev(logits, centers, base)
    _lib.call('adn_bins_fwd', ptr(logits), ptr(centers), B, H * W, nb, dtype_code(logits.dtype), ptr(base), _stream())


def bins_bwd_workspace_bytes(B, HW, nb):
    return _lib.load().adn_bins_bwd_workspace_bytes(B, HW, nb)


def bins_bwd(logits, centers, base, dbase, dmean, dlogits, dcent, workspace):
    B, H, W, nb = logits.shape
    _dev(logits, centers, base, dbase, dmean, dlogits, dcent, workspace)
    _lib.call('adn_bins_bwd', ptr(logits), ptr(centers), ptr(base), ptr(dbase), ptr(dmean), B, H * W, nb,
              dtype_code(logits.dtype), ptr(dlogits), ptr(dcent), 0, ptr(workspace), _nbytes(workspace), _stream())


def distill_pix_stats(base, resid, gt, teacher, max_depth, final_out, stats, workspace):
    _dev(base, resid, gt, teacher, final_out, stats, workspace)
    _lib.call('adn_distill_pix_stats', ptr(base), ptr(resid), ptr(gt), ptr(teacher), base.numel(), float(max_depth),
              ptr(final_out), ptr(stats), ptr(workspace), _nbytes(workspace), _stream())


def distill_pix_grad(base, resid, gt, teacher, max_depth, stats, lt, lr, ls, dbase, dres):
    _dev(base, resid, gt, teacher, stats, dbase, dres)
    _lib.call('adn_distill_pix_grad', ptr(base), ptr(resid), ptr(gt), ptr(teacher), base.numel(), float(max_depth),
              ptr(stats), float(lt), float(lr), float(ls), ptr(dbase), ptr(dres), _stream())


def featcos_grad(a, r, stats, coef, ga):
    B, H, W, Cc = a.shape
    _dev(a, r, stats, ga)
    _lib.call('adn_featcos_grad', ptr(a), ptr(r), ptr(stats), B, H * W, Cc, dtype_code(a.dtype), float(coef), ptr(ga),
              _stream())


def distill_small(mean_s, mean_t, cent_s, cent_t, feat_stats, feat_channels, pix_stats, temperature, lambdas, terms,
                  dmean, dcent):
    """lambdas = (task, response, feature, bin, sparse); feat_stats: 5 tensors [B,3,C] or None (no teacher)."""
    d = AdnDistillSmall()
    has_t = mean_t is not None
    _dev(mean_s, mean_t, cent_s, cent_t, pix_stats, terms, dmean, dcent)
    d.mean_student, d.mean_teacher, d.centers_student, d.centers_teacher = ptr(mean_s), ptr(mean_t), ptr(cent_s), ptr(cent_t)
    for i in range(5):
        d.feat_stats[i] = ptr(feat_stats[i]) if has_t else None
        d.feat_channels[i] = feat_channels[i] if has_t else 0
    d.pix_stats = ptr(pix_stats)
    d.B, d.nb, d.has_teacher = mean_s.shape[0], mean_s.shape[1], int(has_t)
    d.temperature = float(temperature)
    (d.lambda_task, d.lambda_response, d.lambda_feature, d.lambda_bin, d.lambda_sparse) = [float(v) for v in lambdas]
    d.terms, d.dmean, d.dcent = ptr(terms), ptr(dmean), ptr(dcent)
    _lib.call('adn_distill_small', C.byref(d), _stream())


def resize_nearest(src, S):
    """src f32 [..., H, W] -> new tensor [..., S, S], F.interpolate(mode='nearest')."""
    _dev(src)
    src = src.contiguous()
    out = torch.empty(src.shape[:-2] + (S, S), dtype=torch.float32, device=src.device)
    planes = src.numel() // (src.shape[-1] * src.shape[-2])
    _lib.call('adn_resize_nearest', ptr(src), planes, src.shape[-2], src.shape[-1], S, ptr(out), _stream())
    return out


def resize_nearest_bwd(gout, H, W):
    """gout f32 [..., S, S] -> gradient of the nearest-resized source, new tensor [..., H, W]."""
    _dev(gout)
    gout = gout.contiguous()
    S = gout.shape[-1]
    out = torch.empty(gout.shape[:-2] + (H, W), dtype=torch.float32, device=gout.device)
    planes = gout.numel() // (S * S)
    _lib.call('adn_resize_nearest_bwd', ptr(gout), planes, H, W, S, ptr(out), _stream())
    return out


def image_prepare(src, S, out):
    """src uint8 [B,H,W,3] BGR (decoded camera frames) -> out f32 [B,3,S,S] RGB in [0,1] (cv2 INTER_LINEAR resize)."""
    _dev(src, out)
    if src.dtype != torch.uint8 or src.dim() != 4 or src.shape[-1] != 3:
        raise TypeError('image_prepare: src must be uint8 [B,H,W,3]')
    B, H, W, _ = src.shape
    _lib.call('adn_image_prepare', ptr(src), B, H, W, S, ptr(out), _stream())


def depth_prepare(src, S, max_depth, norm, out):
    """Raw depth maps in mm [planes,H,W] (float32 / uint16 / int32) -> metres, cleaned, clipped, nearest-resized."""
    code = {torch.float32: 0, torch.uint16: 1, torch.int32: 2}[src.dtype]
    planes, H, W = src.shape
    _dev(src, out)
    _lib.call('adn_depth_prepare', ptr(src), code, planes, H, W, S, float(max_depth or 0.0), float(norm or 0.0), ptr(out),
              _stream())


# ---- Base + Residual model (csrc/baseres.hip) --------------------------------------------------------------------
def lowpass_workspace_bytes(B, H, W, k):
    return _lib.load().adn_lowpass_workspace_bytes(B, H, W, k)


def lowpass(gt, k, out, workspace):
    """gt f32 [B,1,H,W] -> avg_pool2d(k, 1, k//2) + bilinear resize back to H x W."""
    B, H, W = gt.shape[0], gt.shape[-2], gt.shape[-1]
    _dev(gt, out, workspace)
    _lib.call('adn_lowpass', ptr(gt), B, H, W, k, ptr(out), ptr(workspace), _nbytes(workspace), _stream())


def clamp_add(a, b, max_depth, out):
    _dev(a, b, out)
    _lib.call('adn_clamp_add', ptr(a), ptr(b), a.numel(), float(max_depth), ptr(out), _stream())


def baseres_stats(base, resid, strct, gt, recon, lrecon, lbase, lsparse, stats, terms, workspace):
    _dev(base, resid, strct, gt, recon, stats, terms, workspace)
    _lib.call('adn_baseres_stats', ptr(base), ptr(resid), ptr(strct), ptr(gt), base.numel(), ptr(recon), float(lrecon),
              float(lbase), float(lsparse), ptr(stats), ptr(terms), ptr(workspace), _nbytes(workspace), _stream())


def baseres_grad(base, resid, strct, gt, gfinal, max_depth, stats, lbase, lsparse, dbase, dres):
    _dev(base, resid, strct, gt, gfinal, stats, dbase, dres)
    _lib.call('adn_baseres_grad', ptr(base), ptr(resid), ptr(strct), ptr(gt), ptr(gfinal), base.numel(), float(max_depth),
              ptr(stats), float(lbase), float(lsparse), ptr(dbase), ptr(dres), _stream())
