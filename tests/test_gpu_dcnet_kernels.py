"""GPU parity of the DoubleConv-family kernels (S1 implicit GEMM + csrc/dcnet.hip) against torch-CPU fp32.

Reference ops: DoubleConv / Down / Up (/root/reference/models/rgb_depth_model.py:21-77 and the identical copies in
binaural_attention_model.py:22-78, adabins_distillation_model.py:27-82), the 1x1 depth head + clamp / sigmoid
(rgb_depth_model.py:195-209, binaural_attention_model.py:330-337) and DepthLoss (train_rgb_depth.py:43-87).
Tolerances as in test_gpu_kernels.py: inputs are pre-rounded to the storage dtype, the reference is fp32 on the
same rounded values; f32 outputs <= 2e-5 (f32 path) / 1e-4 (bf16 path) of max|ref|, bf16 outputs <= 6e-3.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def K():
    from audio_depth_estimation_amd import kernels
    return kernels


def rel_err(a, b):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def rounded(x, dtype):
    return x.to(dtype).float()


def nhwc(x, dtype):
    return x.permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV)


def from_nhwc(y):
    return y.float().cpu().permute(0, 3, 1, 2)


TOL_F32_OUT = {torch.float32: 2e-5, torch.bfloat16: 1e-4}
TOL_T_OUT = {torch.float32: 2e-5, torch.bfloat16: 6e-3}
DTYPES = [torch.float32, torch.bfloat16]

# (B, C0, C1, N, H, W, ks)
S1_SHAPES = [
    (2, 64, 0, 128, 16, 16, 3),     # MFMA wide, BN=128
    (2, 64, 64, 64, 8, 16, 3),      # two gathered sources (virtual concat), non-square
    (3, 128, 0, 128, 2, 2, 3),      # tiny grid: every tap partly padded; split-K
    (8, 64, 0, 128, 32, 32, 3),     # fused LDS epilogue
    (4, 128, 128, 64, 64, 64, 3),   # 256-row tiles
    (2, 64, 64, 64, 64, 64, 3),     # RGBDepthNet up4.conv1 at 64x64 (split-K, two segments in the dgrad)
    (2, 128, 0, 64, 32, 32, 3),     # up3.conv2
    (2, 64, 128, 128, 16, 16, 3),   # AdaBins up4.conv1: C = 192, column tiles straddle taps AND sources
    (2, 192, 0, 128, 8, 8, 3),      # AdaBins up3.conv2: single source, C = 192
    (2, 96, 0, 64, 16, 16, 3),      # 96 channels: per-lane taps in the forward loader and in wgrad
    (3, 128, 256, 192, 6, 10, 3),   # AdaBins up3.conv1 (N = 192), non power-of-two image (slow wgrad path)
    (1, 64, 0, 64, 8, 16, 3),       # exactly one 8 x 16 pixel tile (patch kernels: every border at once)
    (8, 64, 64, 64, 128, 128, 3),   # 512 tiles of 16 x 16 pixels: the tall 4 x 1-wave patch kernel (N = 64), two sources
    (3, 128, 64, 192, 24, 48, 3),   # non power-of-two image of 3 x 3 tiles, N = 192, two sources of different width
    (2, 256, 0, 64, 16, 32, 3),     # 4 channel blocks of the patch-staged wgrad
    (2, 8, 0, 64, 16, 16, 3),       # thin input: narrow loader, K = 72 padded to the K-step
    (2, 16, 0, 64, 7, 9, 3),        # narrow, odd sizes
    (2, 6, 0, 10, 5, 5, 3),         # generic direct path
    (1, 3, 5, 1, 5, 4, 3),          # generic, two sources, one output channel
    (2, 64, 0, 128, 16, 16, 1),     # 1x1 conv (attention projections / fusion layers)
    (2, 128, 64, 64, 8, 8, 1),      # 1x1 two sources
    (2, 6, 0, 10, 5, 5, 1),         # 1x1 generic
]


def s1_operands(w, dtype):
    """[N, Cin, k, k] parameter -> (forward operand [N][rs], input-gradient operand [Cin][rs']) device tensors."""
    k = K()
    N, Cin, ks, _ = w.shape
    taps = ks * ks
    master = w.permute(0, 2, 3, 1).contiguous().to(DEV)          # channels_last memory == [N][taps][Cin]
    fwd = torch.empty(N, k.s1_row_stride(dtype, taps, Cin), dtype=dtype, device=DEV)
    k.pack_rows(master, N, taps, Cin, fwd)
    dg = torch.empty(Cin, k.s1_row_stride(dtype, taps, N), dtype=dtype, device=DEV)
    k.pack_transpose_taps(master, N, taps, Cin, dg, flip=True)
    return fwd, dg


def ws_for(dtype, B, H, W, C0, C1, N, segs, ks):
    k = K()
    P, nbytes = k.igemm_query(dtype, k.GEMM_S1, B, H, W, C0, C1, N, segs, ks=ks)
    return P, torch.empty(max(nbytes, 16) // 4, dtype=torch.float32, device=DEV)


def test_pack_s1_layouts():
    torch.manual_seed(0)
    w = torch.randn(6, 10, 3, 3)
    fwd, dg = s1_operands(w, torch.float32)
    assert fwd.shape == (6, 96) and dg.shape == (10, 64)
    np.testing.assert_array_equal(fwd[:, :90].cpu().numpy(), w.permute(0, 2, 3, 1).reshape(6, 90).numpy())
    assert float(fwd[:, 90:].abs().max()) == 0.0
    ref = w.flip(2, 3).permute(1, 2, 3, 0).reshape(10, 54)        # [Cin][flipped taps][N]
    np.testing.assert_array_equal(dg[:, :54].cpu().numpy(), ref.numpy())
    assert float(dg[:, 54:].abs().max()) == 0.0


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', S1_SHAPES)
def test_conv_forward_s1(dtype, shape):
    """S1 geometry == nn.Conv2d(k, padding=k//2, bias=False) forward (rgb_depth_model.py:29-33)."""
    B, C0, C1, N, H, W, ks = shape
    torch.manual_seed(1)
    x = rounded(torch.randn(B, C0 + C1, H, W), dtype)
    w = rounded(torch.randn(N, C0 + C1, ks, ks) * 0.1, dtype)
    ref = F.conv2d(x, w, padding=ks // 2)
    fwd, _ = s1_operands(w, dtype)
    in0 = nhwc(x[:, :C0], dtype)
    in1 = nhwc(x[:, C0:], dtype) if C1 else None
    out = torch.empty(B, H, W, N, dtype=torch.float32, device=DEV)
    _, ws = ws_for(dtype, B, H, W, C0, C1, N, [N], ks)
    k = K()
    k.igemm(dtype, k.GEMM_S1, B, H, W, in0, in1, fwd, N, k.EPI_RAW, [k.Seg(N, out0=out)], ws, ks=ks)
    assert rel_err(from_nhwc(out), ref) <= TOL_F32_OUT[dtype]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', S1_SHAPES)
def test_conv_z_stats_s1(dtype, shape):
    """Z_STATS epilogue on the S1 geometry: raw conv output in dtype + per-channel sum / sum of squares."""
    B, C0, C1, N, H, W, ks = shape
    torch.manual_seed(2)
    x = rounded(torch.randn(B, C0 + C1, H, W), dtype)
    w = rounded(torch.randn(N, C0 + C1, ks, ks) * 0.1, dtype)
    ref = F.conv2d(x, w, padding=ks // 2)
    fwd, _ = s1_operands(w, dtype)
    in0 = nhwc(x[:, :C0], dtype)
    in1 = nhwc(x[:, C0:], dtype) if C1 else None
    z = torch.empty(B, H, W, N, dtype=dtype, device=DEV)
    P, ws = ws_for(dtype, B, H, W, C0, C1, N, [N], ks)
    part = torch.zeros(P * 2 * N, dtype=torch.float32, device=DEV)
    k = K()
    k.igemm(dtype, k.GEMM_S1, B, H, W, in0, in1, fwd, N, k.EPI_Z_STATS, [k.Seg(N, out0=z, partials=part)], ws, ks=ks)
    assert rel_err(from_nhwc(z), ref) <= TOL_T_OUT[dtype]
    sums = part.view(P, 2, N).double().sum(0).cpu()
    zz = from_nhwc(z).double()                                      # statistics are of the STORED values
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    ref1, ref2 = zz.sum((0, 2, 3)), (zz * zz).sum((0, 2, 3))
    assert float((sums[0] - ref1).abs().max()) <= tol * float(ref1.abs().max() + zz.abs().sum((0, 2, 3)).max() * 1e-3)
    assert float((sums[1] - ref2).abs().max()) <= tol * float(ref2.abs().max())


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', S1_SHAPES)
def test_conv_dgrad_s1(dtype, shape):
    """Input gradient of the stride-1 conv = S1 GEMM of dz with the transposed, tap-flipped operand; the two
    output segments are the two halves of the virtual concat (EPI_ADD, second call accumulates)."""
    B, C0, C1, N, H, W, ks = shape
    torch.manual_seed(3)
    Cin = C0 + C1
    w = rounded(torch.randn(N, Cin, ks, ks) * 0.1, dtype)
    dz = rounded(torch.randn(B, N, H, W), dtype)
    ref = F.conv_transpose2d(dz, w, padding=ks // 2)               # == grad of conv2d wrt its input
    _, dg = s1_operands(w, dtype)
    g0 = torch.empty(B, H, W, C0, dtype=dtype, device=DEV)
    g1 = torch.empty(B, H, W, C1, dtype=dtype, device=DEV) if C1 else None
    k = K()
    segs = [k.Seg(C0, out0=g0)] + ([k.Seg(C1, out0=g1)] if C1 else [])
    _, ws = ws_for(dtype, B, H, W, N, 0, Cin, [C0, C1] if C1 else [C0], ks)
    dzd = nhwc(dz, dtype)
    k.igemm(dtype, k.GEMM_S1, B, H, W, dzd, None, dg, Cin, k.EPI_ADD, segs, ws, ks=ks)
    got = torch.cat([from_nhwc(g0)] + ([from_nhwc(g1)] if C1 else []), 1)
    assert rel_err(got, ref) <= TOL_T_OUT[dtype]
    # accumulate on top of the first result: exactly doubles (up to the storage rounding)
    for s in segs:
        s.accumulate = True
    k.igemm(dtype, k.GEMM_S1, B, H, W, dzd, None, dg, Cin, k.EPI_ADD, segs, ws, ks=ks)
    got2 = torch.cat([from_nhwc(g0)] + ([from_nhwc(g1)] if C1 else []), 1)
    assert rel_err(got2, 2 * ref) <= 2 * TOL_T_OUT[dtype]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', S1_SHAPES)
def test_conv_wgrad_s1(dtype, shape):
    """Weight gradient of the stride-1 conv: dw[n][tap][c] = sum_pix dz[pix][n] * in[pix + tap][c]."""
    B, C0, C1, N, H, W, ks = shape
    torch.manual_seed(4)
    Cin = C0 + C1
    x = rounded(torch.randn(B, Cin, H, W), dtype)
    dz = rounded(torch.randn(B, N, H, W), dtype)
    wz = torch.zeros(N, Cin, ks, ks, requires_grad=True)
    F.conv2d(x, wz, padding=ks // 2).backward(dz)
    ref = wz.grad.permute(0, 2, 3, 1).reshape(N, ks * ks, Cin)
    k = K()
    nb = k.wgrad_workspace_bytes(dtype, B, H, W, N, 0, C0, C1, ks=ks)
    ws = torch.empty(max(nb, 16) // 4, dtype=torch.float32, device=DEV)
    dw = torch.full((N, ks * ks, Cin), float('nan'), dtype=torch.float32, device=DEV)
    in0 = nhwc(x[:, :C0], dtype)
    in1 = nhwc(x[:, C0:], dtype) if C1 else None
    k.wgrad(dtype, B, H, W, nhwc(dz, dtype), None, in0, in1, dw, ws, ks=ks)
    assert rel_err(dw, ref) <= TOL_F32_OUT[dtype]


@pytest.mark.parametrize('dtype', DTYPES)
def test_conv_wgrad_s1_thin_input(dtype):
    """First layer: 3 real input channels zero-padded to one 16-byte chunk, compact [N][9][3] gradient."""
    B, N, H, W, Cin = 2, 64, 16, 16, 3
    torch.manual_seed(5)
    epc = 8 if dtype == torch.bfloat16 else 4
    x = rounded(torch.randn(B, Cin, H, W), dtype)
    dz = rounded(torch.randn(B, N, H, W), dtype)
    wz = torch.zeros(N, Cin, 3, 3, requires_grad=True)
    F.conv2d(x, wz, padding=1).backward(dz)
    ref = wz.grad.permute(0, 2, 3, 1).reshape(N, 9, Cin)
    xp = torch.zeros(B, epc, H, W)
    xp[:, :Cin] = x
    k = K()
    nb = k.wgrad_workspace_bytes(dtype, B, H, W, N, 0, epc, 0, c_valid=Cin, ks=3)
    ws = torch.empty(max(nb, 16) // 4, dtype=torch.float32, device=DEV)
    dw = torch.full((N, 9, Cin), float('nan'), dtype=torch.float32, device=DEV)
    k.wgrad(dtype, B, H, W, nhwc(dz, dtype), None, nhwc(xp, dtype), None, dw, ws, c_valid=Cin, ks=3)
    assert rel_err(dw, ref) <= TOL_F32_OUT[dtype]
    # and the forward of the same layer through the padded operand
    w = rounded(torch.randn(N, Cin, 3, 3) * 0.1, dtype)
    master = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    fwd = torch.empty(N, k.s1_row_stride(dtype, 9, epc), dtype=dtype, device=DEV)
    k.pack_rows(master, N, 9, Cin, fwd, y_pad=epc)
    out = torch.empty(B, H, W, N, dtype=torch.float32, device=DEV)
    _, ws2 = ws_for(dtype, B, H, W, epc, 0, N, [N], 3)
    k.igemm(dtype, k.GEMM_S1, B, H, W, nhwc(xp, dtype), None, fwd, N, k.EPI_RAW, [k.Seg(N, out0=out)], ws2, ks=3)
    assert rel_err(from_nhwc(out), F.conv2d(x, w, padding=1)) <= TOL_F32_OUT[dtype]


# ---------------------------------------------------------------------------------------------------------
EW_SHAPES = [(2, 16, 8, 12), (2, 64, 16, 16), (1, 5, 7, 9), (3, 8, 2, 2), (2, 64, 64, 64)]      # (B, C, H, W)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', EW_SHAPES)
def test_maxpool2(dtype, shape):
    B, C, H, W = shape
    torch.manual_seed(6)
    # post-ReLU-like input with many exact ties (zeros, and a coarse grid of positive values)
    x = rounded((torch.randn(B, C, H, W) * 2).round().clamp(min=0) * 0.5, dtype).requires_grad_(True)
    y = F.max_pool2d(x, 2)
    gy = rounded(torch.randn_like(y), dtype)
    y.backward(gy)
    k = K()
    xd = nhwc(x.detach(), dtype)
    yd = torch.empty(B, H // 2, W // 2, C, dtype=dtype, device=DEV)
    k.maxpool2_fwd(xd, yd)
    np.testing.assert_array_equal(from_nhwc(yd).numpy(), y.detach().numpy())
    gx = torch.full((B, H, W, C), float('nan'), dtype=dtype, device=DEV)
    k.maxpool2_bwd(nhwc(gy, dtype), xd, gx, accumulate=False)
    np.testing.assert_array_equal(from_nhwc(gx).numpy(), x.grad.numpy())     # ties resolved like torch
    base = rounded(torch.randn(B, C, H, W), dtype)
    gx2 = nhwc(base, dtype)
    k.maxpool2_bwd(nhwc(gy, dtype), xd, gx2, accumulate=True)
    assert rel_err(from_nhwc(gx2), base + x.grad) <= TOL_T_OUT[dtype]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(2, 16, 4, 6, 0, 0), (2, 64, 8, 8, 0, 0), (1, 5, 3, 4, 1, 1), (2, 8, 1, 1, 0, 1),
                                   (2, 8, 16, 16, 0, 0), (2, 64, 32, 32, 0, 0), (2, 128, 16, 16, 0, 0)])
def test_upsample2x(dtype, shape):
    """Up: bilinear x2 align_corners=True, then F.pad to the skip size (rgb_depth_model.py:61-75)."""
    B, C, Hi, Wi, dH, dW = shape
    Ho, Wo = 2 * Hi + dH, 2 * Wi + dW
    torch.manual_seed(7)
    x = rounded(torch.randn(B, C, Hi, Wi), dtype).requires_grad_(True)
    up = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True)
    up = F.pad(up, [dW // 2, dW - dW // 2, dH // 2, dH - dH // 2])
    g = rounded(torch.randn_like(up), dtype)
    up.backward(g)
    k = K()
    out = torch.full((B, Ho, Wo, C), float('nan'), dtype=dtype, device=DEV)
    k.upsample2x_fwd(nhwc(x.detach(), dtype), out)
    assert rel_err(from_nhwc(out), up.detach()) <= TOL_T_OUT[dtype]
    gx = torch.full((B, Hi, Wi, C), float('nan'), dtype=dtype, device=DEV)
    k.upsample2x_bwd(nhwc(g, dtype), gx)
    assert rel_err(from_nhwc(gx), x.grad) <= TOL_T_OUT[dtype]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(2, 64, 16, 16), (4, 128, 32, 32), (2, 5, 7, 9), (1, 8, 3, 3), (2, 512, 4, 4),
                                   (2, 64, 32, 32), (2, 64, 64, 64)])
def test_relu_bwd_stats(dtype, shape):
    B, C, H, W = shape
    torch.manual_seed(8)
    z = rounded(torch.randn(B, C, H, W), dtype)
    mean, istd = torch.randn(C) * 0.1, torch.rand(C) + 0.5
    y = rounded(torch.relu((z - mean[None, :, None, None]) * istd[None, :, None, None]), dtype)
    g = rounded(torch.randn(B, C, H, W), dtype)
    gm = g * (y > 0)
    xh = (z - mean[None, :, None, None]) * istd[None, :, None, None]
    k = K()
    pixels = B * H * W
    P = k.relu_bwd_stats_num_partials(pixels, C)
    part = torch.full((P, 2, C), float('nan'), dtype=torch.float32, device=DEV)
    gd = nhwc(g, dtype)
    k.relu_bwd_stats(gd, nhwc(y, dtype), nhwc(z, dtype), mean.to(DEV), istd.to(DEV), pixels, C, part)
    np.testing.assert_array_equal(from_nhwc(gd).numpy(), gm.numpy())
    sums = part.double().sum(0).cpu()
    ref1, ref2 = gm.double().sum((0, 2, 3)), (gm * xh).double().sum((0, 2, 3))
    scale = float(gm.abs().sum((0, 2, 3)).max())
    assert float((sums[0] - ref1).abs().max()) <= 1e-5 * scale
    assert float((sums[1] - ref2).abs().max()) <= 1e-5 * scale * float(xh.abs().max())


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('act', [0, 1])
@pytest.mark.parametrize('shape', [(2, 64, 16, 16), (2, 8, 5, 7), (1, 5, 4, 4), (2, 256, 4, 4)])
def test_head1x1(dtype, act, shape):
    """outc (1x1 -> 1 channel) + clamp(0, max) / sigmoid * max: forward, input grad, weight and bias grads."""
    B, C, H, W = shape
    maxd = 30.0
    torch.manual_seed(9)
    x = rounded(torch.randn(B, C, H, W), dtype).requires_grad_(True)
    w = (torch.randn(1, C, 1, 1) * (6.0 if act == 0 else 0.3)).requires_grad_(True)     # act 0: reach both clamps
    b = torch.tensor([2.0 if act == 0 else 0.1], requires_grad=True)
    zz = F.conv2d(x, w, b)
    out = torch.clamp(zz, 0, maxd) if act == 0 else torch.clamp(torch.sigmoid(zz) * maxd, 0, maxd)
    gout = torch.randn_like(out)
    out.backward(gout)
    k = K()
    pixels = B * H * W
    xd = nhwc(x.detach(), dtype)
    wd, bd = w.detach().reshape(C).to(DEV), b.detach().to(DEV)
    zpre = torch.empty(pixels, dtype=torch.float32, device=DEV)
    o = torch.empty(pixels, dtype=torch.float32, device=DEV)
    k.head1x1_fwd(xd, wd, bd, act, maxd, zpre, o)
    assert rel_err(o.view(B, 1, H, W), out.detach()) <= 1e-5
    gx = torch.full((B, H, W, C), float('nan'), dtype=dtype, device=DEV)
    dw = torch.full((C,), float('nan'), dtype=torch.float32, device=DEV)
    db = torch.full((1,), float('nan'), dtype=torch.float32, device=DEV)
    ws = torch.empty(k.head1x1_bwd_workspace_bytes(pixels, C) // 4, dtype=torch.float32, device=DEV)
    k.head1x1_bwd(gout.reshape(-1).to(DEV), zpre, xd, wd, act, maxd, gx, dw, db, ws)
    assert rel_err(from_nhwc(gx), x.grad) <= TOL_T_OUT[dtype]
    assert rel_err(dw, w.grad.reshape(C)) <= 2e-5
    assert rel_err(db, b.grad) <= 2e-5


@pytest.mark.parametrize('shape', [(2, 16, 16), (3, 7, 9), (4, 64, 64)])
@pytest.mark.parametrize('replicas', [1, 2])
def test_l1tv_loss(shape, replicas):
    """DepthLoss (train_rgb_depth.py:43-87): value and d loss / d pred, incl. sign(0) = 0 ties."""
    B, H, W = shape
    torch.manual_seed(10)
    pred = (torch.rand(B, 1, H, W) * 8).round().div(2).requires_grad_(True)     # coarse grid: many zero differences
    gt = (torch.rand(B, 1, H, W) * 8).round().div(2)
    l1 = F.l1_loss(pred, gt)
    sm = (pred[:, :, :, :-1] - pred[:, :, :, 1:]).abs().mean() + (pred[:, :, :-1, :] - pred[:, :, 1:, :]).abs().mean()
    loss = 1.0 * l1 + 0.1 * sm
    loss.backward()
    k = K()
    pd, gd = pred.detach().to(DEV), gt.to(DEV)
    stats = torch.zeros(4, dtype=torch.float64, device=DEV)
    ws = torch.empty(k.l1tv_workspace_bytes(B * H * W) // 8, dtype=torch.float64, device=DEV)
    k.l1tv_stats(pd, gd, stats, ws)
    stats *= replicas                                           # what the all-reduce over identical replicas gives
    lo = torch.zeros(1, dtype=torch.float32, device=DEV)
    grad = torch.full_like(pd, float('nan'))
    k.l1tv_finish(pd, gd, stats, replicas, 1.0, 0.1, lo, grad)
    assert abs(float(lo) - float(loss)) <= 1e-6 * abs(float(loss))
    assert rel_err(grad * replicas, pred.grad) <= 1e-5


# ---- binaural cross-attention ----------------------------------------------------------------------------
def _attn_ref(q, k, v, B, scale):
    """[2B,N,d] stacked [left; right]: entry b attends to entry (b + B) % 2B (binaural_attention_model.py:114-127)."""
    kk, vv = torch.roll(k, -B, 0), torch.roll(v, -B, 0)
    s = torch.einsum('bid,bjd->bij', q, kk) * scale
    p = torch.softmax(s, -1)
    return torch.einsum('bij,bjc->bic', p, vv), torch.logsumexp(s, -1)


# (B, N, dqk, dv)
ATTN_SHAPES = [(1, 16, 2, 16), (2, 100, 4, 32), (2, 256, 8, 64), (1, 300, 16, 128), (1, 128, 64, 512),
               # full-width head dims, N % 128 == 0: the bf16 runs take the MFMA kernels (attn_mfma.hip)
               (1, 256, 16, 128), (2, 1024, 16, 128), (2, 128, 32, 256), (1, 512, 32, 256), (1, 64, 64, 512),
               (2, 192, 64, 512)]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', ATTN_SHAPES)
def test_attention_fwd_bwd(dtype, shape):
    """Streaming-softmax attention vs the explicit softmax(QK^T)V of the reference, forward and backward;
    q | k | v live in ONE fused projection buffer (row stride = dqk + dqk + dv + padding)."""
    B, N, dqk, dv = shape
    B2 = 2 * B
    torch.manual_seed(11)
    ld = dqk * 2 + dv + 8
    qkv = rounded(torch.randn(B2, N, ld), dtype)
    scale = 1.0 / (dv ** 0.5)
    q = (qkv[:, :, :dqk] * 2).clone().requires_grad_(True)
    k = (qkv[:, :, dqk:2 * dqk] * 2).clone().requires_grad_(True)
    v = qkv[:, :, 2 * dqk:2 * dqk + dv].clone().requires_grad_(True)
    qkv = torch.cat([q.detach(), k.detach(), v.detach(), qkv[:, :, 2 * dqk + dv:]], -1)
    o_ref, lse_ref = _attn_ref(q, k, v, B, scale)
    do = rounded(torch.randn_like(o_ref), dtype)
    o_ref.backward(do)
    kk = K()
    dev_qkv = qkv.to(dtype).to(DEV)
    qd, kd, vd = dev_qkv[:, :, :dqk], dev_qkv[:, :, dqk:2 * dqk], dev_qkv[:, :, 2 * dqk:2 * dqk + dv]
    o = torch.full((B2, N, dv), float('nan'), dtype=dtype, device=DEV)
    lse = torch.empty(B2, N, dtype=torch.float32, device=DEV)
    kk.attn_fwd(qd, kd, vd, o, lse, dqk, dv, B, scale)
    tol = TOL_T_OUT[dtype]
    assert rel_err(o, o_ref) <= tol
    assert float((lse.cpu() - lse_ref.detach()).abs().max()) <= 1e-4
    dqkv = torch.zeros(B2, N, ld, dtype=dtype, device=DEV)
    ws = torch.empty(B2 * N, dtype=torch.float32, device=DEV)
    # backward consumes the STORED (rounded) forward output, as the engine does
    kk.attn_bwd(qd, kd, vd, o, lse, dqk, dv, B, scale, do.to(dtype).to(DEV), dqkv[:, :, :dqk], dqkv[:, :, dqk:2 * dqk],
                dqkv[:, :, 2 * dqk:2 * dqk + dv], ws)
    btol = 2e-4 if dtype == torch.float32 else 2e-2
    assert rel_err(dqkv[:, :, :dqk], q.grad) <= btol
    assert rel_err(dqkv[:, :, dqk:2 * dqk], k.grad) <= btol
    assert rel_err(dqkv[:, :, 2 * dqk:2 * dqk + dv], v.grad) <= btol
    assert float(dqkv[:, :, 2 * dqk + dv:].float().abs().max()) == 0.0          # padding untouched


@pytest.mark.parametrize('dtype', DTYPES)
def test_channel_sum_and_gate_bwd(dtype):
    """Bias gradient (column sums) and the backward of x + gamma * (W att + b)."""
    torch.manual_seed(12)
    rows, C = 1000, 24
    x = rounded(torch.randn(rows, C + 8), dtype)
    kk = K()
    out = torch.empty(C, dtype=torch.float32, device=DEV)
    ws = torch.empty(kk.channel_sum_workspace_bytes(rows, C) // 4, dtype=torch.float32, device=DEV)
    kk.channel_sum(x.to(dtype).to(DEV), rows, C, C + 8, out, ws)
    assert rel_err(out, x[:, :C].sum(0)) <= 1e-5
    # gate: y = x + gamma * (att @ W^T + b);  given G = dL/dy
    att = rounded(torch.randn(rows, C), dtype)
    W = torch.randn(C, C, requires_grad=True)
    b = torch.randn(C, requires_grad=True)
    gamma = torch.tensor([0.37], requires_grad=True)
    att_r = att.clone().requires_grad_(True)
    G = rounded(torch.randn(rows, C), dtype)
    (gamma * (att_r @ W.t() + b)).backward(G)
    t = rounded(G @ W.detach(), dtype)                       # unscaled input gradient of the projection
    td = t.to(dtype).to(DEV)
    gsum = G.sum(0).to(DEV)
    dw = (G.t() @ att).contiguous().to(DEV)                  # wgrad(G, att), unscaled
    dgamma = torch.empty(1, dtype=torch.float32, device=DEV)
    dbias = torch.empty(C, dtype=torch.float32, device=DEV)
    ws2 = torch.empty(1024, dtype=torch.float64, device=DEV)
    kk.gate_bwd(td, att.to(dtype).to(DEV), gamma.detach().to(DEV), gsum, b.detach().to(DEV), C, dgamma, dbias, dw, ws2)
    assert rel_err(dgamma, gamma.grad) <= (1e-5 if dtype == torch.float32 else 5e-3)
    assert rel_err(dbias, b.grad) <= 1e-5
    assert rel_err(dw, W.grad) <= 1e-5
    assert rel_err(td, att_r.grad) <= TOL_T_OUT[dtype] * 2


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('C', [64, 12, 5])
def test_pixel_shuffle2_roundtrip(dtype, C):
    """packed [B,H,W,4,C] <-> spatial [B,2H,2W,C] of ConvTranspose2d(k 2, s 2): exact permutation both ways."""
    B, H, W = 2, 5, 7
    g = torch.Generator().manual_seed(3)
    packed = torch.randn(B, H, W, 4 * C, generator=g).to(dtype).to(DEV)
    spatial = torch.empty(B, 2 * H, 2 * W, C, dtype=dtype, device=DEV)
    K().pixel_shuffle2(packed, spatial)
    want = packed.view(B, H, W, 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(B, 2 * H, 2 * W, C)
    assert torch.equal(spatial, want)
    back = torch.empty_like(packed)
    K().pixel_shuffle2(back, spatial, inverse=True)
    assert torch.equal(back, packed)


@pytest.mark.parametrize('H,W,S', [(32, 32, 64), (16, 24, 40), (64, 64, 32), (7, 5, 33), (48, 48, 48)])
def test_resize_bilinear_bwd_is_the_adjoint(H, W, S):
    """adn_resize_bilinear_bwd vs torch autograd of F.interpolate(bilinear, align_corners=False) (up / down / odd)."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, H, W, generator=g)
    go = torch.randn(3, S, S, generator=g)
    xr = x.clone().requires_grad_(True)
    y = F.interpolate(xr[None], size=(S, S), mode='bilinear', align_corners=False)[0]
    y.backward(go)
    out = torch.empty(3, S, S, device=DEV)
    K().resize_bilinear(x.to(DEV), S, False, out)
    assert rel_err(out, y.detach()) <= 2e-6
    gin = torch.empty(3, H, W, device=DEV)
    K().resize_bilinear_bwd(go.to(DEV), H, W, gin)
    assert rel_err(gin, xr.grad) <= 2e-6


def test_clamp_range_forward_backward():
    g = torch.Generator().manual_seed(6)
    x = (40 * torch.rand(5000, generator=g) - 5).to(DEV)
    x[:3] = torch.tensor([0.0, 30.0, 30.000002], device=DEV)
    go = torch.randn(5000, generator=g).to(DEV)
    out, gx = torch.empty_like(x), torch.empty_like(x)
    K().clamp_range(x, 30.0, out)
    K().clamp_range(x, 30.0, gx, g=go)
    xr = x.clone().requires_grad_(True)
    y = torch.clamp(xr, 0, 30.0)
    y.backward(go)
    assert torch.equal(out, y.detach()) and torch.equal(gx, xr.grad)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(2, 64, 16, 24), (3, 128, 9, 7), (1, 256, 4, 4)])
def test_tail_fused_backward_kernels_match_the_two_pass_form(dtype, shape):
    """adn_maxpool2_bwd_tail / adn_upsample2x_bwd_tail (the last gradient writer also applies the ReLU mask and the
    BatchNorm-backward sums) against the plain backward kernel followed by adn_relu_bwd_stats: gradients bit-identical, the
    per-channel sums equal up to the summation order."""
    B, Cc, H, W = shape
    k = K()
    torch.manual_seed(5)
    y = torch.relu(rounded(torch.randn(B, H, W, Cc), dtype)).to(dtype).to(DEV)          # forward output (post-ReLU)
    z = rounded(torch.randn(B, H, W, Cc), dtype).to(dtype).to(DEV)
    mean, istd = torch.randn(Cc, device=DEV), torch.rand(Cc, device=DEV) + 0.5
    old = rounded(torch.randn(B, H, W, Cc) * 0.1, dtype).to(dtype).to(DEV)
    sums = lambda part, rows: part.view(rows, 2, Cc).double().sum(0)
    # (bf16: the fused kernels sum the f32 value BEFORE it is rounded for storage, like the dgrad epilogue does; the
    #  two-pass form sums the stored bf16 values -- rounding noise of 2^-9 per element, averaged)
    stol = 1e-4 if dtype == torch.float32 else 4e-3
    # ---- max-pool (odd sizes: trailing row / column), accumulating and not
    gd = rounded(torch.randn(B, H // 2, W // 2, Cc), dtype).to(dtype).to(DEV)
    for acc in (True, False):
        g0, g1 = old.clone(), old.clone()
        k.maxpool2_bwd(gd, y, g0, acc)
        P0 = k.relu_bwd_stats_num_partials(B * H * W, Cc)
        p0 = torch.full((P0 * 2 * Cc,), float('nan'), device=DEV)
        k.relu_bwd_stats(g0, y, z, mean, istd, B * H * W, Cc, p0)
        P1 = k.tail_stats_blocks(B * ((H + 1) // 2) * ((W + 1) // 2) * (Cc // 8), Cc)
        assert P1 > 0
        p1 = torch.full((P1 * 2 * Cc,), float('nan'), device=DEV)
        k.maxpool2_bwd_tail(gd, y, g1, acc, z, mean, istd, p1)
        assert torch.equal(g0, g1)
        ref = sums(p0, P0)
        assert float((sums(p1, P1) - ref).abs().max()) <= stol * float(ref.abs().max()) + 1e-6
    # ---- upsample (target padded by one row / two columns as in Up.forward)
    Ho, Wo = 2 * H + 1, 2 * W + 2
    gu = rounded(torch.randn(B, Ho, Wo, Cc), dtype).to(dtype).to(DEV)
    for acc in (True, False):
        g0, g1 = old.clone(), old.clone()
        k.upsample2x_bwd(gu, g0, acc)
        P0 = k.relu_bwd_stats_num_partials(B * H * W, Cc)
        p0 = torch.full((P0 * 2 * Cc,), float('nan'), device=DEV)
        k.relu_bwd_stats(g0, y, z, mean, istd, B * H * W, Cc, p0)
        P1 = k.tail_stats_blocks(B * H * W * (Cc // 8), Cc)
        p1 = torch.full((P1 * 2 * Cc,), float('nan'), device=DEV)
        k.upsample2x_bwd_tail(gu, g1, acc, y, z, mean, istd, p1)
        assert torch.equal(g0, g1)
        ref = sums(p0, P0)
        assert float((sums(p1, P1) - ref).abs().max()) <= stol * float(ref.abs().max()) + 1e-6
    assert k.tail_stats_blocks(1000, 96) == 0             # 12 channel groups do not divide 256: the engine keeps the two-pass form
