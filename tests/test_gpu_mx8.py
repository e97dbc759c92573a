"""GPU parity of the block-scaled fp8 (MX e4m3) 3 x 3 convolution path (csrc/mx8.hip) against oracle/mx8_oracle.py.

  * adn_mx8_quantize / adn_mx8_pack: BIT-EXACT element bytes and scale bytes (integer / byte work);
  * adn_conv3x3_mx8 against the float64 convolution of the DEQUANTISED operands (the kernel's products are exact; what
    differs is the f32 accumulation order and the bf16 rounding of the stored output): |diff| <= 2^-8 |ref| + 1e-3 max|ref|
    per element, BatchNorm partial sums 1e-3 relative;
  * and, stated and measured, against the UNQUANTISED float64 convolution: relative L2 error of a single conv with both
    operands in MX e4m3 (3 mantissa bits) <= 4e-2 (measured 2.6e-2 .. 2.9e-2 for Gaussian operands).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _bf16(a):
    return torch.from_numpy(np.asarray(a, np.float32)).to(torch.bfloat16)


def test_quantize_bit_exact():
    from audio_depth_estimation_amd import kernels as K
    from oracle import mx8_oracle as mx
    rng = np.random.default_rng(0)
    rows, Cc = 4096, 128
    x = rng.standard_normal((rows, Cc)) * np.exp2(rng.integers(-30, 30, (rows, Cc // 32, 1)).repeat(32, axis=2).reshape(rows, Cc))
    x[5] = 0.0                                            # all-zero blocks
    x[6, :32] = 448.0 * 2.0 ** 3                          # exactly the top of a binade
    x[7, 32:64] = 1.76 * 2.0 ** -3                        # mantissa above 1.75: the scale moves up, nothing saturates
    x[8, 64:96] = np.linspace(-1, 1, 32) * 3e38           # near the top of the bf16 range
    x[9, 96:] = np.linspace(-1, 1, 32) * 1e-38            # near the bottom
    xb = _bf16(x)
    want_bits, want_sc, _ = mx.quantize(xb.float().numpy())
    q8 = torch.empty(rows, Cc, dtype=torch.uint8, device=DEV)
    sc = torch.empty(rows, Cc // 32, dtype=torch.uint8, device=DEV)
    K.mx8_quantize(xb.to(DEV), q8, sc)
    np.testing.assert_array_equal(sc.cpu().numpy(), want_sc)
    got = q8.cpu().numpy()
    nz = (want_bits & 0x7f) != 0                          # (sign of a zero: not part of the contract)
    np.testing.assert_array_equal(got[nz], want_bits[nz])
    assert not ((got & 0x7f)[~nz]).any()


@pytest.mark.parametrize('X,Y', [(64, 64), (128, 192)])
@pytest.mark.parametrize('transpose', [False, True])
def test_pack_bit_exact(X, Y, transpose):
    from audio_depth_estimation_amd import kernels as K
    from oracle import mx8_oracle as mx
    rng = np.random.default_rng(1)
    w = (rng.standard_normal((X, Y, 3, 3)) * np.exp2(rng.integers(-6, 2, (X, 1, 1, 1)))).astype(np.float32)
    w[3] = 0.0
    want8, wantsc, _ = mx.pack_weights(w, transpose)
    master = torch.from_numpy(np.ascontiguousarray(w.transpose(0, 2, 3, 1))).to(DEV)      # [X][9][Y]
    s8, ssc = K.mx8_pack_shapes(X, Y, transpose)
    w8 = torch.full(s8, 0xAA, dtype=torch.uint8, device=DEV)
    wsc = torch.full(ssc, 0xAA, dtype=torch.uint8, device=DEV)
    K.mx8_pack(master, X, Y, transpose, w8, wsc)
    np.testing.assert_array_equal(wsc.cpu().numpy(), wantsc)
    got = w8.cpu().numpy()
    nz = (want8 & 0x7f) != 0
    np.testing.assert_array_equal(got[nz], want8[nz])
    assert not ((got & 0x7f)[~nz]).any()


def _quant_dev(x_bf16):
    from audio_depth_estimation_amd import kernels as K
    q8 = torch.empty(x_bf16.shape, dtype=torch.uint8, device=DEV)
    sc = torch.empty(x_bf16.shape[:-1] + (x_bf16.shape[-1] // 32,), dtype=torch.uint8, device=DEV)
    K.mx8_quantize(x_bf16.to(DEV), q8, sc)
    return q8, sc


CASES = [(2, 16, 32, 64, 0, 64), (1, 8, 16, 128, 64, 128), (2, 24, 48, 64, 64, 192), (3, 8, 32, 256, 0, 128),
         (2, 256, 256, 64, 64, 64)]       # the last: 512 tiles of 16 x 16 pixels -> the tall 4 x 1-wave form (N = 64)


@pytest.mark.parametrize('B,H,W,C0,C1,N', CASES)
def test_conv_forward_against_oracle(B, H, W, C0, C1, N):
    from audio_depth_estimation_amd import kernels as K
    from oracle import mx8_oracle as mx
    rng = np.random.default_rng(B * 1000 + N)
    Cin = C0 + C1
    # per-pixel / per-row magnitudes spread over a few binades: the block scales really differ
    x = rng.standard_normal((B, H, W, Cin)) * np.exp2(rng.integers(-3, 4, (B, H, W, 1)))
    w = (rng.standard_normal((N, Cin, 3, 3)) * 0.05 * np.exp2(rng.integers(-2, 3, (N, 1, 1, 1)))).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    xb = _bf16(x)
    x0, x1 = xb[..., :C0].contiguous(), (xb[..., C0:].contiguous() if C1 else None)
    _, _, xd0 = mx.quantize(x0.float().numpy())
    xd = xd0 if not C1 else np.concatenate([xd0, mx.quantize(x1.float().numpy())[2]], axis=-1)
    _, _, wd = mx.pack_weights(w, False)
    ref = mx.conv3x3(xd, wd) + bias                                        # float64, [B,H,W,N]
    exact = mx.conv3x3(xb.double().numpy(), w.transpose(0, 2, 3, 1).reshape(N, 9, Cin).astype(np.float64)) + bias

    q0, s0 = _quant_dev(x0)
    q1, s1 = _quant_dev(x1) if C1 else (None, None)
    s8, ssc = K.mx8_pack_shapes(N, Cin, False)
    w8, wsc = torch.empty(s8, dtype=torch.uint8, device=DEV), torch.empty(ssc, dtype=torch.uint8, device=DEV)
    K.mx8_pack(torch.from_numpy(np.ascontiguousarray(w.transpose(0, 2, 3, 1))).to(DEV), N, Cin, False, w8, wsc)
    P = K.conv3x3_mx8_num_partials(B, H, W, N, C0, C1)
    z = torch.full((B, H, W, N), float('nan'), dtype=torch.bfloat16, device=DEV)
    part = torch.full((P, 2, N), float('nan'), device=DEV)
    K.conv3x3_mx8(B, H, W, q0, s0, q1, s1, w8, wsc, N, K.EPI_Z_STATS,
                  [K.Seg(N, out0=z, partials=part, bias=torch.from_numpy(bias).to(DEV))])
    got = z.float().cpu().numpy().astype(np.float64)
    assert np.isfinite(got).all()
    err = np.abs(got - ref)
    assert (err <= 2.0 ** -8 * np.abs(ref) + 1e-3 * np.abs(ref).max()).all(), float(err.max())
    ps = part.double().cpu().numpy().sum(axis=0)
    zb = z.double().cpu().numpy().reshape(-1, N)
    # the kernel sums the UNROUNDED f32 z: compare with the oracle's sums, bf16-rounding-sized slack
    np.testing.assert_allclose(ps[0], ref.reshape(-1, N).sum(0), rtol=0, atol=4e-3 * np.abs(ref).sum(axis=(0, 1, 2)).max())
    np.testing.assert_allclose(ps[1], (ref.reshape(-1, N) ** 2).sum(0), rtol=1e-2)
    assert zb.shape[0] == B * H * W
    rl2 = float(np.linalg.norm(got - exact) / np.linalg.norm(exact))
    assert rl2 <= 4e-2, rl2                               # both operands in MX e4m3: measured 2.6e-2 .. 2.9e-2


def test_conv_dgrad_epilogues_against_oracle():
    """Input-gradient GEMM (transposed, tap-flipped MX pack of the same weights) with the two epilogues the DoubleConv
    engine uses: ADN_EPI_ADD into two segments (one accumulating) and ADN_EPI_BWD (ReLU mask + BatchNorm-backward sums)."""
    from audio_depth_estimation_amd import kernels as K
    from oracle import mx8_oracle as mx
    rng = np.random.default_rng(7)
    B, H, W, N, C0, C1 = 2, 16, 16, 128, 64, 64           # conv: (C0 + C1) -> N; the gradient GEMM: N -> C0 + C1
    Cin = C0 + C1
    g = _bf16(rng.standard_normal((B, H, W, N)) * np.exp2(rng.integers(-8, -2, (B, H, W, 1))))
    w = (rng.standard_normal((N, Cin, 3, 3)) * 0.05).astype(np.float32)
    _, _, gd = mx.quantize(g.float().numpy())
    _, _, wdt = mx.pack_weights(w, True)                  # [Cin][9][N], flipped
    ref = mx.conv3x3(gd, wdt)                             # d loss / d input, float64 [B,H,W,Cin]
    q, s = _quant_dev(g)
    s8, ssc = K.mx8_pack_shapes(N, Cin, True)
    w8, wsc = torch.empty(s8, dtype=torch.uint8, device=DEV), torch.empty(ssc, dtype=torch.uint8, device=DEV)
    K.mx8_pack(torch.from_numpy(np.ascontiguousarray(w.transpose(0, 2, 3, 1))).to(DEV), N, Cin, True, w8, wsc)
    tol = lambda r: 2.0 ** -7 * np.abs(r) + 2e-3 * np.abs(r).max()
    # ADD: segment 0 plain, segment 1 accumulating
    old = _bf16(rng.standard_normal((B, H, W, C1)) * 1e-3)
    o0 = torch.full((B, H, W, C0), float('nan'), dtype=torch.bfloat16, device=DEV)
    o1 = old.to(DEV).clone()
    K.conv3x3_mx8(B, H, W, q, s, None, None, w8, wsc, Cin, K.EPI_ADD,
                  [K.Seg(C0, out0=o0), K.Seg(C1, out0=o1, accumulate=True)])
    r0, r1 = ref[..., :C0], ref[..., C0:] + old.double().numpy()
    assert (np.abs(o0.double().cpu().numpy() - r0) <= tol(r0)).all()
    assert (np.abs(o1.double().cpu().numpy() - r1) <= tol(r1)).all()
    # BWD: masked by the activated forward tensor, BatchNorm-backward partial sums of the producing layer
    a = _bf16(np.maximum(rng.standard_normal((B, H, W, Cin)), 0))
    zf = _bf16(rng.standard_normal((B, H, W, Cin)))
    mean, istd = rng.standard_normal(Cin).astype(np.float32), (1 + rng.random(Cin)).astype(np.float32)
    P = K.conv3x3_mx8_num_partials(B, H, W, Cin, N, 0)
    segs, outs, parts = [], [], []
    for lo, hi in ((0, C0), (C0, Cin)):
        o = torch.full((B, H, W, hi - lo), float('nan'), dtype=torch.bfloat16, device=DEV)
        pt = torch.full((P, 2, hi - lo), float('nan'), device=DEV)
        segs.append(K.Seg(hi - lo, out0=o, ref=a[..., lo:hi].contiguous().to(DEV), slope=0.0,
                          z=zf[..., lo:hi].contiguous().to(DEV), mean=torch.from_numpy(mean[lo:hi]).to(DEV),
                          istd=torch.from_numpy(istd[lo:hi]).to(DEV), partials=pt))
        outs.append(o)
        parts.append(pt)
    K.conv3x3_mx8(B, H, W, q, s, None, None, w8, wsc, Cin, K.EPI_BWD, segs)
    gm = ref * (a.double().numpy() > 0)
    got = np.concatenate([o.double().cpu().numpy() for o in outs], axis=-1)
    assert (np.abs(got - gm) <= tol(gm)).all()
    xhat = (zf.double().numpy() - mean) * istd
    ps = np.concatenate([p.double().cpu().numpy().sum(0) for p in parts], axis=-1)
    scale = np.abs(gm).sum(axis=(0, 1, 2)).max()
    np.testing.assert_allclose(ps[0], gm.reshape(-1, Cin).sum(0), rtol=0, atol=4e-3 * scale)
    np.testing.assert_allclose(ps[1], (gm * xhat).reshape(-1, Cin).sum(0), rtol=0, atol=8e-3 * scale)


def test_full_size_repeatability_and_error_behaviour():
    """Config 5's top level (B 8, 512 x 512, 64 -> 64 channels, 16 384 workgroups): two launches are bit-identical;
    shapes the kernel does not tile raise instead of computing something else."""
    from audio_depth_estimation_amd import kernels as K
    B, H, W, Cc, N = 8, 512, 512, 64, 64
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(B, H, W, Cc, device=DEV, generator=g).to(torch.bfloat16)
    q, s = _quant_dev(x)
    master = (torch.randn(N, 9, Cc, device=DEV, generator=g) * 0.05)
    s8, ssc = K.mx8_pack_shapes(N, Cc, False)
    w8, wsc = torch.empty(s8, dtype=torch.uint8, device=DEV), torch.empty(ssc, dtype=torch.uint8, device=DEV)
    K.mx8_pack(master, N, Cc, False, w8, wsc)
    P = K.conv3x3_mx8_num_partials(B, H, W, N, Cc, 0)
    outs = []
    for _ in range(3):
        z = torch.full((B, H, W, N), float('nan'), dtype=torch.bfloat16, device=DEV)
        part = torch.full((P, 2, N), float('nan'), device=DEV)
        K.conv3x3_mx8(B, H, W, q, s, None, None, w8, wsc, N, K.EPI_Z_STATS, [K.Seg(N, out0=z, partials=part)])
        outs.append((z, part))
    for z, part in outs[1:]:
        assert torch.equal(z.view(torch.int16), outs[0][0].view(torch.int16)) and torch.equal(part, outs[0][1])
    assert bool(torch.isfinite(outs[0][0].float()).all())
    with pytest.raises(RuntimeError):
        K.conv3x3_mx8(1, 12, 16, q[:1, :12, :16].contiguous(), s[:1, :12, :16].contiguous(), None, None, w8, wsc, N,
                      K.EPI_Z_STATS, [K.Seg(N, out0=outs[0][0], partials=outs[0][1])])
    with pytest.raises(RuntimeError):
        K.mx8_quantize(x.cpu(), q, s)


def test_fused_fp8_copies_equal_quantizing_the_bf16_output():
    """bn_act_mx8 / bn_bwd_apply_mx8 / maxpool2_fwd_mx8 / upsample2x_fwd_mx8: the bf16 result is the plain kernel's (bit for
    bit) and the fp8 copy written alongside is bit-identical to adn_mx8_quantize of that result."""
    from audio_depth_estimation_amd import kernels as K
    g = torch.Generator(device=DEV).manual_seed(9)
    B, H, W, Cc = 2, 16, 24, 96
    u8 = lambda *shape: torch.full(shape, 0xAA, dtype=torch.uint8, device=DEV)

    def check(out_bf16, q8, qs):
        want8, wants = _quant_dev(out_bf16)
        assert torch.equal(qs, wants)
        nz = (want8 & 0x7f) != 0
        assert torch.equal(q8[nz], want8[nz]) and not bool(((q8 & 0x7f)[~nz]).any())

    z = (torch.randn(B, H, W, Cc, device=DEV, generator=g) * 3).to(torch.bfloat16)
    scale, shift = torch.rand(Cc, device=DEV, generator=g) + 0.5, torch.randn(Cc, device=DEV, generator=g)
    y0, y1 = torch.empty_like(z), torch.empty_like(z)
    q8, qs = u8(B, H, W, Cc), u8(B, H, W, Cc // 32)
    K.bn_act(z, B * H * W, Cc, scale, shift, 0.0, None, y0)
    K.bn_act_mx8(z, B * H * W, Cc, scale, shift, y1, q8, qs)
    assert torch.equal(y0.view(torch.int16), y1.view(torch.int16))
    check(y1, q8, qs)

    gr = (torch.randn(B, H, W, Cc, device=DEV, generator=g) * 1e-3).to(torch.bfloat16)
    mean, istd = torch.randn(Cc, device=DEV, generator=g), torch.rand(Cc, device=DEV, generator=g) + 0.5
    coef = torch.randn(2 * Cc, device=DEV, generator=g) * 1e-4
    g0, g1 = gr.clone(), gr.clone()
    K.bn_bwd_apply(g0, z, B * H * W, Cc, scale, mean, istd, coef)
    K.bn_bwd_apply_mx8(g1, z, B * H * W, Cc, scale, mean, istd, coef, q8, qs)
    assert torch.equal(g0.view(torch.int16), g1.view(torch.int16))
    check(g1, q8, qs)

    p0, p1 = torch.empty(B, H // 2, W // 2, Cc, dtype=torch.bfloat16, device=DEV), torch.empty(B, H // 2, W // 2, Cc, dtype=torch.bfloat16, device=DEV)
    q8p, qsp = u8(B, H // 2, W // 2, Cc), u8(B, H // 2, W // 2, Cc // 32)
    K.maxpool2_fwd(z, p0)
    K.maxpool2_fwd_mx8(z, p1, q8p, qsp)
    assert torch.equal(p0.view(torch.int16), p1.view(torch.int16))
    check(p1, q8p, qsp)

    Ho, Wo = 2 * H + 1, 2 * W + 2                        # padded target as in Up.forward (F.pad to the skip's size)
    u0, u1 = torch.empty(B, Ho, Wo, Cc, dtype=torch.bfloat16, device=DEV), torch.empty(B, Ho, Wo, Cc, dtype=torch.bfloat16, device=DEV)
    q8u, qsu = u8(B, Ho, Wo, Cc), u8(B, Ho, Wo, Cc // 32)
    K.upsample2x_fwd(z, u0)
    K.upsample2x_fwd_mx8(z, u1, q8u, qsu)
    assert torch.equal(u0.view(torch.int16), u1.view(torch.int16))
    check(u1, q8u, qsu)
