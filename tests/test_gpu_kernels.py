"""GPU parity of the individual libadn kernels against plain torch-CPU fp32 references.

Every call goes through the C ABI (ctypes -> libadn.so).  Tolerances (stated per test):
  * f32 path (exact-f32 MFMA / f32 VALU): max|err| <= 2e-5 * max|ref|   (accumulation order only)
  * bf16 path: inputs are pre-rounded to bf16 and the reference is computed in fp32 from the SAME
    rounded values, so only accumulation order and the final store rounding differ:
    f32 outputs <= 1e-4 * max|ref|, bf16 outputs <= 6e-3 * max|ref| (one bf16 ulp = 2^-8 relative).
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
    """Round to the storage dtype and come back to fp32 (what the kernel will actually read)."""
    return x.to(dtype).float()


def nhwc(x, dtype):
    return x.permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV)


def from_nhwc(y):
    return y.float().cpu().permute(0, 3, 1, 2)


def pack(w_xy44, dtype):
    """[X,Y,4,4] parameter -> (s2 [X,16,Y], t2 [4,Y,4,X]) device tensors through adn_pack_weights."""
    X, Y = w_xy44.shape[:2]
    master = w_xy44.permute(0, 2, 3, 1).contiguous().to(DEV)       # channels_last memory order
    s2 = torch.empty(X, 16, Y, dtype=dtype, device=DEV)
    t2 = torch.empty(4, Y, 4, X, dtype=dtype, device=DEV)
    K().pack_weights(master, X, Y, dtype, s2, t2)
    return s2, t2


def ws_for(dtype, geom, B, Hs, Ws, C0, C1, N, segs, epi=0):
    P, nbytes = K().igemm_query(dtype, geom, B, Hs, Ws, C0, C1, N, segs, epi=epi)
    ws = torch.empty(max(nbytes, 16) // 4, dtype=torch.float32, device=DEV)
    return P, ws


TOL_F32_OUT = {torch.float32: 2e-5, torch.bfloat16: 1e-4}
TOL_T_OUT = {torch.float32: 2e-5, torch.bfloat16: 6e-3}

# (B, Cin0, Cin1, Cout, Hsmall)   Hsmall = small-grid side
SHAPES = [
    (2, 64, 0, 128, 16),     # MFMA, BN=128
    (2, 64, 64, 64, 8),      # MFMA, BN=64, two gathered sources (virtual concat)
    (3, 128, 0, 128, 2),     # MFMA, tiny M -> split-K + reduce
    (8, 64, 0, 128, 64),     # MFMA, >=256 tiles -> fused LDS epilogue (BN=128)
    (8, 64, 64, 64, 32),     # MFMA, fused epilogue for the 4-phase T2 geometry (BN=64)
    (16, 64, 0, 128, 64),    # MFMA, 256-row tiles / 8 waves / 3-stage LDS-DMA ring (S2: 256 tiles)
    (16, 64, 64, 64, 32),    # same for T2 (64 tiles x 4 phases), BN=64
    (8, 64, 64, 64, 64),     # T2: 128 tiles of 16 x 16 pixels x 4 phases -> the tall 4 x 1-wave patch kernel (BN = 64)
    (4, 128, 0, 256, 8),     # 8 x 8 small-grid images: two images per patch tile (PAIR form), S2 and T2
    (6, 64, 64, 128, 8),     # PAIR form with two gathered sources, three image pairs
    (32, 128, 0, 128, 1),    # 1 x 1 small-grid images (innermost level): only the 4 (S2) / 1 (T2 phase) in-range taps are walked
    (8, 64, 64, 128, 1),     # same with two gathered sources
    (2, 6, 0, 10, 4),        # generic direct path
    (1, 3, 5, 1, 5),         # generic, two sources, single output channel, odd size
]


def test_pack_weights_layout():
    torch.manual_seed(0)
    w = torch.randn(6, 10, 4, 4)
    s2, t2 = pack(w, torch.float32)
    np.testing.assert_array_equal(s2.cpu().numpy(), w.permute(0, 2, 3, 1).reshape(6, 16, 10).numpy())
    kh = {(0, 0): 1, (0, 1): 3, (1, 0): 0, (1, 1): 2}
    t2c = t2.cpu()
    for ph in range(2):
        for pw in range(2):
            for ty in range(2):
                for tx in range(2):
                    ref = w[:, :, kh[(ph, ty)], kh[(pw, tx)]].t()            # [Y, X]
                    np.testing.assert_array_equal(t2c[ph * 2 + pw, :, ty * 2 + tx, :].numpy(), ref.numpy())


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('shape', SHAPES)
def test_conv_forward_s2(dtype, shape):
    """S2 geometry == nn.Conv2d(k4,s2,p1) forward (unetbaseline_model.py:187)."""
    B, C0, C1, N, Hs = shape
    torch.manual_seed(1)
    x = rounded(torch.randn(B, C0 + C1, 2 * Hs, 2 * Hs), dtype)
    w = rounded(torch.randn(N, C0 + C1, 4, 4) * 0.1, dtype)
    ref = F.conv2d(x, w, stride=2, padding=1)
    s2, _ = pack(w, dtype)
    in0 = nhwc(x[:, :C0], dtype)
    in1 = nhwc(x[:, C0:], dtype) if C1 else None
    out = torch.empty(B, Hs, Hs, N, dtype=torch.float32, device=DEV)
    _, ws = ws_for(dtype, 0, B, Hs, Hs, C0, C1, N, [N])
    k = K()
    k.igemm(dtype, 0, B, Hs, Hs, in0, in1, s2, N, 0, [k.Seg(N, out0=out)], ws)
    assert rel_err(from_nhwc(out), ref) <= TOL_F32_OUT[dtype]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('shape', SHAPES)
def test_convT_forward_t2(dtype, shape):
    """T2 geometry == nn.ConvTranspose2d(k4,s2,p1) forward (unetbaseline_model.py:196-220)."""
    B, C0, C1, N, Hs = shape
    torch.manual_seed(2)
    x = rounded(torch.randn(B, C0 + C1, Hs, Hs), dtype)
    w = rounded(torch.randn(C0 + C1, N, 4, 4) * 0.1, dtype)
    ref = F.conv_transpose2d(x, w, stride=2, padding=1)
    _, t2 = pack(w, dtype)
    in0 = nhwc(x[:, :C0], dtype)
    in1 = nhwc(x[:, C0:], dtype) if C1 else None
    out = torch.empty(B, 2 * Hs, 2 * Hs, N, dtype=torch.float32, device=DEV)
    _, ws = ws_for(dtype, 1, B, Hs, Hs, C0, C1, N, [N])
    k = K()
    k.igemm(dtype, 1, B, Hs, Hs, in0, in1, t2, N, 0, [k.Seg(N, out0=out)], ws)
    assert rel_err(from_nhwc(out), ref) <= TOL_F32_OUT[dtype]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('shape', SHAPES)
def test_conv_dgrad_and_convT_dgrad(dtype, shape):
    """conv dgrad = T2 with the conv's phase-packed weights; convT dgrad = S2 with the convT's s2 pack."""
    B, C0, C1, N, Hs = shape
    C = C0 + C1
    torch.manual_seed(3)
    k = K()
    # conv: y = conv2d(x[B,C,2Hs,2Hs], w[N,C,4,4]); dX = conv_transpose2d(dY, w)
    w = rounded(torch.randn(N, C, 4, 4) * 0.1, dtype)
    dy = rounded(torch.randn(B, N, Hs, Hs), dtype)
    ref = F.conv_transpose2d(dy, w, stride=2, padding=1)
    _, t2 = pack(w, dtype)                      # [4][C][4][N]
    out = torch.empty(B, 2 * Hs, 2 * Hs, C, dtype=torch.float32, device=DEV)
    _, ws = ws_for(dtype, 1, B, Hs, Hs, N, 0, C, [C])
    k.igemm(dtype, 1, B, Hs, Hs, nhwc(dy, dtype), None, t2, C, 0, [k.Seg(C, out0=out)], ws)
    assert rel_err(from_nhwc(out), ref) <= TOL_F32_OUT[dtype]
    # convT: z = conv_transpose2d(a[B,C,Hs,Hs], wt[C,N,4,4]); dA = conv2d(dZ, wt as [out=C,in=N])
    wt = rounded(torch.randn(C, N, 4, 4) * 0.1, dtype)
    dz = rounded(torch.randn(B, N, 2 * Hs, 2 * Hs), dtype)
    ref2 = F.conv2d(dz, wt, stride=2, padding=1)
    s2, _ = pack(wt, dtype)                     # [C][16][N]
    out2 = torch.empty(B, Hs, Hs, C, dtype=torch.float32, device=DEV)
    _, ws2 = ws_for(dtype, 0, B, Hs, Hs, N, 0, C, [C])
    k.igemm(dtype, 0, B, Hs, Hs, nhwc(dz, dtype), None, s2, C, 0, [k.Seg(C, out0=out2)], ws2)
    assert rel_err(from_nhwc(out2), ref2) <= TOL_F32_OUT[dtype]


WG_SHAPES = [
    (2, 128, 0, 64, 8),      # MFMA: R=128, C=64 (two taps per column tile)
    (2, 128, 128, 128, 4),   # MFMA: two plain sources (convT input = virtual concat)
    (4, 128, 0, 64, 32),     # patch-staged kernel (bf16): 64 tiles of 8 x 8, R = 128, C = 64 (two channel blocks)
    (2, 64, 128, 96, 64),    # patch-staged: two plain sources of different width, C = 96 = three 32-channel blocks, every border
    (16, 64, 0, 32, 16),     # patch-staged: 2 x 2 tiles per image, single 32-channel block
    (2, 8, 0, 6, 4),         # generic
    (1, 4, 4, 1, 8),         # generic, two plain sources, single gathered channel (outermost convT)
]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('shape', WG_SHAPES)
def test_wgrad(dtype, shape):
    """dW of Conv2d (plain=dZ small grid, gathered=input large grid) and of ConvTranspose2d
    (plain=input small grid, gathered=dZ large grid) against torch autograd."""
    B, R0, R1, Cg, Hs = shape
    R = R0 + R1
    torch.manual_seed(4)
    k = K()
    # conv-style: weight [R, Cg, 4, 4]
    x = rounded(torch.randn(B, Cg, 2 * Hs, 2 * Hs), dtype)
    dz = rounded(torch.randn(B, R, Hs, Hs), dtype)
    w = torch.zeros(R, Cg, 4, 4, requires_grad=True)
    F.conv2d(x, w, stride=2, padding=1).backward(dz)
    ref = w.grad.permute(0, 2, 3, 1).reshape(R, 16, Cg)
    dw = torch.empty(R, 16, Cg, dtype=torch.float32, device=DEV)
    nbytes = k.wgrad_workspace_bytes(dtype, B, Hs, Hs, R0, R1, Cg, 0)
    ws = torch.empty(max(nbytes, 16) // 4, dtype=torch.float32, device=DEV)
    p0 = nhwc(dz[:, :R0], dtype)
    p1 = nhwc(dz[:, R0:], dtype) if R1 else None
    k.wgrad(dtype, B, Hs, Hs, p0, p1, nhwc(x, dtype), None, dw, ws)
    assert rel_err(dw, ref) <= TOL_F32_OUT[dtype]
    # convT-style: weight [R(in), Cg(out), 4, 4]; plain = layer input on the small grid
    a = rounded(torch.randn(B, R, Hs, Hs), dtype)
    dzl = rounded(torch.randn(B, Cg, 2 * Hs, 2 * Hs), dtype)
    wt = torch.zeros(R, Cg, 4, 4, requires_grad=True)
    F.conv_transpose2d(a, wt, stride=2, padding=1).backward(dzl)
    ref2 = wt.grad.permute(0, 2, 3, 1).reshape(R, 16, Cg)
    a0 = nhwc(a[:, :R0], dtype)
    a1 = nhwc(a[:, R0:], dtype) if R1 else None
    k.wgrad(dtype, B, Hs, Hs, a0, a1, nhwc(dzl, dtype), None, dw, ws)
    assert rel_err(dw, ref2) <= TOL_F32_OUT[dtype]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('shape', [
    (32, 128, 0, 128, 1),    # one pixel per image: no pixel split, the MFMA kernel's epilogue writes dW and the partials
    (32, 64, 64, 128, 2),    # two plain sources, still unsplit
    (16, 128, 0, 128, 8),    # pixel split -> the slab sum writes dW and the partials
    (16, 64, 0, 32, 16),     # patch-staged kernel + slab sum
    (2, 8, 0, 6, 4),         # generic kernel, unsplit: no fused form (count 0), the caller covers dW by a range
])
def test_wgrad_norm_partials(dtype, shape):
    """AdnWgradDesc.sq_partials: the kernel that writes the final dW also leaves partial sums of dW^2 behind
    (the fused form of clip_grad_norm_'s pass over the gradient, train.py:689).  dW itself must not change (bit-exact
    against the plain call), the partials must add up to sum(dW^2) of the dW that was written (1e-7 relative: the squares of a
    16-byte group are f32 products, summed in f64 from there on), and adn_grad_norm_ranges must turn partials + uncovered ranges into the same
    total norm / clip coefficient as adn_grad_norm over the whole buffer."""
    B, R0, R1, Cg, Hs = shape
    R = R0 + R1
    torch.manual_seed(11)
    k = K()
    x = nhwc(torch.randn(B, Cg, 2 * Hs, 2 * Hs), dtype)
    dz = torch.randn(B, R, Hs, Hs)
    p0 = nhwc(dz[:, :R0], dtype)
    p1 = nhwc(dz[:, R0:], dtype) if R1 else None
    nbytes = k.wgrad_workspace_bytes(dtype, B, Hs, Hs, R0, R1, Cg, 0)
    ws = torch.empty(max(nbytes, 16) // 4, dtype=torch.float32, device=DEV)
    n = R * 16 * Cg
    pad = 40                                                      # an uncovered "parameter" behind the weight
    flat = torch.zeros(n + pad, dtype=torch.float32, device=DEV)
    plain = torch.empty(n, dtype=torch.float32, device=DEV)
    k.wgrad(dtype, B, Hs, Hs, p0, p1, x, None, plain, ws)
    cnt = k.wgrad_sq_count(dtype, B, Hs, Hs, R0, R1, Cg, 0)
    assert (cnt > 0) == (R % 64 == 0)
    sq = torch.full((max(cnt, 1),), float('nan'), dtype=torch.float64, device=DEV)
    k.wgrad(dtype, B, Hs, Hs, p0, p1, x, None, flat[:n], ws, sq=sq)
    assert torch.equal(flat[:n], plain)
    if cnt == 0:
        assert bool(torch.isnan(sq).all())                       # documented: nothing is written
        return
    want = float((plain.double() ** 2).sum())
    assert abs(float(sq.sum()) - want) <= 1e-7 * want
    flat[n:n + 37] = torch.randn(37, device=DEV)
    state_a = torch.zeros(8, dtype=torch.float64, device=DEV)
    state_b = torch.zeros(8, dtype=torch.float64, device=DEV)
    nws = torch.empty(1024 + 8, dtype=torch.float64, device=DEV)
    k.grad_norm(flat, 0.5, state_a, nws)
    ranges = torch.tensor([[n, 40]], dtype=torch.int64, device=DEV)
    k.grad_norm_ranges(flat, ranges, sq, 0.5, state_b, nws)
    assert abs(float(state_a[3]) - float(state_b[3])) <= 1e-9 * float(state_a[3])      # same f32 group sums, f64 order only
    assert abs(float(state_a[4]) - float(state_b[4])) <= 1e-9 * float(state_a[4])


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('shape', [(2, 64, 0, 128, 16), (3, 128, 0, 128, 2), (2, 6, 0, 10, 4), (8, 64, 0, 128, 64),
                                   (16, 64, 0, 128, 64),
                                   (20, 64, 0, 128, 64),     # ring kernel, 128-column tiles, 320 tiles on <= 256 workgroups
                                   (16, 128, 0, 256, 32),    # ring kernel, 64-column tiles (too few 128-column ones)
                                   (12, 64, 0, 384, 32)])    # ring kernel, 64-column tiles, 6 column tiles per pixel tile, 288 tiles
def test_epilogue_z_stats_and_bn(dtype, shape):
    """Z_STATS epilogue + adn_bn_fwd_finalize + adn_bn_act == conv -> BatchNorm2d(train) -> LeakyReLU / ReLU."""
    B, C0, _, N, Hs = shape
    torch.manual_seed(5)
    k = K()
    x = rounded(torch.randn(B, C0, 2 * Hs, 2 * Hs), dtype)
    w = rounded(torch.randn(N, C0, 4, 4) * 0.1, dtype)
    gamma, beta = torch.rand(N) + 0.5, torch.randn(N) * 0.1
    rm, rv = torch.randn(N) * 0.1, torch.rand(N) + 0.5
    zref = F.conv2d(x, w, stride=2, padding=1)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    yref = F.batch_norm(zref, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5)
    s2, _ = pack(w, dtype)
    P, ws = ws_for(dtype, 0, B, Hs, Hs, C0, 0, N, [N])
    z = torch.empty(B, Hs, Hs, N, dtype=dtype, device=DEV)
    partials = torch.zeros(P, 2, N, dtype=torch.float32, device=DEV)
    k.igemm(dtype, 0, B, Hs, Hs, nhwc(x, dtype), None, s2, N, 1, [k.Seg(N, out0=z, partials=partials)], ws)
    assert rel_err(from_nhwc(z), zref) <= TOL_T_OUT[dtype]
    cnt = B * Hs * Hs
    s1 = partials[:, 0].double().sum(0).cpu()
    assert rel_err(s1.float(), zref.sum((0, 2, 3))) <= 1e-4 + TOL_F32_OUT[dtype]
    dev = lambda t: t.to(DEV)
    mean, istd, scale, shift = [torch.empty(N, device=DEV) for _ in range(4)]
    rm_d, rv_d = dev(rm.clone()), dev(rv.clone())
    nbt = torch.zeros(1, dtype=torch.int64, device=DEV)
    k.bn_fwd_finalize(partials, P, N, cnt, dev(gamma), dev(beta), 1e-5, 0.1, rm_d, rv_d, nbt, mean, istd, scale, shift)
    assert rel_err(rm_d, rm_ref) <= 1e-4 and rel_err(rv_d, rv_ref) <= 1e-4 and int(nbt.item()) == 1
    leaky = torch.empty_like(z)
    relu = torch.empty_like(z)
    k.bn_act(z, cnt, N, scale, shift, 0.2, leaky, relu)
    assert rel_err(from_nhwc(leaky), F.leaky_relu(yref, 0.2)) <= 3 * TOL_T_OUT[dtype]
    assert rel_err(from_nhwc(relu), F.relu(yref)) <= 3 * TOL_T_OUT[dtype]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('shape', [(2, 128, 0, 64, 8), (2, 10, 0, 6, 4), (8, 128, 0, 64, 64), (16, 128, 0, 64, 64),
                                   (16, 128, 0, 128, 64),    # ring kernel: S2 BWD with two 128-channel segments, T2 BWD accumulating
                                   (20, 128, 0, 64, 64),     # ring kernel: the two segments inside ONE 128-column tile, 320 tiles
                                   (32, 64, 0, 128, 64)])    # ring kernel: D1 dgrad of unet_256 at B = 32 (4 tiles per workgroup, 2 super-steps)
def test_epilogue_bwd_two_segments(dtype, shape):
    """convT dgrad with ReLU mask, split into a skip segment (no stats) and an up segment (BN-bwd stats),
    then accumulate a second contribution with a LeakyReLU mask."""
    B, Cz, _, Chalf, Hs = shape          # dZ channels, each segment has Chalf channels
    Cin = 2 * Chalf
    torch.manual_seed(6)
    k = K()
    wt = rounded(torch.randn(Cin, Cz, 4, 4) * 0.1, dtype)
    dz = rounded(torch.randn(B, Cz, 2 * Hs, 2 * Hs), dtype)
    dA = F.conv2d(dz, wt, stride=2, padding=1)                       # [B, Cin, Hs, Hs]
    ref_act = [rounded(torch.randn(B, Chalf, Hs, Hs), dtype) for _ in range(2)]   # activated fwd tensors
    zfwd = rounded(torch.randn(B, Chalf, Hs, Hs), dtype)
    mean, istd = torch.randn(Chalf) * 0.1, torch.rand(Chalf) + 0.5
    g0_ref = dA[:, :Chalf] * (ref_act[0] > 0).float()
    g1_ref = dA[:, Chalf:] * (ref_act[1] > 0).float()
    xhat = (zfwd - mean.view(1, -1, 1, 1)) * istd.view(1, -1, 1, 1)
    s2, _ = pack(wt, dtype)
    P, ws = ws_for(dtype, 0, B, Hs, Hs, Cz, 0, Cin, [Chalf, Chalf])
    g0 = torch.empty(B, Hs, Hs, Chalf, dtype=dtype, device=DEV)
    g1 = torch.empty_like(g0)
    partials = torch.zeros(P, 2, Chalf, dtype=torch.float32, device=DEV)
    segs = [k.Seg(Chalf, out0=g0, ref=nhwc(ref_act[0], dtype), slope=0.0),
            k.Seg(Chalf, out0=g1, ref=nhwc(ref_act[1], dtype), z=nhwc(zfwd, dtype), mean=mean.to(DEV),
                  istd=istd.to(DEV), partials=partials, slope=0.0)]
    k.igemm(dtype, 0, B, Hs, Hs, nhwc(dz, dtype), None, s2, Cin, 3, segs, ws)
    assert rel_err(from_nhwc(g0), g0_ref) <= TOL_T_OUT[dtype]
    assert rel_err(from_nhwc(g1), g1_ref) <= TOL_T_OUT[dtype]
    assert rel_err(partials[:, 0].sum(0), g1_ref.sum((0, 2, 3))) <= 1e-3
    assert rel_err(partials[:, 1].sum(0), (g1_ref * xhat).sum((0, 2, 3))) <= 1e-3
    # second contribution accumulated with a leaky mask (conv dgrad -> T2 geometry)
    wc = rounded(torch.randn(Cz, Chalf, 4, 4) * 0.1, dtype)          # conv weight [Cout=Cz, Cin=Chalf]
    dzs = rounded(torch.randn(B, Cz, Hs // 2, Hs // 2), dtype)
    dX = F.conv_transpose2d(dzs, wc, stride=2, padding=1)            # [B, Chalf, Hs, Hs]
    acc_ref = from_nhwc(g0) + dX * torch.where(ref_act[0] > 0, 1.0, 0.2)
    _, t2 = pack(wc, dtype)
    P2, ws2 = ws_for(dtype, 1, B, Hs // 2, Hs // 2, Cz, 0, Chalf, [Chalf])
    k.igemm(dtype, 1, B, Hs // 2, Hs // 2, nhwc(dzs, dtype), None, t2, Chalf, 3,
            [k.Seg(Chalf, out0=g0, ref=nhwc(ref_act[0], dtype), slope=0.2, accumulate=True)], ws2)
    assert rel_err(from_nhwc(g0), acc_ref) <= 2 * TOL_T_OUT[dtype]


@pytest.mark.parametrize('shapes', [[(32, 128, 0, 64), (32, 256, 0, 128)],                      # (Hs, R0, R1, C) at B = 8
                                    [(64, 128, 0, 64), (32, 256, 0, 128), (32, 512, 0, 256)],
                                    [(32, 128, 128, 64), (32, 256, 256, 128), (32, 64, 0, 32), (64, 64, 64, 64)]])
def test_wgrad_patch_batch_against_single_launches(shapes):
    """adn_wgrad_patch_batch (patch-staged layers in one launch with 1/n of the pixel splits each, then one slab sum per
    layer) against one adn_wgrad per layer: dW to 1e-5 of its max (another summation order), sums of the norm partials to 1e-6."""
    k = K()
    dtype, B = torch.bfloat16, 8
    torch.manual_seed(23)
    assert k.wgrad_patch_batch_workspace_bytes(dtype, B, [(h, h, r0, r1, c, 0) for h, r0, r1, c in shapes]) >= 0
    probs, refs = [], []
    for Hs, R0, R1, C in shapes:
        p0 = torch.randn(B, Hs, Hs, R0, device=DEV).to(dtype)
        p1 = torch.randn(B, Hs, Hs, R1, device=DEV).to(dtype) if R1 else None
        g = torch.randn(B, 2 * Hs, 2 * Hs, C, device=DEV).to(dtype)
        n = (R0 + R1) * 16 * C
        nsq = k.wgrad_sq_count(dtype, B, Hs, Hs, R0, R1, C, 0)
        ws = torch.empty(max(k.wgrad_workspace_bytes(dtype, B, Hs, Hs, R0, R1, C, 0), 16) // 4, device=DEV)
        dw_ref = torch.empty(n, device=DEV)
        sq_ref = torch.zeros(max(nsq, 1), dtype=torch.float64, device=DEV)
        k.wgrad(dtype, B, Hs, Hs, p0, p1, g, None, dw_ref, ws, sq=sq_ref if nsq else None)
        dw = torch.full((n,), float('nan'), device=DEV)
        sq = torch.full((max(nsq, 1),), float('nan'), dtype=torch.float64, device=DEV)
        probs.append((Hs, Hs, p0, p1, g, None, dw, sq if nsq else None))
        refs.append((dw_ref, sq_ref, nsq))
    need = k.wgrad_patch_batch_workspace_bytes(dtype, B, [(h, h, r0, r1, c, 0) for h, r0, r1, c in shapes])
    ws = torch.empty(max(need, 16) // 4, device=DEV)
    k.wgrad_patch_batch(dtype, B, probs, ws)
    for (_, _, _, _, _, _, dw, sq), (dw_ref, sq_ref, nsq) in zip(probs, refs):
        assert rel_err(dw, dw_ref.cpu()) <= 1e-5
        if nsq:
            assert abs(float(sq.sum()) - float(sq_ref.sum())) <= 1e-6 * float(sq_ref.sum())


def test_wgrad_batch_equals_single_launches():
    """adn_wgrad_batch (several small-image weight gradients in one launch) is bit-identical to one adn_wgrad per problem,
    dW and the norm partials alike; a problem the unsplit tap-staged kernel does not take is refused."""
    k = K()
    dtype, B = torch.bfloat16, 32
    torch.manual_seed(19)
    shapes = [(2, 512, 0, 512), (1, 512, 0, 512), (2, 512, 512, 512), (4, 512, 512, 512), (1, 128, 0, 256)]   # Hs, R0, R1, C
    probs, singles = [], []
    for Hs, R0, R1, C in shapes:
        assert k.wgrad_batchable(dtype, B, Hs, Hs, R0, R1, C, 0)[0] == 1, (Hs, R0, R1, C)
        p0 = torch.randn(B, Hs, Hs, R0, device=DEV).to(dtype)
        p1 = torch.randn(B, Hs, Hs, R1, device=DEV).to(dtype) if R1 else None
        g = torch.randn(B, 2 * Hs, 2 * Hs, C, device=DEV).to(dtype)
        n = (R0 + R1) * 16 * C
        nsq = k.wgrad_sq_count(dtype, B, Hs, Hs, R0, R1, C, 0)
        assert nsq > 0 and nsq == k.wgrad_batchable(dtype, B, Hs, Hs, R0, R1, C, 0)[1]      # (unsplit layers: the same count)
        ws = torch.empty(max(k.wgrad_workspace_bytes(dtype, B, Hs, Hs, R0, R1, C, 0), 16) // 4, device=DEV)
        dw_ref, sq_ref = torch.empty(n, device=DEV), torch.zeros(nsq, dtype=torch.float64, device=DEV)
        k.wgrad(dtype, B, Hs, Hs, p0, p1, g, None, dw_ref, ws, sq=sq_ref)
        dw, sq = torch.full((n,), float('nan'), device=DEV), torch.full((nsq,), float('nan'), dtype=torch.float64, device=DEV)
        probs.append((Hs, Hs, p0, p1, g, None, dw, sq))
        singles.append((dw_ref, sq_ref))
    k.wgrad_batch(dtype, B, probs)
    for (_, _, _, _, _, _, dw, sq), (dw_ref, sq_ref) in zip(probs, singles):
        assert torch.equal(dw, dw_ref) and torch.equal(sq, sq_ref)
    k.wgrad_batch(dtype, B, [probs[1][:7] + (None,)])                      # one problem, no norm partials
    assert torch.equal(probs[1][6], singles[1][0])
    assert k.wgrad_batchable(dtype, B, 64, 64, 128, 0, 64, 0) == (0, 0)   # a patch-staged layer
    # a layer whose lone launch splits the pixels (slab sum) runs unsplit inside a batch: same dW up to the summation order,
    # tiles_r * tiles_c norm partials; the power-of-two-image form is its own class
    for (Hs, R0, R1, C), want in (((4, 512, 0, 512), 1), ((8, 512, 0, 512), 2), ((8, 512, 512, 512), 2)):
        cls, nsq = k.wgrad_batchable(dtype, B, Hs, Hs, R0, R1, C, 0)
        assert cls == want and nsq == ((R0 + R1) // 128) * (16 * C // 128)
        p0 = torch.randn(B, Hs, Hs, R0, device=DEV).to(dtype)
        p1 = torch.randn(B, Hs, Hs, R1, device=DEV).to(dtype) if R1 else None
        g = torch.randn(B, 2 * Hs, 2 * Hs, C, device=DEV).to(dtype)
        n = (R0 + R1) * 16 * C
        ws = torch.empty(max(k.wgrad_workspace_bytes(dtype, B, Hs, Hs, R0, R1, C, 0), 16) // 4, device=DEV)
        dw_ref = torch.empty(n, device=DEV)
        k.wgrad(dtype, B, Hs, Hs, p0, p1, g, None, dw_ref, ws)
        dw, sq = torch.empty(n, device=DEV), torch.full((nsq,), float('nan'), dtype=torch.float64, device=DEV)
        k.wgrad_batch(dtype, B, [(Hs, Hs, p0, p1, g, None, dw, sq)])
        assert rel_err(dw, dw_ref.cpu()) <= 1e-5
        assert abs(float(sq.sum()) - float(dw_ref.double().pow(2).sum())) <= 1e-5 * float(dw_ref.double().pow(2).sum())
    with pytest.raises(RuntimeError, match='one class per launch'):
        k.wgrad_batch(dtype, B, [probs[0], (8, 8, torch.zeros(B, 8, 8, 512, device=DEV, dtype=dtype), None,
                                            torch.zeros(B, 16, 16, 512, device=DEV, dtype=dtype), None,
                                            torch.empty(512 * 16 * 512, device=DEV), None)])
    Hs = 64
    bad = (Hs, Hs, torch.zeros(B, Hs, Hs, 128, device=DEV, dtype=dtype), None,
           torch.zeros(B, 2 * Hs, 2 * Hs, 64, device=DEV, dtype=dtype), None, torch.empty(128 * 16 * 64, device=DEV), None)
    with pytest.raises(RuntimeError, match='not batchable'):
        k.wgrad_batch(dtype, B, [bad])


@pytest.mark.parametrize('accumulate', [False, True])
@pytest.mark.parametrize('geom,shape', [(0, (16, 128, 128, 64)), (0, (32, 64, 256, 64)), (1, (16, 256, 128, 16)),
                                        (1, (8, 128, 64, 32)), (0, (2, 128, 128, 8))])
def test_epilogue_bwd_mask_from_z(geom, shape, accumulate):
    """BWD epilogue with the forward's scale / shift passed along (bf16): ref = act(z * scale + shift) as the forward's
    apply kernel writes it, so a kernel may take the mask from z and skip ref -- the ring kernel does (first four
    shapes; the last runs the split-K path, which reads ref).  Gradient and both BatchNorm sums against torch."""
    dtype = torch.bfloat16
    B, Cz, N, Hs = shape                 # gradient channels in, channels out, small-grid size
    torch.manual_seed(13)
    k = K()
    if geom == 0:                        # dgrad of a transposed conv (S2 geometry): out on the small grid
        wt = rounded(torch.randn(N, Cz, 4, 4) * 0.1, dtype)
        dz = rounded(torch.randn(B, Cz, 2 * Hs, 2 * Hs), dtype)
        dA = F.conv2d(dz, wt, stride=2, padding=1)
        w_op, Ho = pack(wt, dtype)[0], Hs
    else:                                # dgrad of a conv (T2 geometry): out on the large grid
        wt = rounded(torch.randn(Cz, N, 4, 4) * 0.1, dtype)
        dz = rounded(torch.randn(B, Cz, Hs, Hs), dtype)
        dA = F.conv_transpose2d(dz, wt, stride=2, padding=1)
        w_op, Ho = pack(wt, dtype)[1], 2 * Hs
    zfwd = rounded(torch.randn(B, N, Ho, Ho), dtype)
    scale, shift = torch.randn(N) * 0.5 + 1.0, torch.randn(N) * 0.3
    scale[::7] *= -1.0                                                    # (a negative gamma flips the mask)
    slope = 0.2
    act = torch.addcmul(shift.view(1, -1, 1, 1), zfwd, scale.view(1, -1, 1, 1))
    ref_act = rounded(F.leaky_relu(act, slope), dtype)
    mean, istd = torch.randn(N) * 0.1, torch.rand(N) + 0.5
    old = rounded(torch.randn(B, N, Ho, Ho), dtype)
    g_ref = dA * torch.where(act > 0, 1.0, slope) + (old if accumulate else 0.0)
    xhat = (zfwd - mean.view(1, -1, 1, 1)) * istd.view(1, -1, 1, 1)
    P, ws = ws_for(dtype, geom, B, Hs, Hs, Cz, 0, N, [N], epi=3)
    out = nhwc(old, dtype).clone()
    partials = torch.zeros(P, 2, N, dtype=torch.float32, device=DEV)
    seg = k.Seg(N, out0=out, ref=nhwc(ref_act, dtype), z=nhwc(zfwd, dtype), mean=mean.to(DEV), istd=istd.to(DEV),
                partials=partials, slope=slope, accumulate=accumulate, scale=scale.to(DEV), shift=shift.to(DEV))
    k.igemm(dtype, geom, B, Hs, Hs, nhwc(dz, dtype), None, w_op, N, 3, [seg], ws)
    assert rel_err(from_nhwc(out), g_ref) <= 2 * TOL_T_OUT[dtype]
    assert rel_err(partials[:, 0].sum(0), g_ref.sum((0, 2, 3))) <= 2e-3
    assert rel_err(partials[:, 1].sum(0), (g_ref * xhat).sum((0, 2, 3))) <= 2e-3


@pytest.mark.parametrize('shape', [(8, 64, 64, 128, 32),      # ring kernel T2, 64-column tiles, two gathered sources
                                   (16, 128, 128, 128, 32),   # ring kernel T2, 128-column tiles
                                   (5, 128, 0, 256, 64),      # ring kernel T2, 320 pixel tiles x 4 phases x 2 column tiles
                                   (2, 64, 64, 128, 16)])     # (split-K path for comparison)
def test_convT_z_stats_t2(shape):
    """Z_STATS epilogue of the transposed-conv geometry in bf16 (the up path's forward): z and the BatchNorm column sums
    against ConvTranspose2d; the large shapes run the ring-fed persistent kernel (igemm_ring.h)."""
    dtype = torch.bfloat16
    B, C0, C1, N, Hs = shape
    torch.manual_seed(11)
    k = K()
    x = rounded(torch.randn(B, C0 + C1, Hs, Hs), dtype)
    w = rounded(torch.randn(C0 + C1, N, 4, 4) * 0.1, dtype)
    zref = F.conv_transpose2d(x, w, stride=2, padding=1)
    _, t2 = pack(w, dtype)
    in0 = nhwc(x[:, :C0], dtype)
    in1 = nhwc(x[:, C0:], dtype) if C1 else None
    P, ws = ws_for(dtype, 1, B, Hs, Hs, C0, C1, N, [N])
    z = torch.empty(B, 2 * Hs, 2 * Hs, N, dtype=dtype, device=DEV)
    partials = torch.zeros(P, 2, N, dtype=torch.float32, device=DEV)
    k.igemm(dtype, 1, B, Hs, Hs, in0, in1, t2, N, 1, [k.Seg(N, out0=z, partials=partials)], ws)
    assert rel_err(from_nhwc(z), zref) <= TOL_T_OUT[dtype]
    assert rel_err(partials[:, 0].double().sum(0).float(), zref.sum((0, 2, 3))) <= 2e-4
    assert rel_err(partials[:, 1].double().sum(0).float(), (zref * zref).sum((0, 2, 3))) <= 2e-4


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_bn_backward_pieces(dtype):
    """adn_bn_bwd_finalize + adn_bn_bwd_apply == BatchNorm2d(train) backward."""
    torch.manual_seed(7)
    k = K()
    B, C, H = 4, 24, 6
    z = rounded(torch.randn(B, C, H, H), dtype).requires_grad_(True)
    gamma = (torch.rand(C) + 0.5).requires_grad_(True)
    beta = torch.zeros(C, requires_grad=True)
    y = F.batch_norm(z, None, None, gamma, beta, True, 0.1, 1e-5)
    g = rounded(torch.randn(B, C, H, H), dtype)
    y.backward(g)
    zd = z.detach()
    mu = zd.mean((0, 2, 3))
    var = zd.var((0, 2, 3), unbiased=False)
    istd = 1.0 / torch.sqrt(var + 1e-5)
    xhat = (zd - mu.view(1, -1, 1, 1)) * istd.view(1, -1, 1, 1)
    cnt = B * H * H
    partials = torch.stack([g.sum((0, 2, 3)), (g * xhat).sum((0, 2, 3))]).view(1, 2, C).to(DEV)
    dgamma, dbeta = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    coef = torch.empty(2 * C, device=DEV)
    k.bn_bwd_finalize(partials, 1, C, cnt, dgamma, dbeta, coef)
    assert rel_err(dgamma, gamma.grad) <= 1e-4 and rel_err(dbeta, beta.grad) <= 1e-4
    gd = nhwc(g, dtype)
    k.bn_bwd_apply(gd, nhwc(zd, dtype), cnt, C, (gamma.detach() * istd).to(DEV), mu.to(DEV), istd.to(DEV), coef)
    assert rel_err(from_nhwc(gd), z.grad) <= 2 * TOL_T_OUT[dtype]


def test_layout_roundtrip():
    k = K()
    x = torch.randn(2, 3, 5, 7, device=DEV)
    for dt in (torch.float32, torch.bfloat16):
        y = torch.empty(2, 5, 7, 3, dtype=dt, device=DEV)
        k.nchw_to_nhwc(x, y)
        assert torch.equal(y, x.permute(0, 2, 3, 1).to(dt))
        back = torch.empty_like(x)
        k.nhwc_to_nchw(y, back)
        assert torch.equal(back, x.to(dt).float())


def test_missing_gpu_tensor_fails_loudly():
    k = K()
    with pytest.raises(RuntimeError):
        k.nchw_to_nhwc(torch.randn(1, 1, 2, 2), torch.empty(1, 2, 2, 1))


@pytest.mark.parametrize('P', [7, 256, 257, 3000])
def test_bn_finalize_many_partial_rows(P):
    """Both finalize kernels over P partial rows (P > 256 takes the one-workgroup-per-channel variant) vs f64 sums."""
    k = K()
    C, cnt = 24, 12345
    g = torch.Generator().manual_seed(P)
    part = torch.randn(P, 2, C, generator=g)
    part[:, 1] = part[:, 1].abs() * 50 + 3          # sum of squares side: positive, variance stays > 0
    s1, s2 = part[:, 0].double().sum(0), part[:, 1].double().sum(0)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rm, rv = torch.zeros(C), torch.ones(C)
    mu = s1 / cnt
    var = (s2 / cnt - mu * mu).clamp_min(0)
    istd_ref = 1.0 / torch.sqrt(var + 1e-5)
    dev = lambda t: t.float().to(DEV)
    mean, istd, scale, shift = (torch.empty(C, device=DEV) for _ in range(4))
    rm_d, rv_d, nbt = dev(rm), dev(rv), torch.zeros(1, dtype=torch.int64, device=DEV)
    k.bn_fwd_finalize(part.to(DEV).contiguous(), P, C, cnt, dev(gamma), dev(beta), 1e-5, 0.1, rm_d, rv_d, nbt, mean, istd,
                      scale, shift)
    assert rel_err(mean, mu) <= 1e-6 and rel_err(istd, istd_ref) <= 1e-6
    assert rel_err(scale, gamma.double() * istd_ref) <= 1e-6
    assert rel_err(shift, beta.double() - mu * gamma.double() * istd_ref) <= 1e-5
    assert rel_err(rm_d, 0.1 * mu) <= 1e-6 and int(nbt) == 1
    assert rel_err(rv_d, 0.9 + 0.1 * var * cnt / (cnt - 1)) <= 1e-6
    dgamma, dbeta, coef = torch.empty(C, device=DEV), torch.empty(C, device=DEV), torch.empty(2 * C, device=DEV)
    k.bn_bwd_finalize(part.to(DEV).contiguous(), P, C, cnt, dgamma, dbeta, coef)
    assert rel_err(dbeta, s1) <= 1e-6 and rel_err(dgamma, s2) <= 1e-6
    assert rel_err(coef[:C], s1 / cnt) <= 1e-6 and rel_err(coef[C:], s2 / cnt) <= 1e-6


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('shape', [(37, 7, 64), (512, 2048, 512), (3, 130, 96)])
def test_bn_fused_small_tensor_kernels(dtype, shape):
    """adn_bn_fwd_fused / adn_bn_bwd_fused (one launch) == the finalize + apply pairs on the same partial rows."""
    k = K()
    P, pixels, C = shape
    g = torch.Generator().manual_seed(P + C)
    part = torch.randn(P, 2, C, generator=g)
    part[:, 1] = part[:, 1].abs() * 50 + 3
    part = part.to(DEV).contiguous()
    z = torch.randn(pixels, C, generator=g).to(dtype).to(DEV)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(DEV), torch.randn(C, generator=g).to(DEV)
    cnt = pixels

    def run(fused):
        mean, istd, scale, shift = (torch.empty(C, device=DEV) for _ in range(4))
        rm, rv, nbt = torch.zeros(C, device=DEV), torch.ones(C, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV)
        lk, rl = torch.empty_like(z), torch.empty_like(z)
        if fused:
            k.bn_fwd_fused(part, P, C, cnt, gamma, beta, 1e-5, 0.1, rm, rv, nbt, mean, istd, scale, shift, z, pixels, 0.2,
                           lk, rl)
        else:
            k.bn_fwd_finalize(part, P, C, cnt, gamma, beta, 1e-5, 0.1, rm, rv, nbt, mean, istd, scale, shift)
            k.bn_act(z, pixels, C, scale, shift, 0.2, lk, rl)
        return mean, istd, scale, shift, rm, rv, nbt, lk, rl

    a, b = run(True), run(False)
    for x, y in zip(a[:6], b[:6]):
        assert rel_err(x, y) <= 1e-6
    assert int(a[6]) == int(b[6]) == 1
    assert rel_err(a[7], b[7]) <= TOL_T_OUT[dtype] and rel_err(a[8], b[8]) <= TOL_T_OUT[dtype]
    # a single output is allowed
    lk2 = torch.empty_like(z)
    k.bn_fwd_fused(part, P, C, cnt, gamma, beta, 1e-5, 0.1, None, None, None, a[0].clone(), a[1].clone(), a[2].clone(),
                   a[3].clone(), z, pixels, 0.0, None, lk2)
    assert torch.equal(lk2, a[8])
    # backward
    mean, istd, scale = a[0], a[1], a[2]
    gsrc = torch.randn(pixels, C, generator=g).to(dtype).to(DEV)
    g1, g2 = gsrc.clone(), gsrc.clone()
    dg1, db1, dg2, db2 = (torch.empty(C, device=DEV) for _ in range(4))
    coef = torch.empty(2 * C, device=DEV)
    k.bn_bwd_fused(part, P, C, cnt, dg1, db1, g1, z, pixels, scale, mean, istd)
    k.bn_bwd_finalize(part, P, C, cnt, dg2, db2, coef)
    k.bn_bwd_apply(g2, z, pixels, C, scale, mean, istd, coef)
    assert rel_err(dg1, dg2) <= 1e-6 and rel_err(db1, db2) <= 1e-6
    assert rel_err(g1, g2) <= TOL_T_OUT[dtype]


def test_bn_partials_prereduction():
    """adn_bn_partials_reduce + finalize on the 64 pre-reduced rows equals the finalize over all rows (f64 sums of the same f32
    values, re-associated: <= 1e-6 relative on mean / istd), forward and backward form; ragged last slice."""
    k = K()
    torch.manual_seed(2)
    P, Cc, count = 5000, 96, 5000 * 128
    part = (torch.randn(P, 2, Cc) * 3).abs().to(DEV)                      # [P][2][C]: sum z, sum z^2 (kept positive / consistent)
    part[:, 1] = part[:, 1] + part[:, 0] ** 2 / 128
    scratch = torch.full((k.BN_REDUCE_SLICES * 2 * Cc,), float('nan'), device=DEV)
    gamma, beta = torch.rand(Cc, device=DEV) + 0.5, torch.randn(Cc, device=DEV)
    outs = []
    for sc in (None, scratch):
        mean, istd, scale, shift = [torch.empty(Cc, device=DEV) for _ in range(4)]
        k.bn_fwd_finalize(part.view(-1), P, Cc, count, gamma, beta, 1e-5, 0.1, None, None, None, mean, istd, scale, shift,
                          scratch=sc)
        outs.append((mean, istd, scale, shift))
    for a, b in zip(*outs):
        assert float((a - b).abs().max()) <= 1e-6 * float(a.abs().max()) + 1e-9
    rows = scratch.view(k.BN_REDUCE_SLICES, 2, Cc)
    R = -(-P // k.BN_REDUCE_SLICES)
    want = torch.stack([part[s * R:(s + 1) * R].double().sum(0) for s in range(k.BN_REDUCE_SLICES)]).float()
    assert float((rows - want).abs().max()) <= 1e-6 * float(want.abs().max())
    coefs = []
    for sc in (None, scratch):
        dg, db, coef = torch.empty(Cc, device=DEV), torch.empty(Cc, device=DEV), torch.empty(2 * Cc, device=DEV)
        k.bn_bwd_finalize(part.view(-1), P, Cc, count, dg, db, coef, scratch=sc)
        coefs.append(torch.cat([dg, db, coef]))
    assert float((coefs[0] - coefs[1]).abs().max()) <= 1e-6 * float(coefs[0].abs().max())
