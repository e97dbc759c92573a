"""GPU parity of the thin edge layers: first Conv2d (Cin=2) and last ConvTranspose2d (Cout=1).

They run on the same MFMA kernels as the wide layers with the thin channel dimension zero-padded to one
16-byte chunk (8 bf16 / 4 f32), plus the dedicated pointwise-GEMM + col2im forward of the Cout=1 layer.
Tolerances as in test_gpu_kernels.py.
"""
import pytest
import torch
import torch.nn.functional as F

from test_gpu_kernels import DEV, K, TOL_F32_OUT, TOL_T_OUT, from_nhwc, nhwc, rel_err, rounded, ws_for

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_first_conv_padded_channels(dtype):
    k = K()
    epc = 8 if dtype == torch.bfloat16 else 4
    torch.manual_seed(8)
    B, Cin, Cout, Hs = 8, 2, 64, 64                      # M = 32768 -> fused epilogue
    x = rounded(torch.randn(B, Cin, 2 * Hs, 2 * Hs), dtype)
    w = rounded(torch.randn(Cout, Cin, 4, 4) * 0.1, dtype)
    ref = F.conv2d(x, w, stride=2, padding=1)
    xp = torch.empty(B, 2 * Hs, 2 * Hs, epc, dtype=dtype, device=DEV)
    k.nchw_to_nhwc(x.to(DEV), xp)
    assert float(xp[..., Cin:].float().abs().max()) == 0.0
    master = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    s2 = torch.empty(Cout, 16, epc, dtype=dtype, device=DEV)
    k.pack_weights(master, Cout, Cin, dtype, s2, None, y_pad=epc)
    out = torch.empty(B, Hs, Hs, Cout, dtype=torch.float32, device=DEV)
    _, ws = ws_for(dtype, 0, B, Hs, Hs, epc, 0, Cout, [Cout])
    assert ws.numel() <= 4                                # MFMA path, no slab
    k.igemm(dtype, 0, B, Hs, Hs, xp, None, s2, Cout, 0, [k.Seg(Cout, out0=out)], ws)
    assert rel_err(from_nhwc(out), ref) <= TOL_F32_OUT[dtype]
    # weight gradient: plain = dZ (R=64, half an MFMA row tile), gathered = padded input, compact to Cin=2
    dz = rounded(torch.randn(B, Cout, Hs, Hs), dtype)
    wz = torch.zeros(Cout, Cin, 4, 4, requires_grad=True)
    F.conv2d(x, wz, stride=2, padding=1).backward(dz)
    refw = wz.grad.permute(0, 2, 3, 1).reshape(Cout, 16, Cin)
    dw = torch.empty(Cout, 16, Cin, dtype=torch.float32, device=DEV)
    nbytes = k.wgrad_workspace_bytes(dtype, B, Hs, Hs, Cout, 0, epc, 0, Cin)
    wsw = torch.empty(max(nbytes, 16) // 4, dtype=torch.float32, device=DEV)
    k.wgrad(dtype, B, Hs, Hs, nhwc(dz, dtype), None, xp, None, dw, wsw, c_valid=Cin)
    assert rel_err(dw, refw) <= TOL_F32_OUT[dtype]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('final_act', [0, 1])
def test_last_convt_single_channel(dtype, final_act):
    k = K()
    epc = 8 if dtype == torch.bfloat16 else 4
    torch.manual_seed(9)
    B, C0, C1, Hs = 4, 64, 64, 32
    a = rounded(torch.randn(B, C0 + C1, Hs, Hs), dtype)
    wt = rounded(torch.randn(C0 + C1, 1, 4, 4) * 0.05, dtype)
    bias = torch.tensor([0.3])
    z = F.conv_transpose2d(a, wt, bias, stride=2, padding=1)
    ref = torch.sigmoid(z) if final_act else F.relu(z)
    master = wt.permute(0, 2, 3, 1).contiguous().view(-1).to(DEV)     # [Cin][16][1] f32
    out = torch.empty(B, 2 * Hs, 2 * Hs, dtype=torch.float32, device=DEV)
    ws = torch.empty(k.convt_n1_workspace_bytes(B, Hs, Hs) // 4, dtype=torch.float32, device=DEV)
    a0, a1 = nhwc(a[:, :C0], dtype), nhwc(a[:, C0:], dtype)
    k.convt_n1_forward(dtype, B, Hs, Hs, a0, a1, master, bias.to(DEV), final_act, out, ws)
    assert rel_err(out.cpu().view(B, 1, 2 * Hs, 2 * Hs), ref) <= TOL_F32_OUT[dtype]
    # backward of the final activation into a padded single-channel gradient
    gout = torch.randn(B, 1, 2 * Hs, 2 * Hs)
    dzp = torch.empty(B, 2 * Hs, 2 * Hs, epc, dtype=dtype, device=DEV)
    k.final_act_bwd(gout.to(DEV), out, final_act, dzp)
    outc = out.cpu().view(B, 1, 2 * Hs, 2 * Hs)
    dz_ref = gout * (outc * (1 - outc) if final_act else (outc > 0).float())
    assert rel_err(dzp[..., 0].float().cpu(), dz_ref[:, 0]) <= TOL_T_OUT[dtype]
    assert float(dzp[..., 1:].float().abs().max()) == 0.0
    dzr = dzp[..., 0].float().cpu().unsqueeze(1)                        # the rounded gradient the kernels see
    # dgrad: S2 geometry over the padded gradient
    dA = F.conv2d(dzr, wt, stride=2, padding=1)
    s2 = torch.empty(C0 + C1, 16, epc, dtype=dtype, device=DEV)
    k.pack_weights(master, C0 + C1, 1, dtype, s2, None, y_pad=epc)
    og = torch.empty(B, Hs, Hs, C0 + C1, dtype=torch.float32, device=DEV)
    _, ws2 = ws_for(dtype, 0, B, Hs, Hs, epc, 0, C0 + C1, [C0 + C1])
    k.igemm(dtype, 0, B, Hs, Hs, dzp, None, s2, C0 + C1, 0, [k.Seg(C0 + C1, out0=og)], ws2)
    assert rel_err(from_nhwc(og), dA) <= TOL_F32_OUT[dtype]
    # wgrad: plain = two-source input (64 + 64 inside one row tile), gathered = padded gradient, compact to 1
    wz = torch.zeros(C0 + C1, 1, 4, 4, requires_grad=True)
    F.conv_transpose2d(a, wz, stride=2, padding=1).backward(dzr)
    refw = wz.grad.permute(0, 2, 3, 1).reshape(C0 + C1, 16, 1)
    dw = torch.empty(C0 + C1, 16, 1, dtype=torch.float32, device=DEV)
    nbytes = k.wgrad_workspace_bytes(dtype, B, Hs, Hs, C0, C1, epc, 0, 1)
    wsw = torch.empty(max(nbytes, 16) // 4, dtype=torch.float32, device=DEV)
    k.wgrad(dtype, B, Hs, Hs, a0, a1, dzp, None, dw, wsw, c_valid=1)
    assert rel_err(dw, refw) <= TOL_F32_OUT[dtype]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_narrow_channel_counts_on_mfma(dtype):
    """Cin = 16 / 32: several taps inside one K-step of the implicit GEMM (per-chunk tap decode)."""
    k = K()
    torch.manual_seed(10)
    for Cin in (16, 32):
        B, N, Hs = 2, 64, 8
        x = rounded(torch.randn(B, Cin, 2 * Hs, 2 * Hs), dtype)
        w = rounded(torch.randn(N, Cin, 4, 4) * 0.1, dtype)
        ref = F.conv2d(x, w, stride=2, padding=1)
        master = w.permute(0, 2, 3, 1).contiguous().to(DEV)
        s2 = torch.empty(N, 16, Cin, dtype=dtype, device=DEV)
        t2 = torch.empty(4, Cin, 4, N, dtype=dtype, device=DEV)
        k.pack_weights(master, N, Cin, dtype, s2, t2)
        out = torch.empty(B, Hs, Hs, N, dtype=torch.float32, device=DEV)
        _, ws = ws_for(dtype, 0, B, Hs, Hs, Cin, 0, N, [N])
        k.igemm(dtype, 0, B, Hs, Hs, nhwc(x, dtype), None, s2, N, 0, [k.Seg(N, out0=out)], ws)
        assert rel_err(from_nhwc(out), ref) <= TOL_F32_OUT[dtype]
        # T2 with the same narrow input: transposed conv Cin -> 64
        wt = rounded(torch.randn(Cin, N, 4, 4) * 0.1, dtype)
        reft = F.conv_transpose2d(x, wt, stride=2, padding=1)
        mt = wt.permute(0, 2, 3, 1).contiguous().to(DEV)
        t2b = torch.empty(4, N, 4, Cin, dtype=dtype, device=DEV)
        k.pack_weights(mt, Cin, N, dtype, None, t2b)
        outt = torch.empty(B, 4 * Hs, 4 * Hs, N, dtype=torch.float32, device=DEV)
        _, wst = ws_for(dtype, 1, B, 2 * Hs, 2 * Hs, Cin, 0, N, [N])
        k.igemm(dtype, 1, B, 2 * Hs, 2 * Hs, nhwc(x, dtype), None, t2b, N, 0, [k.Seg(N, out0=outt)], wst)
        assert rel_err(from_nhwc(outt), reft) <= TOL_F32_OUT[dtype]


# ---- dedicated bf16 kernels of the thin layers (csrc/edge.hip): planar f32 thin operand, one MFMA per 16 pixels ----
BF = torch.bfloat16


@pytest.mark.parametrize('shape', [(2, 16, 32), (3, 8, 16), (1, 64, 64)])
def test_l0_forward_kernel(shape):
    """First conv 2 -> 64 (unetbaseline_model.py:187-191 outermost downconv): leaky + relu copies, all borders."""
    k = K()
    B, Hs, Ws = shape
    torch.manual_seed(21)
    x = torch.randn(B, 2, 2 * Hs, 2 * Ws)
    w = rounded(torch.randn(64, 2, 4, 4) * 0.1, BF)
    ref = F.conv2d(rounded(x, BF), w, stride=2, padding=1)
    master = w.permute(0, 2, 3, 1).contiguous().to(DEV)                # [64][kh][kw][2] = parameter memory
    lk = torch.empty(B, Hs, Ws, 64, dtype=BF, device=DEV)
    rl = torch.empty_like(lk)
    k.l0_forward(x.to(DEV), master, B, Hs, Ws, 0.2, lk, rl)
    assert rel_err(from_nhwc(lk), F.leaky_relu(ref, 0.2)) <= TOL_T_OUT[BF]
    assert rel_err(from_nhwc(rl), F.relu(ref)) <= TOL_T_OUT[BF]
    k.l0_forward(x.to(DEV), master, B, Hs, Ws, 0.2, None, rl)          # a single output is allowed
    assert rel_err(from_nhwc(rl), F.relu(ref)) <= TOL_T_OUT[BF]


@pytest.mark.parametrize('shape', [(2, 16, 32), (3, 8, 16), (1, 64, 64)])
@pytest.mark.parametrize('stats', [True, False, 'zmask'])
def test_d0_dgrad_kernel(shape, stats):
    """Input gradient of the last transposed conv 128 -> 1: ReLU masks of both halves + BN-backward sums of the up half."""
    k = K()
    B, Hs, Ws = shape
    torch.manual_seed(22)
    dz = torch.randn(B, 1, 2 * Hs, 2 * Ws)
    wt = rounded(torch.randn(128, 1, 4, 4) * 0.05, BF)
    dA = F.conv2d(rounded(dz, BF), wt, stride=2, padding=1)              # [B,128,Hs,Ws]
    master = wt.permute(0, 2, 3, 1).contiguous().view(-1).to(DEV)       # [128][16]
    ref0 = rounded(torch.randn(B, 64, Hs, Ws), BF)
    z1 = rounded(torch.randn(B, 64, Hs, Ws), BF)
    if stats == 'zmask':       # ref1 as the forward's apply kernel writes it: the kernel takes the mask from z, scale, shift
        scale, shift = torch.randn(64) * 0.5 + 1.0, torch.randn(64) * 0.3
        scale[::5] *= -1.0
        ref1 = rounded(F.relu(torch.addcmul(shift.view(1, -1, 1, 1), z1, scale.view(1, -1, 1, 1))), BF)
    else:
        ref1 = rounded(torch.randn(B, 64, Hs, Ws), BF)
    mean, istd = torch.randn(64) * 0.1, torch.rand(64) + 0.5
    g0 = dA[:, :64] * (ref0 > 0)
    g1 = dA[:, 64:] * (ref1 > 0)
    o0 = torch.empty(B, Hs, Ws, 64, dtype=BF, device=DEV)
    o1 = torch.empty_like(o0)
    P = k.d0_dgrad_num_partials(B, Hs, Ws)
    part = torch.full((P, 2, 64), float('nan'), device=DEV)
    s0 = k.Seg(64, out0=o0, ref=nhwc(ref0, BF), slope=0.0)
    if stats:
        extra = dict(scale=scale.to(DEV), shift=shift.to(DEV)) if stats == 'zmask' else {}
        s1 = k.Seg(64, out0=o1, ref=nhwc(ref1, BF), slope=0.0, z=nhwc(z1, BF), mean=mean.to(DEV), istd=istd.to(DEV),
                   partials=part, **extra)
    else:
        s1 = k.Seg(64, out0=o1, ref=nhwc(ref1, BF), slope=0.0)
    k.d0_dgrad(dz.to(DEV), master, B, Hs, Ws, s0, s1)
    assert rel_err(from_nhwc(o0), g0) <= TOL_T_OUT[BF]
    assert rel_err(from_nhwc(o1), g1) <= TOL_T_OUT[BF]
    if stats:
        tot = part.sum(0).cpu()
        xh = (z1 - mean.view(1, -1, 1, 1)) * istd.view(1, -1, 1, 1)
        assert rel_err(tot[0], g1.sum((0, 2, 3))) <= 1e-3
        assert rel_err(tot[1], (g1 * xh).sum((0, 2, 3))) <= 1e-3


@pytest.mark.parametrize('shape', [(2, 16, 32), (3, 8, 32), (1, 64, 64)])
def test_thin_wgrad_kernels(shape):
    """Weight gradients against the thin operand: last transposed conv (thin = dz) and first conv (thin = input)."""
    k = K()
    B, Hs, Ws = shape
    torch.manual_seed(23)
    # (a) ConvTranspose2d 128 -> 1: dW[c][kh][kw] = sum in[b,c,i,j] * dz[b,0,2i-1+kh,2j-1+kw]
    a = rounded(torch.randn(B, 128, Hs, Ws), BF)
    dz = torch.randn(B, 1, 2 * Hs, 2 * Ws)
    wz = torch.zeros(128, 1, 4, 4, requires_grad=True)
    F.conv_transpose2d(a, wz, stride=2, padding=1).backward(rounded(dz, BF))
    refw = wz.grad.permute(0, 2, 3, 1).reshape(128, 16)
    dw = torch.full((128, 16), float('nan'), device=DEV)
    ws = torch.empty(k.thin_wgrad_workspace_bytes(B, Hs, Ws, 1, 64, 64) // 4, device=DEV)
    k.thin_wgrad(dz.to(DEV), nhwc(a[:, :64], BF), nhwc(a[:, 64:], BF), B, Hs, Ws, dw, ws)
    assert rel_err(dw, refw) <= TOL_F32_OUT[BF]
    dw2 = torch.empty_like(dw)
    k.thin_wgrad(dz.to(DEV), nhwc(a[:, :64], BF), nhwc(a[:, 64:], BF), B, Hs, Ws, dw2, ws)
    assert torch.equal(dw, dw2)                                          # fixed-order sums: bit-reproducible
    # (b) Conv2d 2 -> 64: dW[o][kh][kw][ci] = sum g[b,o,oy,ox] * x[b,ci,2oy-1+kh,2ox-1+kw]
    x = torch.randn(B, 2, 2 * Hs, 2 * Ws)
    g = rounded(torch.randn(B, 64, Hs, Ws), BF)
    wc = torch.zeros(64, 2, 4, 4, requires_grad=True)
    F.conv2d(rounded(x, BF), wc, stride=2, padding=1).backward(g)
    refc = wc.grad.permute(0, 2, 3, 1).reshape(64, 32)
    dwc = torch.full((64, 32), float('nan'), device=DEV)
    ws2 = torch.empty(k.thin_wgrad_workspace_bytes(B, Hs, Ws, 2, 64, 0) // 4, device=DEV)
    k.thin_wgrad(x.to(DEV), nhwc(g, BF), None, B, Hs, Ws, dwc, ws2)
    assert rel_err(dwc, refc) <= TOL_F32_OUT[BF]


def test_edge_kernels_reject_other_shapes():
    k = K()
    x = torch.zeros(1, 3, 32, 32, device=DEV)
    w = torch.zeros(64 * 16 * 3, device=DEV)
    o = torch.empty(1, 16, 16, 64, dtype=BF, device=DEV)
    with pytest.raises(RuntimeError, match='2 -> 64'):
        k.l0_forward(x, w, 1, 16, 16, 0.2, o, None)
    with pytest.raises(RuntimeError, match='Ws % 16'):
        k.l0_forward(torch.zeros(1, 2, 16, 16, device=DEV), torch.zeros(2048, device=DEV), 1, 8, 8, 0.2,
                     torch.empty(1, 8, 8, 64, dtype=BF, device=DEV), None)
