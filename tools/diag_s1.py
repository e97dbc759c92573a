"""Repeat-determinism and patch-vs-tap-staged comparison of the S1 implicit GEMM at the DoubleConv nets' full sizes.
Run once with ADN_IGEMM_PATCH=0 (writes /tmp/s1_ref_*.pt) and once with the default (compares)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_depth_estimation_amd import kernels as K

DEV, T = 'cuda', torch.bfloat16
B = int(os.environ.get('B', 32))
SHAPES = [('inc2', 256, 64, 0, 64), ('up4c1', 256, 64, 64, 64), ('up4dg', 256, 64, 0, 128), ('d1c2', 128, 128, 0, 128),
          ('up3c1', 128, 128, 128, 128), ('d2c2', 64, 256, 0, 256), ('d3c2', 32, 512, 0, 512)]
patch = os.environ.get('ADN_IGEMM_PATCH', '1') != '0'
for name, H, C0, C1, N in SHAPES:
    g = torch.Generator(device=DEV).manual_seed(5)
    in0 = torch.randn(B, H, H, C0, device=DEV, generator=g).to(T)
    in1 = torch.randn(B, H, H, C1, device=DEV, generator=g).to(T) if C1 else None
    w = (torch.randn(N, K.s1_row_stride(T, 9, C0 + C1), device=DEV, generator=g) * 0.05).to(T)
    P, wsb = K.igemm_query(T, K.GEMM_S1, B, H, H, C0, C1, N, [N], ks=3)
    ws = torch.empty(max(wsb, 16) // 4, device=DEV)
    outs, parts = [], []
    for r in range(4):
        out = torch.full((B, H, H, N), float('nan'), device=DEV, dtype=T)
        part = torch.full((P * 2 * N,), float('nan'), device=DEV)
        K.igemm(T, K.GEMM_S1, B, H, H, in0, in1, w, N, 1, [K.Seg(N, out0=out, partials=part)], ws, ks=3)
        torch.cuda.synchronize()
        outs.append(out)
        parts.append(part)
    for r in range(1, 4):
        eq = torch.equal(outs[0].view(torch.int16), outs[r].view(torch.int16))
        peq = torch.equal(parts[0].view(torch.int32), parts[r].view(torch.int32))
        msg = f'{name}: repeat {r} out equal={eq} partials equal={peq} P={P}'
        if not eq:
            d = (outs[0].float() - outs[r].float()).abs()
            idx = torch.nonzero(d.amax(dim=3) > 0)
            msg += f' n_bad_pixels={idx.shape[0]} max={float(d.max()):.3f} first={idx[:6].tolist()}'
        print(msg, flush=True)
    print(f'{name}: nan in out={bool(torch.isnan(outs[0].float()).any())} nan in partials={bool(torch.isnan(parts[0]).any())}', flush=True)
    f = f'/tmp/s1_ref_{name}.pt'
    if not patch:
        torch.save((outs[0].cpu(), parts[0].cpu()), f)
    elif os.path.exists(f):
        ro, rp = torch.load(f)
        d = (outs[0].cpu().float() - ro.float()).abs()
        idx = torch.nonzero(d.amax(dim=3) > 0.05)
        print(f'{name}: vs tap-staged max diff {float(d.max()):.4f} bad pixels (>0.05) {idx.shape[0]} first {idx[:8].tolist()}', flush=True)
