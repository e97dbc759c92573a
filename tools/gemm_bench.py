"""Micro-benchmark of single implicit-GEMM / wgrad launches at the unet_256 B=32 layer shapes (tuning aid).

    python tools/gemm_bench.py [--iters 20] [--only NAME]
Prints TFLOP/s per shape from HIP events around `iters` back-to-back launches (random bf16 data).
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from audio_depth_estimation_amd import kernels as K  # noqa: E402

DEV = 'cuda'
T = torch.bfloat16
B = 32
# name, geom, Hs, C0, C1, N
IGEMM = [
    ('L1_fwd', 0, 64, 64, 0, 128), ('L2_fwd', 0, 32, 128, 0, 256), ('L3_fwd', 0, 16, 256, 0, 512),
    ('L4_fwd', 0, 8, 512, 0, 512),
    ('D1_fwd', 1, 64, 128, 128, 64), ('D2_fwd', 1, 32, 256, 256, 128), ('D3_fwd', 1, 16, 512, 512, 256),
    ('D4_fwd', 1, 8, 512, 512, 512),
    ('D1_dgrad', 0, 64, 64, 0, 256), ('D2_dgrad', 0, 32, 128, 0, 512), ('L2_dgrad', 1, 32, 256, 0, 128),
    ('L1_dgrad', 1, 64, 128, 0, 64),
]
# the deep (small-M, split-K) layers: --deep
DEEP = [('L5_fwd', 0, 4, 512, 0, 512), ('L6_fwd', 0, 2, 512, 0, 512), ('L7_fwd', 0, 1, 512, 0, 512),
        ('D7_fwd', 1, 1, 512, 0, 512), ('D6_fwd', 1, 2, 512, 512, 512), ('D5_fwd', 1, 4, 512, 512, 512),
        ('D4_fwd', 1, 8, 512, 512, 512), ('L4_fwd', 0, 8, 512, 0, 512),
        ('D5_dgrad', 0, 4, 512, 0, 1024), ('D6_dgrad', 0, 2, 512, 0, 1024), ('L5_dgrad', 1, 4, 512, 0, 512)]
# name, Hs, R0, R1, C
WGRAD = [('L1_wgrad', 64, 128, 0, 64), ('L2_wgrad', 32, 256, 0, 128), ('L3_wgrad', 16, 512, 0, 256),
         ('D1_wgrad', 64, 128, 128, 64), ('D2_wgrad', 32, 256, 256, 128), ('D3_wgrad', 16, 512, 512, 256)]


# stride-1 3x3 layers of the DoubleConv nets (RGBDepthNet, 256x256, B=32): name, H, C0, C1, N
S1_IGEMM = [('s1_inc2_fwd', 256, 64, 0, 64), ('s1_up4c1_fwd', 256, 64, 64, 64), ('s1_up4c1_dgrad', 256, 64, 0, 128),
            ('s1_d1c2_fwd', 128, 128, 0, 128), ('s1_up3c1_fwd', 128, 128, 128, 128), ('s1_d2c2_fwd', 64, 256, 0, 256),
            ('s1_d3c2_fwd', 32, 512, 0, 512)]
# name, H, R, C0, C1
S1_WGRAD = [('s1_inc2_wgrad', 256, 64, 64, 0), ('s1_up4c1_wgrad', 256, 64, 64, 64), ('s1_d1c2_wgrad', 128, 128, 128, 0),
            ('s1_up3c1_wgrad', 128, 128, 128, 128), ('s1_d2c2_wgrad', 64, 256, 256, 0), ('s1_d3c2_wgrad', 32, 512, 512, 0)]


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


_FLUSH = None


def timeit_cold(fn, iters):
    """Every launch timed on its own, a 1 GB fill in front of it (beyond L2 + the 256 MB memory-side cache): operands come
    from HBM as they do inside a training step."""
    global _FLUSH
    if _FLUSH is None:
        _FLUSH = torch.empty(1 << 28, device=DEV)
    fn()
    tot = 0.0
    for _ in range(iters):
        _FLUSH.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot * 1e-3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cold', action='store_true', help='flush the caches in front of every timed launch')
    ap.add_argument('--bwd-epi', action='store_true', help='run the *_dgrad shapes with the BWD epilogue (act mask, BN-backward sums)')
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--only', default=None)
    ap.add_argument('--attn', action='store_true', help='binaural cross-attention levels instead')
    ap.add_argument('--bwd', action='store_true')
    ap.add_argument('--b2', type=int, default=8)
    ap.add_argument('--s1', action='store_true', help='stride-1 3x3 shapes of the DoubleConv nets instead')
    ap.add_argument('--mx8', action='store_true', help='MX-fp8 3x3 conv beside the bf16 patch kernel (config 5 shapes)')
    ap.add_argument('--b', type=int, default=0, help='batch for --mx8 (default 8)')
    ap.add_argument('--deep', action='store_true', help='the small-M split-K layers of unet_256 instead')
    ap.add_argument('--edge', action='store_true', help='thin outermost layers of unet_256 with their real epilogues')
    args = ap.parse_args()
    torch.manual_seed(0)
    if args.cold:
        globals()['timeit'] = timeit_cold
    if args.mx8:
        return mx8_main(args)
    if args.s1:
        return s1_main(args)
    if args.attn:
        return attn_main(args)
    if args.edge:
        return edge_main(args)
    for name, geom, Hs, C0, C1, N in (DEEP if args.deep else IGEMM):
        if args.only and args.only not in name:
            continue
        hin = 2 * Hs if geom == 0 else Hs
        in0 = torch.randn(B, hin, hin, C0, device=DEV).to(T)
        in1 = torch.randn(B, hin, hin, C1, device=DEV).to(T) if C1 else None
        taps = 16 if geom == 0 else 4
        w = (torch.randn((1 if geom == 0 else 4) * N * taps * (C0 + C1), device=DEV) * 0.05).to(T)
        hout = Hs if geom == 0 else 2 * Hs
        out = torch.empty(B, hout, hout, N, device=DEV, dtype=T)
        bwd = args.bwd_epi and name.endswith('dgrad')
        P, wsb = K.igemm_query(T, geom, B, Hs, Hs, C0, C1, N, [N], epi=3 if bwd else 1)
        ws = torch.empty(max(wsb, 16) // 4, device=DEV)
        part = torch.empty(P * 2 * N, device=DEV)
        if bwd:
            ref, z = torch.randn_like(out), torch.randn_like(out)
            mean, istd = torch.zeros(N, device=DEV), torch.ones(N, device=DEV)
            fn = lambda: K.igemm(T, geom, B, Hs, Hs, in0, in1, w, N, 3,
                                 [K.Seg(N, out0=out, ref=ref, slope=0.2, z=z, mean=mean, istd=istd, partials=part,
                                        scale=istd, shift=mean)], ws)      # (scale / shift as the engine passes them: mask from z)
        else:
            fn = lambda: K.igemm(T, geom, B, Hs, Hs, in0, in1, w, N, 1, [K.Seg(N, out0=out, partials=part)], ws)
        t = timeit(fn, args.iters)
        fl = 2.0 * B * Hs * Hs * N * 16 * (C0 + C1)
        print(f'{name:10s} M={B*Hs*Hs*(1 if geom == 0 else 4):7d} N={N:4d} K={taps*(C0+C1):5d}  {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s', flush=True)
    for name, Hs, R0, R1, C in ([] if args.deep else WGRAD):
        if args.only and args.only not in name:
            continue
        p0 = torch.randn(B, Hs, Hs, R0, device=DEV).to(T)
        p1 = torch.randn(B, Hs, Hs, R1, device=DEV).to(T) if R1 else None
        g = torch.randn(B, 2 * Hs, 2 * Hs, C, device=DEV).to(T)
        dw = torch.empty((R0 + R1) * 16 * C, device=DEV)
        ws = torch.empty(max(K.wgrad_workspace_bytes(T, B, Hs, Hs, R0, R1, C, 0), 16) // 4, device=DEV)
        fn = lambda: K.wgrad(T, B, Hs, Hs, p0, p1, g, None, dw, ws)
        t = timeit(fn, args.iters)
        fl = 2.0 * B * Hs * Hs * (R0 + R1) * 16 * C
        print(f'{name:10s} M={B*Hs*Hs:7d} R={R0+R1:4d} C={C:4d}        {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s (incl. slab sum)', flush=True)


def edge_main(args):
    """D0 dgrad: the gradient of the 1-channel output (padded to 8) through the outermost ConvT, split into the skip
    half (ReLU mask) and the up half (ReLU mask + BN-backward statistics): K = 128, N = 128, M = 524288 -- all epilogue."""
    Hs = 128
    dz = torch.randn(B, 2 * Hs, 2 * Hs, 8, device=DEV).to(T)
    w = (torch.randn(128 * 16 * 8, device=DEV) * 0.05).to(T)
    mk = lambda: torch.randn(B, Hs, Hs, 64, device=DEV).to(T)
    Gd, Gu, rd, ru, zu = mk(), mk(), mk(), mk(), mk()
    mean, istd = torch.zeros(64, device=DEV), torch.ones(64, device=DEV)
    P, wsb = K.igemm_query(T, 0, B, Hs, Hs, 8, 0, 128, [64, 64])
    ws = torch.empty(max(wsb, 16) // 4, device=DEV)
    part = torch.empty(P * 2 * 64, device=DEV)
    segs = [K.Seg(64, out0=Gd, ref=rd, slope=0.0),
            K.Seg(64, out0=Gu, ref=ru, slope=0.0, z=zu, mean=mean, istd=istd, partials=part)]
    fn = lambda: K.igemm(T, 0, B, Hs, Hs, dz, None, w, 128, 3, segs, ws)
    t = timeit(fn, args.iters)
    byts = B * Hs * Hs * 64 * 2 * 5 + dz.numel() * 2
    print(f'D0_dgrad_bwd M={B*Hs*Hs} N=128 K=128  {t*1e6:8.1f} us  {byts/t/1e12:6.2f} TB/s (algorithmic bytes)', flush=True)
    # L0 forward: 2 -> 64 channels (input padded to 8), K = 128, N = 64, raw output + leaky copy
    x = torch.randn(B, 2 * Hs, 2 * Hs, 8, device=DEV).to(T)
    w0 = (torch.randn(64 * 16 * 8, device=DEV) * 0.05).to(T)
    o0, o1 = mk(), mk()
    P, wsb = K.igemm_query(T, 0, B, Hs, Hs, 8, 0, 64, [64])
    ws = torch.empty(max(wsb, 16) // 4, device=DEV)
    sc, sh = torch.ones(64, device=DEV), torch.zeros(64, device=DEV)
    fn = lambda: K.igemm(T, 0, B, Hs, Hs, x, None, w0, 64, 2, [K.Seg(64, out0=o0, out1=o1, scale=sc, shift=sh, slope=0.2)], ws)
    t = timeit(fn, args.iters)
    byts = B * Hs * Hs * 64 * 2 * 2 + x.numel() * 2
    print(f'L0_fwd_act   M={B*Hs*Hs} N= 64 K=128  {t*1e6:8.1f} us  {byts/t/1e12:6.2f} TB/s (algorithmic bytes)', flush=True)


def mx8_main(args):
    """MX-fp8 3 x 3 conv (csrc/mx8.hip) beside the bf16 patch kernel on the same shapes (config 5: B 8, 512 x 512)."""
    Bm = args.b if args.b else 8
    shapes = [('inc2', 512, 64, 0, 64), ('up4c1', 512, 64, 64, 64), ('up4dg', 512, 64, 0, 128), ('d1c2', 256, 128, 0, 128),
              ('up3c1', 256, 128, 128, 128), ('d2c2', 128, 256, 0, 256), ('d3c2', 64, 512, 0, 512), ('d4c2', 32, 512, 0, 512)]
    for name, H, C0, C1, N in shapes:
        if args.only and args.only not in name:
            continue
        Cin = C0 + C1
        in0 = torch.randn(Bm, H, H, C0, device=DEV).to(T)
        in1 = torch.randn(Bm, H, H, C1, device=DEV).to(T) if C1 else None
        master = torch.randn(N, 9, Cin, device=DEV) * 0.05
        out = torch.empty(Bm, H, H, N, device=DEV, dtype=T)
        P, wsb = K.igemm_query(T, K.GEMM_S1, Bm, H, H, C0, C1, N, [N], ks=3)
        ws = torch.empty(max(wsb, 16) // 4, device=DEV)
        part = torch.empty(max(P, K.conv3x3_mx8_num_partials(Bm, H, H, N, C0, C1)) * 2 * N, device=DEV)
        w16 = torch.empty(N, K.s1_row_stride(T, 9, Cin), device=DEV, dtype=T)
        K.pack_rows(master, N, 9, Cin, w16)
        f16 = lambda: K.igemm(T, K.GEMM_S1, Bm, H, H, in0, in1, w16, N, 1, [K.Seg(N, out0=out, partials=part)], ws, ks=3)
        q0, s0 = torch.empty(in0.shape, dtype=torch.uint8, device=DEV), torch.empty(Bm, H, H, C0 // 32, dtype=torch.uint8, device=DEV)
        K.mx8_quantize(in0, q0, s0)
        q1 = s1 = None
        if C1:
            q1, s1 = torch.empty(in1.shape, dtype=torch.uint8, device=DEV), torch.empty(Bm, H, H, C1 // 32, dtype=torch.uint8, device=DEV)
            K.mx8_quantize(in1, q1, s1)
        s8, ssc = K.mx8_pack_shapes(N, Cin, False)
        w8, wsc = torch.empty(s8, dtype=torch.uint8, device=DEV), torch.empty(ssc, dtype=torch.uint8, device=DEV)
        K.mx8_pack(master, N, Cin, False, w8, wsc)
        f8 = lambda: K.conv3x3_mx8(Bm, H, H, q0, s0, q1, s1, w8, wsc, N, 1, [K.Seg(N, out0=out, partials=part)])
        fq = lambda: K.mx8_quantize(in0, q0, s0)
        t16, t8, tq = timeit(f16, args.iters), timeit(f8, args.iters), timeit(fq, args.iters)
        fl = 2.0 * Bm * H * H * N * 9 * Cin
        print(f'{name:8s} M={Bm*H*H:8d} N={N:4d} K={9*Cin:5d}  bf16 {t16*1e6:8.1f} us {fl/t16/1e12:7.1f} TF/s | mx-fp8 {t8*1e6:8.1f} us '
              f'{fl/t8/1e12:7.1f} TF/s ({t16/t8:4.2f}x) | quantize in0 {tq*1e6:7.1f} us', flush=True)


def s1_main(args):
    for name, H, C0, C1, N in S1_IGEMM:
        if args.only and args.only not in name:
            continue
        in0 = torch.randn(B, H, H, C0, device=DEV).to(T)
        in1 = torch.randn(B, H, H, C1, device=DEV).to(T) if C1 else None
        w = (torch.randn(N, K.s1_row_stride(T, 9, C0 + C1), device=DEV) * 0.05).to(T)
        out = torch.empty(B, H, H, N, device=DEV, dtype=T)
        P, wsb = K.igemm_query(T, K.GEMM_S1, B, H, H, C0, C1, N, [N], ks=3)
        ws = torch.empty(max(wsb, 16) // 4, device=DEV)
        part = torch.empty(P * 2 * N, device=DEV)
        fn = lambda: K.igemm(T, K.GEMM_S1, B, H, H, in0, in1, w, N, 1, [K.Seg(N, out0=out, partials=part)], ws, ks=3)
        t = timeit(fn, args.iters)
        fl = 2.0 * B * H * H * N * 9 * (C0 + C1)
        print(f'{name:15s} M={B*H*H:8d} N={N:4d} K={9*(C0+C1):5d}  {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s', flush=True)
    for name, H, R, C0, C1 in S1_WGRAD:
        if args.only and args.only not in name:
            continue
        p0 = torch.randn(B, H, H, R, device=DEV).to(T)
        g0 = torch.randn(B, H, H, C0, device=DEV).to(T)
        g1 = torch.randn(B, H, H, C1, device=DEV).to(T) if C1 else None
        dw = torch.empty(R * 9 * (C0 + C1), device=DEV)
        ws = torch.empty(max(K.wgrad_workspace_bytes(T, B, H, H, R, 0, C0, C1, ks=3), 16) // 4, device=DEV)
        fn = lambda: K.wgrad(T, B, H, H, p0, None, g0, g1, dw, ws, ks=3)
        t = timeit(fn, args.iters)
        fl = 2.0 * B * H * H * R * 9 * (C0 + C1)
        print(f'{name:15s} M={B*H*H:8d} R={R:4d} C={C0+C1:4d}        {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s (incl. slab sum)',
              flush=True)


# binaural cross-attention levels at 256x256 input: name, N, dqk, dv (B2 = stacked batch entries)
ATTN = [('attn_L2', 16384, 16, 128), ('attn_L3', 4096, 32, 256), ('attn_L4', 1024, 64, 512), ('attn_L5', 256, 64, 512)]


def attn_main(args):
    B2 = args.b2
    for name, N, dqk, dv in ATTN:
        if args.only and args.only not in name:
            continue
        ld = (2 * dqk + dv + 63) // 64 * 64
        qkv = torch.randn(B2, N, ld, device=DEV).to(T)
        q, k, v = qkv[:, :, :dqk], qkv[:, :, dqk:2 * dqk], qkv[:, :, 2 * dqk:2 * dqk + dv]
        o = torch.empty(B2, N, dv, device=DEV, dtype=T)
        lse = torch.empty(B2, N, device=DEV)
        sc = 1.0 / dv ** 0.5
        fn = lambda: K.attn_fwd(q, k, v, o, lse, dqk, dv, B2 // 2, sc)
        t = timeit(fn, args.iters)
        fl = 2.0 * B2 * N * N * (dqk + dv)          # S = QK^T and O = PV
        print(f'{name}_fwd B2={B2:3d} N={N:6d} dqk={dqk:3d} dv={dv:4d}  {t*1e6:10.1f} us  {fl/t/1e12:7.1f} TF/s', flush=True)
        if args.bwd:
            do = torch.randn(B2, N, dv, device=DEV).to(T)
            dqkv = torch.zeros(B2, N, ld, device=DEV, dtype=T)
            ws = torch.empty(B2 * N, device=DEV)
            fnb = lambda: K.attn_bwd(q, k, v, o, lse, dqk, dv, B2 // 2, sc, do, dqkv[:, :, :dqk], dqkv[:, :, dqk:2 * dqk],
                                     dqkv[:, :, 2 * dqk:2 * dqk + dv], ws)
            t = timeit(fnb, args.iters)
            flb = 2.0 * B2 * N * N * (3 * dqk + 2 * dv)
            print(f'{name}_bwd B2={B2:3d} N={N:6d} dqk={dqk:3d} dv={dv:4d}  {t*1e6:10.1f} us  {flb/t/1e12:7.1f} TF/s (algorithmic)',
                  flush=True)


if __name__ == '__main__':
    main()
