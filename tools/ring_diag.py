"""Where a K-step of the ring-fed implicit GEMM spends its cycles (diagnostic build with s_memtime stamps).

    make -C audio-depth-estimation_amd/csrc diag && ADN_LIB=audio-depth-estimation_amd/libadn_diag.so python tools/ring_diag.py

Per layer shape: mean cycles per K-step and wave in (a) the counted wait + barrier, (b) issuing the LDS-DMA requests,
(c) / (d) the two taps (16 MFMAs + 8 fragment reads each; 256 MFMA cycles per wave, two waves per SIMD), and per tile in the
epilogue.  The stamps fence the scheduler, so read the SHARES, not the total.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from audio_depth_estimation_amd import kernels as K  # noqa: E402

DEV, T, B = 'cuda', torch.bfloat16, 32
SHAPES = [('D1_fwd', 1, 64, 128, 128, 64), ('L1_fwd', 0, 64, 64, 0, 128), ('L2_fwd', 0, 32, 128, 0, 256), ('L3_fwd', 0, 16, 256, 0, 512),
          ('D2_fwd', 1, 32, 256, 256, 128), ('D3_fwd', 1, 16, 512, 512, 256), ('D2_dgrad', 0, 32, 128, 0, 512)]


def main():
    torch.manual_seed(0)
    for name, geom, Hs, C0, C1, N in SHAPES:
        hin = 2 * Hs if geom == 0 else Hs
        in0 = torch.randn(B, hin, hin, C0, device=DEV).to(T)
        in1 = torch.randn(B, hin, hin, C1, device=DEV).to(T) if C1 else None
        taps = 16 if geom == 0 else 4
        w = (torch.randn((1 if geom == 0 else 4) * N * taps * (C0 + C1), device=DEV) * 0.05).to(T)
        hout = Hs if geom == 0 else 2 * Hs
        out = torch.empty(B, hout, hout, N, device=DEV, dtype=T)
        P, _ = K.igemm_query(T, geom, B, Hs, Hs, C0, C1, N, [N])
        ws = torch.zeros(256 * 8 * 16 * 2, device=DEV)                # 256 workgroups x 8 waves x 16 u64
        part = torch.empty(P * 2 * N, device=DEV)
        for _ in range(3):
            K.igemm(T, geom, B, Hs, Hs, in0, in1, w, N, 1, [K.Seg(N, out0=out, partials=part)], ws)
        torch.cuda.synchronize()
        r = ws.view(torch.int64).view(256, 8, 16).cpu().double()
        r = r[r[:, :, 5] > 0]
        if r.numel() == 0:
            print(name, 'no stamps (not the diagnostic build, or the ring kernel did not run)')
            continue
        n = r[:, 5]
        wait, issue, t0, t1 = [(r[:, i] / n).mean().item() for i in range(4)]
        epi = (r[:, 4] / r[:, 7]).mean().item()
        ep = [(r[:, i] / r[:, 7]).mean().item() for i in (9, 10, 11, 12)]
        tot = r[:, 6].mean().item()
        steps = n.mean().item()
        print(f'{name:9s} steps/wave {steps:6.0f}  per step: wait+barrier {wait:7.0f}  issue {issue:6.0f}  tap0 {t0:6.0f}  tap1 {t1:6.0f}'
              f'  = {wait + issue + t0 + t1:7.0f} cyc | epilogue/tile {epi:7.0f} (stores {ep[0]:.0f} stats {ep[1]:.0f} barrier {ep[2]:.0f} final {ep[3]:.0f}) | kernel {tot:9.0f} cyc, clock {(r[:, 6] / r[:, 8]).mean().item() * 100:.0f} MHz', flush=True)


if __name__ == '__main__':
    main()
