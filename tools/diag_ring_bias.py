"""Is the ring kernel's output systematically different from the patch kernel's on identical inputs?  (children: one per knob)"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

B, C0, C1, N, Hs = 16, 256, 256, 128, 32


def data():
    g = torch.Generator().manual_seed(7)
    x = torch.relu(torch.randn(B, C0 + C1, Hs, Hs, generator=g)).bfloat16().float()
    w = (torch.randn(C0 + C1, N, 4, 4, generator=g) * 0.02).bfloat16().float()
    return x, w


def child(path):
    from audio_depth_estimation_amd import kernels as K
    x, w = data()
    T = torch.bfloat16
    X, Y = w.shape[:2]
    master = w.permute(0, 2, 3, 1).contiguous().cuda()
    s2 = torch.empty(X, 16, Y, dtype=T, device='cuda')
    t2 = torch.empty(4, Y, 4, X, dtype=T, device='cuda')
    K.pack_weights(master, X, Y, T, s2, t2)
    in0 = x[:, :C0].permute(0, 2, 3, 1).contiguous().to(T).cuda()
    in1 = x[:, C0:].permute(0, 2, 3, 1).contiguous().to(T).cuda()
    P, wsb = K.igemm_query(T, 1, B, Hs, Hs, C0, C1, N, [N], epi=1)
    ws = torch.empty(max(wsb, 16) // 4, device='cuda')
    z = torch.empty(B, 2 * Hs, 2 * Hs, N, dtype=T, device='cuda')
    part = torch.zeros(P, 2, N, device='cuda')
    K.igemm(T, 1, B, Hs, Hs, in0, in1, t2, N, 1, [K.Seg(N, out0=z, partials=part)], ws)
    torch.cuda.synchronize()
    torch.save({'z': z.float().cpu(), 's1': part[:, 0].double().sum(0).cpu(), 's2': part[:, 1].double().sum(0).cpu(), 'P': P}, path)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == 'child':
        return child(sys.argv[2])
    x, w = data()
    ref = F.conv_transpose2d(x.double(), w.double(), stride=2, padding=1).permute(0, 2, 3, 1)     # NHWC f64
    res = {}
    for ring in ('0', '1'):
        path = f'/tmp/ringbias_{ring}.pt'
        subprocess.run([sys.executable, __file__, 'child', path], check=True, env=dict(os.environ, ADN_IGEMM_RING=ring),
                       stdout=subprocess.DEVNULL)
        res[ring] = torch.load(path)
    refb = ref.float().bfloat16().float()          # the correctly rounded result
    for ring in ('0', '1'):
        z = res[ring]['z']
        d = (z.double() - ref)
        print(f'ring={ring} P={res[ring]["P"]}: mean signed err {float(d.mean()):+.3e}  rms err {float(d.pow(2).mean().sqrt()):.3e}  (ref rms {float(ref.pow(2).mean().sqrt()):.3e})'
              f'  != correctly rounded: {float((z != refb).float().mean()):.4f}'
              f'  s1 rel err {float(((res[ring]["s1"] - ref.sum((0, 1, 2))).abs().max()) / ref.sum((0, 1, 2)).abs().max()):.2e}'
              f'  s2 rel err {float(((res[ring]["s2"] - ref.pow(2).sum((0, 1, 2))).abs().max()) / ref.pow(2).sum((0, 1, 2)).abs().max()):.2e}')
    a, b = res['0']['z'], res['1']['z']
    print('ring vs patch: fraction of elements that differ', float((a != b).float().mean()), ' mean signed diff', float((b - a).double().mean()))


if __name__ == '__main__':
    main()
