"""Which layer differs between the ring-fed and the patch-staged implicit GEMM inside the full unet_256 step (B = 32, bf16)?
Runs the step in two child processes (ADN_IGEMM_RING=0 / 1: the knob is read once per process), saves activations-free
summaries (per-parameter gradients, prediction) and prints the per-tensor relative difference in parameter order."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(path):
    from types import SimpleNamespace

    import torch
    from audio_depth_estimation_amd.engine import FusedTrainer
    from audio_depth_estimation_amd.models.unetbaseline_model import define_G
    torch.manual_seed(0)
    m = define_G(SimpleNamespace(dataset=SimpleNamespace(depth_norm=False, max_depth=30.0)), 2, 1, 64, 'unet_256')
    m.compute_dtype = torch.bfloat16
    m = m.to('cuda')
    with torch.no_grad():
        m.model.model[3].bias.fill_(1.0)
    m.train()
    eng = m.engine()
    g = torch.Generator().manual_seed(1234)
    B = int(os.environ.get('DIAG_B', '32'))
    audio = torch.rand(B, 2, 256, 256, generator=g).to('cuda')
    gt = 30 * torch.rand(B, 1, 256, 256, generator=g)
    gt[gt < 3] = 0
    tr = FusedTrainer(eng, 'Combined', 0.237, 0.637, 0.869, max_depth=30.0, optimizer='AdamW', lr=0.002, clip_norm=1.0)
    loss, pred = tr.step(audio, gt.to('cuda'))
    out = {'pred': pred.float().cpu(), 'loss': float(loss)}
    for k, prm in m.named_parameters():
        out['g/' + k] = eng.grad_view(prm).detach().float().cpu().clone()
    gi = torch.Generator().manual_seed(3)
    for i, lv in enumerate(eng.levels):
        for name in ('zd', 'ad', 'rd', 'zu', 'ru', 'Gd', 'Gu', 'part_d', 'part_u', 'bpart_d', 'bpart_u', 'mean_d', 'istd_d',
                     'mean_u', 'istd_u'):
            t = lv.get(name)
            if torch.is_tensor(t):
                flat = t.detach().reshape(-1)
                stride = max(1, flat.numel() // 65536) | 1
                out[f'L{i}.{name}'] = flat[::stride].float().cpu() if not name.startswith(('part', 'bpart')) else flat.float().cpu()
    out['FULLF.dz0'] = eng.levels[0]['dz0'].detach().float().cpu()
    out['FULLF.out'] = eng.levels[0]['out'].detach().float().cpu()
    if os.environ.get('DIAG_FULL'):
        for i in (1, 2, 3):
            for name in ('zd', 'zu'):
                out[f'FULL.L{i}.{name}'] = eng.levels[i][name].detach().cpu()
    torch.save(out, path)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == 'child':
        return child(sys.argv[2])
    os.makedirs('gpurun_out/r3', exist_ok=True)
    res = []
    variants = [('0', {'ADN_IGEMM_RING': '0'}), (os.environ.get('DIAG_TAG', '1'), {})]
    if os.environ.get('DIAG_BASE_ENV'):            # perturb the BASE run too (e.g. another tile rule): 'A=1,B=2'
        variants[0][1].update(dict(kv.split('=') for kv in os.environ['DIAG_BASE_ENV'].split(',')))
    for ring, extra in variants:
        path = f'/tmp/diag_ring_{ring}.pt'
        env = dict(os.environ, **extra)
        subprocess.run([sys.executable, __file__, 'child', path], check=True, env=env)
        import torch
        res.append(torch.load(path))
    a, b = res
    print('loss', a['loss'], b['loss'])
    for k in a:
        if k == 'loss':
            continue
        x, y = a[k], b[k]
        if x.shape != y.shape:            # partial-row buffers have another row count: compare the column sums
            print(f'{k:70s} shapes {tuple(x.shape)} {tuple(y.shape)}')
            continue
        if k.startswith('FULLF.'):
            xf, yf = x.double().reshape(-1), y.double().reshape(-1)
            top = xf.abs().topk(5)
            print(f'{k:12s} norms {float(xf.norm()):.5e} {float(yf.norm()):.5e}  rel diff {float((xf - yf).norm() / xf.norm()):.3e}  max|x| {float(xf.abs().max()):.4e} '
                  f'max|y| {float(yf.abs().max()):.4e}; top-5 |x| {[f"{v:.3e}" for v in top.values.tolist()]} y there {[f"{v:.3e}" for v in yf[top.indices].tolist()]}')
            if k.endswith('out'):
                print('   out: min', float(xf.min()), float(yf.min()), ' count(out < 1e-2)', int((xf < 1e-2).sum()), int((yf < 1e-2).sum()),
                      ' count(0 < out < 1e-3)', int(((xf > 0) & (xf < 1e-3)).sum()), int(((yf > 0) & (yf < 1e-3)).sum()))
            continue
        if k.startswith('FULL.'):
            xf, yf = x.float(), y.float()
            diff = (xf - yf).abs()
            rms = float(xf.pow(2).mean().sqrt())
            print(f'{k:30s} rms {rms:.4e} max|diff| {float(diff.max()):.4e} count(|diff| > 5% rms) {int((diff > 0.05 * rms).sum())} of {diff.numel()}'
                  f'  nonfinite {int((~torch.isfinite(xf)).sum())} {int((~torch.isfinite(yf)).sum())}')
            continue
        d = float((x - y).norm() / (x.norm() + 1e-30))
        print(f'{k:70s} rel diff {d:9.3e}   norms {float(x.norm()):10.4e} {float(y.norm()):10.4e}')


if __name__ == '__main__':
    main()
