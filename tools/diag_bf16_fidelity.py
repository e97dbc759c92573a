"""bf16 gradient fidelity of the fused unet_256 step at B = 32 against the exact-f32 engine (which matches the reference's
numbers: tests/test_gpu_fullsize.py::test_b32_train_step_against_the_reference[float32]).  Every variant runs in its own
child process (the tuning knobs are read once per process); prints per gradient tensor the norm ratio and the cosine.

    python tools/diag_bf16_fidelity.py "ADN_IGEMM_RING=0" "ADN_IGEMM_RING=1" "ADN_IGEMM_RING_EPI=1" ...
"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(path, dtype):
    from types import SimpleNamespace

    import torch
    from audio_depth_estimation_amd.engine import FusedTrainer
    from audio_depth_estimation_amd.models.unetbaseline_model import define_G
    torch.manual_seed(0)
    m = define_G(SimpleNamespace(dataset=SimpleNamespace(depth_norm=False, max_depth=30.0)), 2, 1, 64, 'unet_256')
    m.compute_dtype = torch.float32 if dtype == 'f32' else torch.bfloat16
    m = m.to('cuda')
    with torch.no_grad():
        m.model.model[3].bias.fill_(1.0)
    m.train()
    eng = m.engine()
    B = int(os.environ.get('DIAG_B', '32'))
    g = torch.Generator().manual_seed(1234)
    audio = torch.rand(B, 2, 256, 256, generator=g).to('cuda')
    gt = 30 * torch.rand(B, 1, 256, 256, generator=g)
    gt[gt < 3] = 0
    tr = FusedTrainer(eng, 'Combined', 0.237, 0.637, 0.869, max_depth=30.0, optimizer='AdamW', lr=0.002, clip_norm=1.0)
    steps = int(os.environ.get('DIAG_STEPS', '1'))
    for _ in range(steps):
        loss, pred = tr.step(audio, gt.to('cuda'))
    out = {'loss': float(loss)}
    for k, prm in m.named_parameters():
        out[k] = eng.grad_view(prm).detach().float().cpu().clone()
    torch.save(out, path)


def main():
    if len(sys.argv) > 3 and sys.argv[1] == 'child':
        return child(sys.argv[2], sys.argv[3])
    import torch
    variants = sys.argv[1:] or ['ADN_IGEMM_RING=0', 'ADN_IGEMM_RING=1']
    runs = []
    for i, v in enumerate(['f32'] + variants):
        path = f'/tmp/diag_fid_{i}.pt'
        env = dict(os.environ)
        if v != 'f32':
            env.update(dict(kv.split('=') for kv in v.split(',') if kv))
        subprocess.run([sys.executable, __file__, 'child', path, 'f32' if v == 'f32' else 'bf16'], check=True, env=env,
                       stdout=subprocess.DEVNULL)
        runs.append(torch.load(path))
    ref = runs[0]
    print('loss f32', ref['loss'], ' bf16:', [r['loss'] for r in runs[1:]])
    print(f'{"tensor":66s}' + ''.join(f' | {v[-24:]:>24s}' for v in variants))
    for k in ref:
        if k == 'loss':
            continue
        a = ref[k].double().reshape(-1)
        cells = []
        for r in runs[1:]:
            b = r[k].double().reshape(-1)
            cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-300))
            cells.append(f' | ratio {float(b.norm() / (a.norm() + 1e-300)):7.3f} cos {cos:6.3f}')
        print(f'{k[-66:]:66s}' + ''.join(cells))


if __name__ == '__main__':
    main()
