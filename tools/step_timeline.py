"""Per-launch timeline of one fused unet_256 train step (tuning aid).

    python tools/step_timeline.py [--batch 32] [--reps 5] [--dtype bf16]
Replays the recorded launch plan with a HIP event pair around EVERY launch (on the stream the kernels run on) and
prints, in launch order, the entry point, a short shape tag and the median duration over `reps` replays, then the
totals per entry point.  Event pairs add ~2 us of gaps, so the sum is larger than the hipGraph step time; the
per-launch durations are what this is for.
"""
import argparse
import collections
import contextlib
import os
import sys
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from audio_depth_estimation_amd import _lib  # noqa: E402
from audio_depth_estimation_amd.engine import FusedTrainer  # noqa: E402
from audio_depth_estimation_amd.models.unetbaseline_model import define_G  # noqa: E402


def tag(name, args):
    try:
        if name in ('adn_igemm', 'adn_wgrad'):
            d = args[0]._obj
            if name == 'adn_igemm':
                return f'g{d.geom} Hs{d.Hs} C{d.C0}+{d.C1} N{d.N} epi{d.epi}'
            return f'Hs{d.Hs} R{d.R0}+{d.R1} C{d.C0}+{d.C1}'
    except Exception:
        pass
    return ''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--dtype', default='bf16')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    cfg = SimpleNamespace(dataset=SimpleNamespace(depth_norm=False, max_depth=30.0))
    torch.manual_seed(0)
    with contextlib.redirect_stdout(sys.stderr):
        model = define_G(cfg, 2, 1, 64, 'unet_256')
    model.compute_dtype = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    model = model.to(dev).train()
    tr = FusedTrainer(model.engine(), 'Combined', 0.237, 0.637, 0.869, max_depth=30.0, optimizer='AdamW', lr=0.002,
                      clip_norm=1.0)
    g = torch.Generator().manual_seed(1234)
    audio = torch.rand(args.batch, 2, 256, 256, generator=g).to(dev)
    gt = 30.0 * torch.rand(args.batch, 1, 256, 256, generator=g)
    gt[gt < 3.0] = 0.0
    gt = gt.to(dev)
    for _ in range(3):
        tr.step(audio, gt)
    tr.enable_launch_plan(after_steps=0)
    tr.step(audio, gt)
    plan = tr._plan
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in plan]
    times = [[] for _ in plan]
    for _ in range(args.reps):
        for i, (fn, a, name, meta) in enumerate(plan):
            evs[i][0].record()
            if fn is None:
                a()
            else:
                fn(*a)
            evs[i][1].record()
        torch.cuda.synchronize()
        for i in range(len(plan)):
            times[i].append(evs[i][0].elapsed_time(evs[i][1]) * 1e3)
    tot = collections.OrderedDict()
    total = 0.0
    for i, (fn, a, name, meta) in enumerate(plan):
        t = sorted(times[i])[len(times[i]) // 2]
        total += t
        k = tot.setdefault(name, [0, 0.0])
        k[0] += 1
        k[1] += t
        fl = meta.get('flops')
        extra = f'  {fl / t / 1e6:7.1f} TF/s' if fl else ''
        print(f'{i:4d} {name:28s} {tag(name, a):34s} {t:8.1f} us{extra}')
    print(f'--- {len(plan)} launches, sum {total:.1f} us')
    for name, (n, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print(f'{name:28s} x{n:3d} {t:9.1f} us')


if __name__ == '__main__':
    main()
