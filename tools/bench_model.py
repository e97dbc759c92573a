"""Secondary-workload timing (NOT the headline bench): one fused train step of a DoubleConv-family model.

    python tools/bench_model.py --model rgb --batch 32 --size 256 --steps 10

Prints a per-kernel-family table (launch-plan replay with a HIP event pair around EVERY launch, grouped by entry
point) and the hipGraph step time.  GEMM families also show algorithmic TFLOP/s.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def adabins_main(args, dev, dtype, gt, g):
    from audio_depth_estimation_amd import _lib
    from audio_depth_estimation_amd.adabins_engine import AdaBinsTrainer
    from audio_depth_estimation_amd.models.adabins_distillation_model import AdaBinsDistillationModel
    B, S = args.batch, args.size
    audio, rgb = torch.rand(B, 2, S, S, generator=g).to(dev), torch.rand(B, 3, S, S, generator=g).to(dev)
    gt[gt < 3] = 0
    if args.model == 'baseres':
        from audio_depth_estimation_amd.base_residual_engine import BaseResidualTrainer
        from audio_depth_estimation_amd.models.base_residual_model import BaseResidualDepthNet
        model = BaseResidualDepthNet(2, 64, True, S, 30.0)
        model.compute_dtype = dtype
        model = model.to(dev).train()
        inner = BaseResidualTrainer(model.engine(), use_silog=True, lr=1e-4)

        class _T:
            def step(self, a, r, t):
                return inner.step(a, t)
        tr = _T()
    else:
        model = AdaBinsDistillationModel(128, 64, S, 30.0)
        model.compute_dtype = dtype
        model = model.to(dev).train()
        tr = AdaBinsTrainer(model.engine(), lr=1e-4)
    for _ in range(2):
        tr.step(audio, rgb, gt)
    torch.cuda.synchronize()
    _lib.RECORD = []
    tr.step(audio, rgb, gt)
    plan, _lib.RECORD = _lib.RECORD, None
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in plan]
    fam = {}
    for i, (fn, a, name, meta) in enumerate(plan):
        evs[i][0].record()
        a() if fn is None else fn(*a)
        evs[i][1].record()
    torch.cuda.synchronize()
    for i, (fn, a, name, meta) in enumerate(plan):
        acc = fam.setdefault(name, [0.0, 0.0, 0])
        acc[0] += evs[i][0].elapsed_time(evs[i][1])
        acc[1] += meta.get('flops', 0.0)
        acc[2] += 1
    print(f'{"entry point":34s} {"ms/step":>9s} {"launches":>9s} {"TFLOP/s":>9s}')
    for name, (ms, fl, n) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
        tf = f'{fl / (ms * 1e-3) / 1e12:9.1f}' if fl else '         '
        print(f'{name:34s} {ms:9.3f} {n:9d} {tf}')
    (inner if args.model == 'baseres' else tr).enable_graph(after_steps=0)
    tr.step(audio, rgb, gt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = tr.step(audio, rgb, gt)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(json.dumps({'model': args.model, 'batch': B, 'size': S, 'dtype': args.dtype, 'ms_per_step': 1e3 * el / args.steps,
                      'depth_maps_per_s': B * args.steps / el, 'gemm_gflop_per_step': sum(v[1] for v in fam.values()) / 1e9,
                      'loss': float(loss), 'mem_gb': torch.cuda.max_memory_allocated() / 2 ** 30}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='rgb', choices=['rgb', 'binaural', 'adabins', 'baseres'])
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--base', type=int, default=64)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32', 'mxfp8'],
                    help='mxfp8: 3x3 conv forward / input-gradient GEMMs in block-scaled fp8 (DoubleConv nets only)')
    ap.add_argument('--detail', action='store_true', help='list every GEMM launch')
    args = ap.parse_args()
    from audio_depth_estimation_amd.engine import FusedTrainer
    dev = torch.device('cuda', 0)
    dtype = {'bf16': torch.bfloat16, 'f32': torch.float32, 'mxfp8': torch.float8_e4m3fn}[args.dtype]
    torch.manual_seed(0)
    B, S = args.batch, args.size
    g = torch.Generator().manual_seed(1)
    gt = (30 * torch.rand(B, 1, S, S, generator=g)).to(dev)
    if args.model == 'rgb':
        from audio_depth_estimation_amd.models.rgb_depth_model import RGBDepthNet
        model = RGBDepthNet(args.base, True, S, 30.0)
        x = torch.rand(B, 3, S, S, generator=g).to(dev)
        trainer_args = dict(criterion='DepthLoss', l1_weight=1.0, silog_weight=0.1, optimizer='AdamW', lr=1e-4,
                            weight_decay=0.01, clip_norm=None)
    elif args.model == 'binaural':
        from audio_depth_estimation_amd.models.binaural_attention_model import BinauralAttentionDepthNet
        model = BinauralAttentionDepthNet(args.base, True, S, 30.0)
        x = torch.rand(B, 2, S, S, generator=g).to(dev)
        gt[gt < 3] = 0
        trainer_args = dict(criterion='L1', optimizer='AdamW', lr=1e-3, weight_decay=0.01, clip_norm=None,
                            mask_mode='gt0')
    if args.model in ('adabins', 'baseres'):
        return adabins_main(args, dev, dtype, gt, g)
    model.compute_dtype = dtype
    model = model.to(dev).train()
    tr = FusedTrainer(model.engine(), **trainer_args)
    for _ in range(3):
        tr.step(x, gt)
    torch.cuda.synchronize()
    tr.enable_launch_plan(after_steps=0)
    tr.step(x, gt)
    torch.cuda.synchronize()
    plan = tr._plan
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in plan]
    fam, rows = {}, []
    reps = 3
    for it in range(reps):
        for i, (fn, a, name, meta) in enumerate(plan):
            evs[i][0].record()
            if fn is None:
                a()
            else:
                fn(*a)
            evs[i][1].record()
        torch.cuda.synchronize()
        for i, (fn, a, name, meta) in enumerate(plan):
            ms = evs[i][0].elapsed_time(evs[i][1])
            acc = fam.setdefault(name, [0.0, 0.0, 0])
            acc[0] += ms
            acc[1] += meta.get('flops', 0.0)
            acc[2] += 1
            if it == reps - 1 and 'flops' in meta:
                tag = ''
                try:
                    d = a[0]._obj
                    if name == 'adn_igemm':
                        tag = f'g{d.geom} H{d.Hs} C{d.C0}+{d.C1} N{d.N} epi{d.epi}'
                    elif name == 'adn_wgrad':
                        tag = f'H{d.Hs} R{d.R0}+{d.R1} C{d.C0}+{d.C1}'
                except Exception:
                    pass
                rows.append((name + ' ' + tag, ms, meta['flops']))
    total = sum(v[0] for v in fam.values()) / reps
    print(f'{"entry point":34s} {"ms/step":>9s} {"launches":>9s} {"TFLOP/s":>9s}')
    for name, (ms, fl, n) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
        tf = f'{fl / (ms * 1e-3) / 1e12:9.1f}' if fl else '         '
        print(f'{name:34s} {ms / reps:9.3f} {n // reps:9d} {tf}')
    print(f'{"sum of kernel events":34s} {total:9.3f}')
    if args.detail:
        for name, ms, fl in rows:
            print(f'   {name:44s} {ms:8.3f} ms  {fl / 1e9:9.1f} GFLOP  {fl / (ms * 1e-3) / 1e12:7.1f} TF/s')
    tr._plan, tr._plan_after = None, None
    tr.enable_graph(after_steps=0)
    tr.step(x, gt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = tr.step(x, gt)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    gemm_fl = sum(v[1] for v in fam.values()) / reps
    print(json.dumps({'model': args.model, 'batch': B, 'size': S, 'dtype': args.dtype, 'ms_per_step': 1e3 * el / args.steps,
                      'depth_maps_per_s': B * args.steps / el, 'gemm_gflop_per_step': gemm_fl / 1e9,
                      'step_tflops': gemm_fl / (el / args.steps) / 1e12, 'loss': float(loss),
                      'mem_gb': torch.cuda.max_memory_allocated() / 2 ** 30}))


if __name__ == '__main__':
    main()
