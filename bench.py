"""Headline benchmark: depth-maps/s of a full U-Net train step on synthetic BatVisionV2-shaped batches.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1]: models.unetbaseline_model.define_G(unet_256, ngf 64) on 256x256
two-channel inputs, bf16 MFMA compute, batch 32 PER GPU (weak scaling), one step = forward + masked
Combined loss (conf/mode/train.yaml weights) + backward + [gradient all-reduce] + clip_grad_norm(1.0) +
AdamW -- exactly /root/reference/train.py:633-691.  Inputs are resident in HBM when the timed region starts.
Rank 0 prints ONE JSON line; it also carries
  roofline:     the dominant kernel family (MFMA implicit GEMM) timed with HIP events on its own stream
                inside the timed region, algorithmic FLOPs / time vs the bf16 dense MFMA peak;
  cpu_baseline: the CPU oracle (port of the reference's torch-CPU path) timed on this box's host cores
                on a bounded sample (rank 0, N=1 only).
"""
import argparse
import contextlib
import json
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_PEAK_TF = {'bf16': 2500.0, 'f32': 157.3}      # MI355X_MICROARCH.md, dense
L1_W, SILOG_W, SILOG_LAMBDA, LR = 0.237, 0.637, 0.869, 0.002   # conf/mode/train.yaml


def synth_batch(B, S, seed, device):
    """audio ~ U[0,1) (mel min-max range), depth 30*U with <3 m invalid."""
    g = torch.Generator().manual_seed(seed)
    audio = torch.rand(B, 2, S, S, generator=g)
    gt = 30.0 * torch.rand(B, 1, S, S, generator=g)
    gt[gt < 3.0] = 0.0
    return audio.to(device), gt.to(device)


def effective_cores():
    """CPU share of this process: min(affinity mask, cgroup quota) -- os.cpu_count() reports the whole host."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


def cpu_baseline(B, steps, warmup):
    """Reference CPU path restated by oracle/ (fwd + loss + bwd through torch autograd, then the same
    clip_grad_norm_ / AdamW the reference calls), all host cores, fp32."""
    from oracle import loss_oracle, unet_oracle
    from audio_depth_estimation_amd.models.unetbaseline_model import define_G
    cores = effective_cores()
    torch.set_num_threads(cores)
    cfg = SimpleNamespace(dataset=SimpleNamespace(depth_norm=False, max_depth=30.0))
    torch.manual_seed(0)
    with contextlib.redirect_stdout(sys.stderr):      # define_G prints its init banner like the reference: keep stdout = the JSON line
        model = define_G(cfg, 2, 1, 64, 'unet_256')
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    pkeys = unet_oracle.param_keys(8)
    params = [sd[k].requires_grad_(True) for k in pkeys]
    opt = torch.optim.AdamW(params, lr=LR)
    times = []
    for it in range(warmup + steps):
        audio, gt = synth_batch(B, 256, 1234 + it, 'cpu')
        t0 = time.perf_counter()
        opt.zero_grad()
        pred, stats = unet_oracle.unet_forward(sd, audio, 8, False, training=True)
        loss = loss_oracle.masked_loss(pred, gt, 'Combined', L1_W, SILOG_W, SILOG_LAMBDA)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        for k, v in stats.items():
            sd[k] = v
        dt = time.perf_counter() - t0
        if it >= warmup:
            times.append(dt)
    times.sort()
    med = times[len(times) // 2]
    return {'value': B / med, 'unit': 'depth-maps/s', 'cores': cores, 'kind': 'port',
            'sample': f'unet_256 ngf64 fp32 B={B} 256x256, median of {steps} train steps after {warmup} warm-up, '
                      f'torch {torch.__version__} CPU, {cores} threads'}


def build_trainer(dtype, device, reducer):
    from audio_depth_estimation_amd.engine import FusedTrainer
    from audio_depth_estimation_amd.models.unetbaseline_model import define_G
    cfg = SimpleNamespace(dataset=SimpleNamespace(depth_norm=False, max_depth=30.0))
    torch.manual_seed(0)
    with contextlib.redirect_stdout(sys.stderr):      # define_G prints its init banner like the reference: keep stdout = the JSON line
        model = define_G(cfg, 2, 1, 64, 'unet_256')
    model.compute_dtype = dtype
    model = model.to(device).train()
    trainer = FusedTrainer(model.engine(), 'Combined', L1_W, SILOG_W, SILOG_LAMBDA, max_depth=30.0,
                           optimizer='AdamW', lr=LR, clip_norm=1.0, ddp=reducer)
    return model, trainer


def gemm_event_pass(trainer, batches, prof_steps):
    """The recorded launch plan replayed with a HIP event pair around every GEMM launch, on the stream the kernels run
    on.  The host replays prebuilt calls, so the GPU queue stays full and the event intervals are kernel durations.
    Returns {label: [algorithmic flops, seconds, launches]}."""
    plan = trainer._plan
    timed = [(i, e[3]) for i, e in enumerate(plan) if e[3].get('label') in ('igemm', 'wgrad')]
    evs = {i: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for i, _ in timed}
    fam = {}
    for it in range(prof_steps):
        trainer._g_audio.copy_(batches[it % len(batches)][0])
        trainer._g_gt.copy_(batches[it % len(batches)][1])
        for i, (fn, a, name, meta) in enumerate(plan):
            if i in evs:
                evs[i][0].record()
            if fn is None:
                a()
            else:
                fn(*a)
            if i in evs:
                evs[i][1].record()
        torch.cuda.synchronize()
        for i, meta in timed:
            acc = fam.setdefault(meta['label'], [0.0, 0.0, 0])
            acc[0] += meta['flops']
            acc[1] += evs[i][0].elapsed_time(evs[i][1]) * 1e-3
            acc[2] += 1
    return fam


def per_step_times(trainer, batches, n_steps):
    """Sustained run: ``n_steps`` back-to-back steps, one HIP event between consecutive steps; returns the per-step
    durations (ms) and the wall time.  The events sit on the stream the step runs on.  The step count is FIXED by the
    caller (from the all-reduced time of the timed region): under N > 1 every rank must issue the same number of
    collectives, a per-rank "until t seconds have passed" loop could leave one rank a step ahead and hang the job."""
    evs = [torch.cuda.Event(enable_timing=True)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs[0].record()
    for n in range(1, n_steps + 1):
        trainer.step(*batches[(n - 1) % len(batches)])
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        evs.append(e)
        if n % 64 == 0:
            evs[-1].synchronize()          # keep the host at most 64 steps ahead: wall time then tracks the device
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return [evs[k].elapsed_time(evs[k + 1]) for k in range(n_steps)], wall


def f32_exact_record(device, B, S, rank):
    """The same train step on the exact-f32 MFMA path (the path that carries the 1e-4 tolerance of north_star)."""
    model, trainer = build_trainer(torch.float32, device, None)
    batches = [synth_batch(B, S, 1234 + 1000 * rank + i, device) for i in range(2)]
    for i in range(3):
        trainer.step(*batches[i % 2])
    torch.cuda.synchronize()
    trainer.enable_launch_plan(after_steps=0)
    trainer.step(*batches[0])
    torch.cuda.synchronize()
    fam = gemm_event_pass(trainer, batches, 2)
    trainer._plan, trainer._plan_after = None, None
    trainer.enable_graph(after_steps=0)
    trainer.step(*batches[0])
    times, wall = per_step_times(trainer, batches, 64)
    times.sort()
    med = times[len(times) // 2]
    flops = sum(v[0] for v in fam.values())
    secs = sum(v[1] for v in fam.values())
    del trainer, model
    torch.cuda.empty_cache()
    return {'dtype': 'f32', 'value': B / (med * 1e-3), 'unit': 'depth-maps/s', 'median_ms_per_step': med,
            'steps': len(times), 'gemm_tflops': flops / secs / 1e12, 'peak': MFMA_PEAK_TF['f32'],
            'frac': flops / secs / 1e12 / MFMA_PEAK_TF['f32'],
            'note': 'exact-f32 MFMA path (v_mfma_f32_16x16x4_f32): predictions within 1e-4 relative L1 of the CPU reference'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=600)      # ~2 s of timed region at ~3.3 ms per step
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=32, help='per-GPU batch')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-graph', action='store_true', help='launch every kernel eagerly (default for --gpus > 1)')
    ap.add_argument('--no-f32', action='store_true', help='skip the exact-f32 sub-record (N=1 only)')
    ap.add_argument('--sustain-seconds', type=float, default=2.0,
                    help='length of the sustained per-step-event run behind the timed region (0 = skip)')
    ap.add_argument('--ddp-payload', default='f32', choices=['f32', 'bf16'],
                    help='what the gradient all-reduce carries (N > 1): f32 = DataParallel arithmetic (default), bf16 = half the bytes')
    ap.add_argument('--bucket-mb', type=int, default=32, help='all-reduce bucket size (N > 1)')
    ap.add_argument('--cpu-batch', type=int, default=8)
    ap.add_argument('--cpu-steps', type=int, default=25)     # B=8: about 12 s of host work (bounded sample)
    args = ap.parse_args()

    from audio_depth_estimation_amd import ddp as addp

    rank, world, local = addp.init_from_env('nccl')
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node N for --gpus N > 1')
    device = torch.device('cuda', local % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(device)
    dtype = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    reducer = addp.GradientAllReducer(bucket_bytes=args.bucket_mb << 20, payload=args.ddp_payload) if world > 1 else None
    backend = dist.get_backend() if world > 1 else None
    model, trainer = build_trainer(dtype, device, reducer)
    B, S = args.batch, 256
    batches = [synth_batch(B, S, 1234 + 1000 * rank + i, device) for i in range(4)]
    if reducer is not None:
        model.engine().bind_parameters()
        reducer.broadcast_parameters(model.engine().flat_p)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    use_graph = world == 1 and not args.no_graph
    for i in range(args.warmup):
        trainer.step(*batches[i % len(batches)])
    barrier()
    # Record the step's launch plan (prebuilt ctypes calls).  N>1 replays it (collectives stay ordinary
    # torch.distributed calls inside it); N=1 captures the same step into one hipGraph further below.
    trainer.enable_launch_plan(after_steps=0)
    trainer.step(*batches[0])
    barrier()
    prof_steps = max(1, min(args.steps, 5))
    fam = gemm_event_pass(trainer, batches, prof_steps)
    barrier()
    if use_graph:
        trainer._plan, trainer._plan_after = None, None
        trainer.enable_graph(after_steps=0)          # N=1: the whole step is one hipGraph
        trainer.step(*batches[0])                    # capture + first replay (untimed)
        barrier()
    # ---- the timed region of the contract: exactly K steps between two barrier + synchronize pairs
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss, _ = trainer.step(*batches[i % len(batches)])
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    final_loss = float(loss.item())
    # ---- sustained run behind it: >= 2 s of back-to-back steps, one event per step, median reported
    sustained = None
    if args.sustain_seconds > 0:
        n_sust = max(50, min(4000, int(args.sustain_seconds / (elapsed / args.steps)) + 1))    # same on every rank
        times, wall = per_step_times(trainer, batches, n_sust)
        barrier()
        st = sorted(times)
        med = st[len(st) // 2]
        sustained = {'steps': len(times), 'seconds': wall, 'value': world * B * len(times) / wall,
                     'median_ms_per_step': med, 'p10_ms': st[len(st) // 10], 'p90_ms': st[(9 * len(st)) // 10],
                     'value_at_median': world * B / (med * 1e-3)}

    if rank == 0:
        dom = max(fam, key=lambda k: fam[k][1])
        flops, secs, launches = fam[dom]
        achieved = flops / secs / 1e12
        peak = MFMA_PEAK_TF[args.dtype]
        gemm_secs = sum(v[1] for v in fam.values())
        # HBM bytes per launch of the dominant kernel: NOT measured in this run -- taken from the committed PMC passes
        # of this command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, read side doubled per
        # MI355X_MICROARCH.md; tools/pmc_traffic.py); null when no such file matches
        traffic, traffic_src, traffic_commit = None, None, None
        for name in ('r03_pmc_traffic_gemm.json', 'r02_pmc_traffic_gemm.json', 'r01_pmc_traffic_gemm.json'):
            try:
                pm = json.load(open(os.path.join(ROOT, 'profiles', name)))
                fams = (('igemm_ring_kernel', 'igemm_patch_kernel', 'igemm_mfma_kernel') if dom == 'igemm'
                        else ('wgrad_k4_patch_kernel', 'wgrad_mfma_kernel'))
                rows = [v for k, v in pm.items() if k.startswith(fams) and isinstance(v, dict)]
                if rows and args.dtype == 'bf16' and B == 32:
                    traffic = sum(r['launches'] * r['hbm_bytes_per_launch'] for r in rows) / sum(r['launches'] for r in rows)
                    traffic_src, traffic_commit = 'profiles/' + name, pm.get('_commit')
                    break
            except Exception:
                continue
        par = f'dp{world}'
        if world > 1:
            par += (f' ({"RCCL" if backend == "nccl" else backend} gradient all-reduce in {args.bucket_mb} MiB buckets beside '
                    f'the backward, {args.ddp_payload} payload, backend={backend})')
        result = {
            'metric': 'depth-maps/sec (train step)', 'value': world * B * args.steps / elapsed,
            'unit': 'depth-maps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': 'unetbaseline_model unet_256 (ngf 64) train step, BatVisionV2-shaped 256x256, '
                                   f'batch {B}/GPU, Combined L1+SIlog loss, clip 1.0, AdamW',
                       'global_batch': world * B, 'image_size': S, 'launch': 'hipGraph' if use_graph else 'launch-plan',
                       'parallelism': par},
            'roofline': {'bound': 'mfma',
                         'kernel': {'igemm': 'igemm_ring_kernel + igemm_patch_kernel + igemm_mfma_kernel (implicit-GEMM family: forward + input gradients)',
                                    'wgrad': 'wgrad_k4_patch_kernel + wgrad_mfma_kernel (weight gradients)'}.get(dom, dom),
                         'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s', 'frac': achieved / peak,
                         'traffic': traffic, 'traffic_source': traffic_src, 'traffic_commit': traffic_commit, 'launches': launches,
                         'avg_launch_ms': 1e3 * secs / launches,
                         'gemm_ms_per_step': 1e3 * gemm_secs / max(1, prof_steps),
                         'all_gemm_tflops': sum(v[0] for v in fam.values()) / gemm_secs / 1e12},
            'sustained': sustained,
            'final_loss': final_loss,
        }
        if world == 1 and not args.no_f32 and args.dtype == 'bf16':
            del trainer, model
            torch.cuda.empty_cache()
            result['f32_exact'] = f32_exact_record(device, B, S, rank)
        if world == 1 and not args.no_cpu_baseline:
            result['cpu_baseline'] = cpu_baseline(args.cpu_batch, args.cpu_steps, 2)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
