"""CPU (gloo, world_size 2) coverage of the data-parallel path in audio-depth-estimation_amd/ddp.py.

1. bucket/watermark logic of GradientAllReducer: buckets are cut from the END of the flat gradient buffer,
   launched as the watermark passes them, every element is summed over ranks exactly once;
2. semantics: per-replica BatchNorm + all-reduced loss statistics + SUM-reduced gradients reproduce the
   reference's DataParallel step (one loss over the gathered global batch, train.py:642-669), checked with the
   CPU oracle: two ranks with one shard each == one process that runs both shards and one global loss.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


class _FakeEngine:
    def __init__(self, n):
        self.flat_g = torch.zeros(n)
        self.on_grad_ready = None


def _shard(rank, B=2, S=128):
    g = torch.Generator().manual_seed(100 + rank)
    audio = torch.rand(B, 2, S, S, generator=g)
    gt = 30 * torch.rand(B, 1, S, S, generator=g)
    gt[gt < (3 + 6 * rank)] = 0          # different valid-pixel counts per shard
    return audio, gt


def _model_sd():
    from types import SimpleNamespace
    from audio_depth_estimation_amd.models.unetbaseline_model import define_G
    torch.manual_seed(0)
    m = define_G(SimpleNamespace(dataset=SimpleNamespace(depth_norm=False)), 2, 1, 4, 'unet_128')
    with torch.no_grad():
        m.model.model[3].bias.fill_(1.0)
    return {k: v.detach().clone() for k, v in m.state_dict().items()}


HYPER = ('Combined', 0.237, 0.637, 0.869)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from audio_depth_estimation_amd.ddp import GradientAllReducer
        from oracle import loss_oracle, unet_oracle
        torch.set_num_threads(2)
        # ---- 1. bucket logic
        n = 1000
        eng = _FakeEngine(n)
        red = GradientAllReducer(bucket_bytes=4 * 300)          # 300-element buckets -> 4 buckets
        red.attach(eng)
        assert red.buckets == [(700, 1000), (400, 700), (100, 400), (0, 100)]
        eng.flat_g.copy_(torch.arange(n, dtype=torch.float32) * (rank + 1))
        red.begin_backward()
        eng.on_grad_ready(750)
        assert red._next == 0                                    # bucket [700,1000) not complete yet
        eng.on_grad_ready(650)
        assert red._next == 1
        eng.on_grad_ready(100)
        assert red._next == 3
        red.finish()
        assert red._next == 4
        expect = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
        assert torch.equal(eng.flat_g, expect)
        stats = torch.tensor([1.0, 2.0, 3.0, 4.0], dtype=torch.float64) * (rank + 1)
        red.all_reduce_loss_stats(stats)
        assert torch.equal(stats, torch.tensor([3.0, 6.0, 9.0, 12.0], dtype=torch.float64))

        # ---- 1b. suffix view (AdaBins: the frozen teacher's prefix of the flat buffer is never exchanged): the
        #          engine reports ABSOLUTE watermarks, the trainer translates them (adabins_engine._GradView)
        from audio_depth_estimation_amd.adabins_engine import _GradView
        n, off = 1000, 400
        full = torch.arange(n, dtype=torch.float32) * (rank + 1)
        view = _GradView(full[off:])
        red2 = GradientAllReducer(bucket_bytes=4 * 250)          # 600 trainable elements -> 250 + 250 + 100
        red2.attach(view)
        assert red2.buckets == [(350, 600), (100, 350), (0, 100)]
        hook = lambda lo: view.on_grad_ready(max(0, lo - off))
        red2.begin_backward()
        hook(800)
        assert red2._next == 0
        hook(740)                                                # absolute 740 = relative 340 <= 350
        assert red2._next == 1
        hook(120)                                                # below the prefix boundary: everything trainable is final
        assert red2._next == 3
        red2.finish()
        tot = sum(r + 1 for r in range(world))
        assert torch.equal(full[off:], torch.arange(off, n, dtype=torch.float32) * tot)
        assert torch.equal(full[:off], torch.arange(off, dtype=torch.float32) * (rank + 1))      # untouched prefix

        # ---- 1c. bf16 payload: the exchange runs on a bf16 copy of every bucket, the f32 buffer receives the sum back
        eng3 = _FakeEngine(1000)
        red3 = GradientAllReducer(bucket_bytes=4 * 300, payload='bf16')
        red3.attach(eng3)
        g3 = torch.Generator().manual_seed(5 + rank)
        eng3.flat_g.copy_(torch.randn(1000, generator=g3))
        mine = eng3.flat_g.clone()
        others = [torch.randn(1000, generator=torch.Generator().manual_seed(5 + r)) for r in range(world)]
        red3.begin_backward()
        eng3.on_grad_ready(0)
        red3.finish()
        assert torch.equal(mine, others[rank])
        want = sum(o.to(torch.bfloat16).float() for o in others)         # bf16 operands, summed (gloo sums in bf16 too)
        assert float((eng3.flat_g - want).abs().max()) <= 2 ** -7 * float(want.abs().max())
        assert eng3.flat_g.dtype == torch.float32 and red3.g16.dtype == torch.bfloat16

        # ---- 2. data-parallel step with the oracle
        sd = _model_sd()
        pkeys = unet_oracle.param_keys(7)
        for k in pkeys:
            sd[k].requires_grad_(True)
        audio, gt = _shard(rank)
        pred, _ = unet_oracle.unet_forward(sd, audio, 7, False, training=True)     # per-replica BatchNorm
        local = torch.stack(loss_oracle.loss_stats(pred, gt)).double()
        total = local.detach().clone()
        red.all_reduce_loss_stats(total)
        glob = total + (local - local.detach())        # value = global stats, gradient = local contribution
        loss = loss_oracle.loss_from_stats(*glob, *HYPER)
        loss.backward()
        sizes = [sd[k].numel() for k in pkeys]
        eng2 = _FakeEngine(sum(sizes))
        eng2.flat_g = torch.cat([sd[k].grad.reshape(-1) for k in pkeys]).float()
        red2 = GradientAllReducer(bucket_bytes=4 * 4096)
        red2.attach(eng2)
        red2.begin_backward()
        off = sum(sizes)
        for s in reversed(sizes):                       # gradients become final from the end of the buffer
            off -= s
            eng2.on_grad_ready(off)
        red2.finish()
        if rank == 0:
            out.put((loss.item(), eng2.flat_g.numpy().tobytes()))      # plain bytes: no shared-memory handles
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_matches_global_batch_step():
    from oracle import loss_oracle, unet_oracle
    world = 2
    port = _free_port()
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    loss_dp, raw = out.get(timeout=240)
    import numpy as np
    grads_dp = torch.from_numpy(np.frombuffer(raw, dtype=np.float32).copy())
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single process: both shards (each with its own BatchNorm statistics), ONE loss over all valid pixels
    sd = _model_sd()
    pkeys = unet_oracle.param_keys(7)
    for k in pkeys:
        sd[k].requires_grad_(True)
    stats = None
    for r in range(world):
        audio, gt = _shard(r)
        pred, _ = unet_oracle.unet_forward(sd, audio, 7, False, training=True)
        st = torch.stack(loss_oracle.loss_stats(pred, gt)).double()
        stats = st if stats is None else stats + st
    loss = loss_oracle.loss_from_stats(*stats, *HYPER)
    loss.backward()
    grads = torch.cat([sd[k].grad.reshape(-1) for k in pkeys]).float()
    assert abs(loss.item() - loss_dp) <= 1e-6 * abs(loss.item())
    assert float((grads - grads_dp).abs().max()) <= 1e-5 * float(grads.abs().max())
