"""Two ranks on ONE GPU (gloo collectives on device tensors) through the real fused trainers: the N > 1 path of
bench.py / train*.py with the HIP engines instead of the fake engine of test_ddp_gloo.py.

Checked per model (U-Net baseline, RGBDepthNet): after three fused steps on different shards the replicas hold
bit-identical parameters (same SUM-reduced gradients, same optimizer), and the first step equals what ONE process
computes when it runs both shards through two engines sharing the weights, sums their gradients and applies one
optimizer step with the global-batch loss -- DataParallel semantics with per-replica BatchNorm.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _make(kind):
    from types import SimpleNamespace
    torch.manual_seed(0)
    if kind == 'unet':
        from audio_depth_estimation_amd.models.unetbaseline_model import define_G
        m = define_G(SimpleNamespace(dataset=SimpleNamespace(depth_norm=False)), 2, 1, 64, 'unet_128')
        with torch.no_grad():
            m.model.model[3].bias.fill_(1.0)
        targs = dict(criterion='Combined', l1_weight=0.237, silog_weight=0.637, silog_lambda=0.869, clip_norm=1.0, lr=2e-3)
        cin = 2
    else:
        from audio_depth_estimation_amd.models.rgb_depth_model import RGBDepthNet
        m = RGBDepthNet(32, True, 128, 30.0)
        targs = dict(criterion='DepthLoss', l1_weight=1.0, silog_weight=0.1, clip_norm=None, lr=1e-3, weight_decay=0.01)
        cin = 3
    m.compute_dtype = torch.float32
    return m.to('cuda').train(), targs, cin


def _shard(rank, cin, B=2, S=128):
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.rand(B, cin, S, S, generator=g)
    gt = 30 * torch.rand(B, 1, S, S, generator=g)
    gt[gt < (3 + 6 * rank)] = 0
    return x.to('cuda'), gt.to('cuda')


def _worker(rank, world, port, kind, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from audio_depth_estimation_amd.ddp import GradientAllReducer
        from audio_depth_estimation_amd.engine import FusedTrainer
        model, targs, cin = _make(kind)
        red = GradientAllReducer(bucket_bytes=1 << 20)
        tr = FusedTrainer(model.engine(), ddp=red, **targs)
        model.engine().bind_parameters()
        red.broadcast_parameters(model.engine().flat_p)
        x, gt = _shard(rank, cin)
        loss, _ = tr.step(x, gt)
        loss = float(loss)                                # (the returned 0-dim tensor is a view of the trainer's buffer)
        first = model.engine().flat_p.detach().cpu().clone()
        for _ in range(2):
            tr.step(x, gt)
        torch.cuda.synchronize()
        out.put((rank, loss, first.numpy().tobytes(), model.engine().flat_p.detach().cpu().numpy().tobytes()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('kind', ['unet', 'rgb'])
def test_two_ranks_one_gpu(kind):
    import numpy as np
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, kind, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, l0, f0, p0), (_, l1, f1, p1) = res
    assert l0 == l1                                   # one global-batch loss on both ranks
    assert p0 == p1                                   # replicas stay bit-identical
    # single-process emulation of the first step: two engines (one per shard) sharing the initial weights
    from audio_depth_estimation_amd import kernels as K
    from audio_depth_estimation_amd.engine import FusedTrainer
    ma, targs, cin = _make(kind)
    mb, _, _ = _make(kind)
    ea, eb = ma.engine(), mb.engine()
    ea.bind_parameters()
    eb.bind_parameters()
    shards = [_shard(0, cin), _shard(1, cin)]
    preds = [ea.forward(shards[0][0], True).clone(), eb.forward(shards[1][0], True).clone()]
    tr = FusedTrainer(ea, **targs)
    tr._setup(preds[0].device)
    stats = torch.zeros(4, dtype=torch.float64, device='cuda')
    tot = torch.zeros(4, dtype=torch.float64, device='cuda')
    crit = tr.criterion
    for (x, gt), pr in zip(shards, preds):
        if crit == 3:
            K.l1tv_stats(pr, gt, stats, tr.loss_ws)
        else:
            K.loss_stats(pr, gt, 1.0, tr.mask_mode, 1e-6, stats, tr.loss_ws)
        tot += stats
    gouts = []
    for (x, gt), pr in zip(shards, preds):
        go = torch.empty_like(pr)
        if crit == 3:
            K.l1tv_finish(pr, gt, tot, 2, tr.l1_weight, tr.silog_weight, tr.loss, go)
        else:
            K.loss_finish(pr, gt, 1.0, tr.mask_mode, 1e-6, tot, crit, tr.l1_weight, tr.silog_weight, tr.silog_lambda, tr.loss, go)
        gouts.append(go)
    ea.backward(gouts[0])
    eb.backward(gouts[1])
    ea.flat_g += eb.flat_g
    if tr.clip_norm is not None:
        K.grad_norm(ea.flat_g, float(tr.clip_norm), tr.state, tr.norm_ws)
    K.optimizer_step(ea.flat_p, ea.flat_g, tr.exp_avg, tr.exp_avg_sq, tr.opt_kind, tr.lr, tr.betas[0], tr.betas[1], tr.eps,
                     tr.weight_decay, tr.clip_norm is not None, tr.state, bf16_copy=ea.flat_w16)
    want = ea.flat_p.detach().cpu().numpy()
    got = np.frombuffer(f0, dtype=np.float32)
    assert abs(float(tr.loss) - l0) <= 1e-6 * abs(l0)
    # BN running statistics are per replica (rank 0 = shard 0): parameters (incl. them) must agree with engine a
    assert float(np.abs(got - want).max()) <= 1e-6 + 0.02 * tr.lr


# ---- AdaBins distillation / Base+Residual trainers (the reference wraps both in nn.DataParallel when given gpu_ids:
# adabins_distillation_model.py:493-496, base_residual_model.py:266-269) ---------------------------------------------
def _make2(kind, ddp=None):
    torch.manual_seed(0)
    if kind == 'adabins':
        from audio_depth_estimation_amd.adabins_engine import AdaBinsTrainer
        from audio_depth_estimation_amd.models.adabins_distillation_model import AdaBinsDistillationModel
        m = AdaBinsDistillationModel(16, 64, 32, 30.0)
        m.compute_dtype = torch.float32
        m = m.to('cuda').train()
        tr = AdaBinsTrainer(m.engine(), lr=1e-3, ddp=ddp)
    else:
        from audio_depth_estimation_amd.base_residual_engine import BaseResidualTrainer
        from audio_depth_estimation_amd.models.base_residual_model import BaseResidualDepthNet
        m = BaseResidualDepthNet(2, 64, True, 32, 30.0)
        m.compute_dtype = torch.float32
        m = m.to('cuda').train()
        tr = BaseResidualTrainer(m.engine(), use_silog=True, lr=1e-3, ddp=ddp)
    m.engine().bind_parameters()
    return m, tr


def _shard2(seed, B=2, S=32):
    g = torch.Generator().manual_seed(200 + seed)
    audio, rgb = torch.rand(B, 2, S, S, generator=g), torch.rand(B, 3, S, S, generator=g)
    gt = 30 * torch.rand(B, 1, S, S, generator=g)
    gt[gt < (3 + 6 * seed)] = 0
    return audio.to('cuda'), rgb.to('cuda'), gt.to('cuda')


def _step2(kind, tr, shard):
    audio, rgb, gt = shard
    loss, terms = tr.step(audio, rgb, gt) if kind == 'adabins' else tr.step(audio, gt)
    return float(loss), terms.detach().cpu().clone()


def _worker2(rank, world, port, kind, same, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from audio_depth_estimation_amd.ddp import GradientAllReducer
        red = GradientAllReducer(bucket_bytes=1 << 20)
        model, tr = _make2(kind, red)
        red.broadcast_parameters(model.engine().flat_p)
        shard = _shard2(0 if same else rank)
        loss, terms = _step2(kind, tr, shard)
        first = model.engine().flat_p.detach().cpu().clone()
        for _ in range(2):
            _step2(kind, tr, shard)
        torch.cuda.synchronize()
        out.put((rank, loss, terms.numpy().tobytes(), first.numpy().tobytes(),
                 model.engine().flat_p.detach().cpu().numpy().tobytes()))
    finally:
        dist.destroy_process_group()


def _spawn2(kind, same):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker2, args=(r, 2, port, kind, same, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize('kind', ['adabins', 'baseres'])
def test_two_ranks_distillation_trainers(kind):
    import numpy as np
    # (1) different shards: one global loss, replicas stay bit-identical over three steps
    (_, l0, t0, _, p0), (_, l1, t1, _, p1) = _spawn2(kind, same=False)
    assert l0 == l1 and t0 == t1
    assert p0 == p1
    # (2) both ranks on the SAME shard: the global-batch loss and the SUM-reduced gradient equal the single-process
    #     ones (every normaliser doubles), so the first step must match a plain trainer step on that shard
    (_, l0, t0, f0, _), _ = _spawn2(kind, same=True)
    model, tr = _make2(kind)
    loss, terms = _step2(kind, tr, _shard2(0))
    want = model.engine().flat_p.detach().cpu().numpy()
    got = np.frombuffer(f0, dtype=np.float32)
    assert abs(loss - l0) <= 1e-5 * abs(loss)
    tg, tw = np.frombuffer(t0, dtype=np.float32), terms.numpy()
    nt = 7 if kind == 'adabins' else 4                # (AdaBins terms[7] is the valid-pixel count: global = 2x)
    np.testing.assert_allclose(tg[:nt], tw[:nt], rtol=1e-5, atol=1e-7)
    if kind == 'adabins':
        assert tg[7] == 2 * tw[7]
    assert float(np.abs(got - want).max()) <= 1e-6 + 0.02 * tr.lr


def test_torchrun_entry_point_two_ranks(tmp_path):
    """The train_adabins_distillation counterpart under torchrun (2 ranks on one GPU, gloo): rank 0 writes the
    checkpoint, each rank takes its half of the 8 synthetic items (2 fused steps of batch 2 per rank)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ADN_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0',
               PYTHONPATH=root + os.pathsep + os.environ.get('PYTHONPATH', ''))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', str(_free_port()), '-m', 'audio_depth_estimation_amd.train_adabins_distillation',
           '--synthetic', '8', '--batch_size', '2', '--nb_epochs', '1', '--experiment_name', 'ddp', '--precision', 'f32']
    r = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    ck = torch.load(os.path.join(tmp_path, 'results', 'ddp', 'best_model.pth'), map_location='cpu')
    assert ck['epoch'] == 1 and float(next(iter(ck['optimizer_state_dict']['state'].values()))['step']) == 2
    assert all(torch.isfinite(v).all() for v in ck['model_state_dict'].values() if v.is_floating_point())


def test_torchrun_resume_keeps_optimizer_state_and_exchange(tmp_path):
    """resuming under the reducer.  Two ranks (gloo) train the binaural script for one epoch, a second
    launch resumes from that checkpoint for a second epoch; the result must be BIT-identical to an uninterrupted
    two-epoch two-rank run: Adam moments and step count survived the resume (a re-bound flat buffer would zero them) and
    the gradients are still exchanged afterwards (a reducer left on a stale gradient buffer would leave rank 0 on its own
    shard's gradients)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ADN_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0',
               PYTHONPATH=root + os.pathsep + os.environ.get('PYTHONPATH', ''))

    def launch(exp, extra):
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
               '127.0.0.1', '--master-port', str(_free_port()), '-m', 'audio_depth_estimation_amd.train_binaural_attention',
               '--synthetic', '8', '--batch_size', '2', '--base_channels', '8', '--save_frequency', '1', '--precision', 'f32',
               '--experiment_name', exp] + extra
        r = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        return r.stdout

    launch('resume', ['--nb_epochs', '1'])
    out = launch('resume', ['--nb_epochs', '2', '--checkpoints', '1'])
    assert 'Loaded checkpoint from epoch 1' in out
    launch('straight', ['--nb_epochs', '2'])
    a = torch.load(os.path.join(tmp_path, 'checkpoints', 'resume', 'epoch_0002.pth'), map_location='cpu')
    b = torch.load(os.path.join(tmp_path, 'checkpoints', 'straight', 'epoch_0002.pth'), map_location='cpu')
    sa, sb = a['optimizer_state_dict']['state'], b['optimizer_state_dict']['state']
    assert a['epoch'] == b['epoch'] == 2 and float(sa[0]['step']) == float(sb[0]['step']) == 4     # 2 steps / epoch / rank
    for k in a['model_state_dict']:
        assert torch.equal(a['model_state_dict'][k], b['model_state_dict'][k]), k
    for i in sa:
        assert torch.equal(sa[i]['exp_avg'], sb[i]['exp_avg']) and torch.equal(sa[i]['exp_avg_sq'], sb[i]['exp_avg_sq']), i
    assert float(sa[0]['exp_avg'].abs().max()) > 0


# ---- RCCL itself: backend 'nccl' (= RCCL on ROCm), world size 1, in a fresh child process ----------------------------
def _rccl_worker(port, dtype_name, plan, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        from audio_depth_estimation_amd.ddp import GradientAllReducer
        from audio_depth_estimation_amd.engine import FusedTrainer
        assert dist.get_backend() == 'nccl'
        dtype = getattr(torch, dtype_name)
        results = []
        for use_ddp in (True, False):
            model, targs, cin = _make('unet')
            model.compute_dtype = dtype
            red = GradientAllReducer(bucket_bytes=1 << 20) if use_ddp else None     # several buckets for unet_128
            tr = FusedTrainer(model.engine(), ddp=red, **targs)
            if use_ddp:
                model.engine().bind_parameters()
                red.broadcast_parameters(model.engine().flat_p)
                if plan:
                    tr.enable_launch_plan(after_steps=1)          # collectives replayed from inside the launch plan
            x, gt = _shard(0, cin)
            losses = []
            for _ in range(4):
                loss, _ = tr.step(x, gt)
                losses.append(float(loss))
            torch.cuda.synchronize()
            nb = len(red.buckets) if use_ddp else 0
            results.append((losses, model.engine().flat_p.detach().cpu().numpy().tobytes(), nb))
        out.put(results)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('dtype_name', ['float32', 'bfloat16'])
@pytest.mark.parametrize('plan', [False, True])
def test_rccl_backend_world1_matches_plain_step(dtype_name, plan):
    """The data-parallel step over the REAL RCCL backend (torch.distributed 'nccl'): asynchronous bucketed all-reduce
    launched from the gradient-ready watermark on RCCL's stream, finish() ordering it before the clip / optimizer
    kernels, the 4-double loss-statistics all-reduce on device doubles.  With one rank every collective is the
    identity, so four steps must leave bit-identical losses and parameters to the plain (no reducer) trainer -- any
    missing stream dependency between RCCL's stream and the compute stream would show as a difference."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), dtype_name, plan, q))
    p.start()
    (l_ddp, p_ddp, nb), (l_ref, p_ref, _) = q.get(timeout=600)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert nb >= 3                                   # the exchange really was bucketed
    assert l_ddp == l_ref
    assert p_ddp == p_ref
