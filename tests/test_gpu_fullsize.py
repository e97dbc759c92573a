"""Size-independent properties at BASELINE.json's FULL headline configuration (unet_256, ngf 64, batch 32, 256x256) --
the oracle is too slow there, so the HIP path is checked through what must hold at any size:

  * determinism: two fused train steps from the same state are bit-identical (no float atomics anywhere);
  * linearity of the backward pass in the upstream gradient (forward fixed): grads(a u + b v) = a grads(u) + b grads(v)
    -- exercises every dgrad / wgrad / BN-backward kernel at the real tile counts and split-K factors;
  * eval-mode batch independence: sample i of a batch of 32 equals the same sample run alone (different tilings /
    split counts of the same layers must agree);
  * the fused loss + clip + AdamW tail against torch on the very same prediction / gradients.
f32 compute runs the exact f32 MFMA path (tight bounds); bf16 bounds are the rounding of the stored activations.
"""
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'
B, S = 32, 256


def _model(dtype):
    from audio_depth_estimation_amd.models.unetbaseline_model import define_G
    torch.manual_seed(0)
    m = define_G(SimpleNamespace(dataset=SimpleNamespace(depth_norm=False, max_depth=30.0)), 2, 1, 64, 'unet_256')
    m.compute_dtype = dtype
    return m.to(DEV)


def _batch():
    g = torch.Generator().manual_seed(1234)
    audio = torch.rand(B, 2, S, S, generator=g)
    gt = 30 * torch.rand(B, 1, S, S, generator=g)
    gt[gt < 3] = 0
    return audio.to(DEV), gt.to(DEV)


def _rel(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-30))


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_backward_is_linear_in_the_upstream_gradient(dtype):
    model = _model(dtype).train()
    eng = model.engine()
    audio, _ = _batch()
    pred = eng.forward(audio, True)
    g = torch.Generator().manual_seed(5)
    u = torch.randn(pred.shape, generator=g).to(DEV) / pred.numel()
    v = torch.randn(pred.shape, generator=g).to(DEV) / pred.numel()
    grads = []
    for up in (u, v, 0.7 * u - 1.9 * v):
        eng.backward(up)
        grads.append(eng.flat_g.clone())
    want = 0.7 * grads[0] - 1.9 * grads[1]
    tol = 2e-5 if dtype == torch.float32 else 3e-2          # bf16: every dz tensor is rounded to 8 bits of mantissa
    assert _rel(grads[2], want) <= tol
    for p, off, n in eng.param_meta:                         # and per tensor, so a small layer cannot hide
        a, b = grads[2][off:off + n], want[off:off + n]
        if float(b.norm()) > 0:
            assert _rel(a, b) <= 10 * tol, (tuple(p.shape), _rel(a, b))


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_eval_batch_independence(dtype):
    model = _model(dtype).eval()
    audio, _ = _batch()
    with torch.no_grad():
        full = model(audio).clone()
        for i in (0, 17, 31):
            one = model(audio[i:i + 1]).clone()
            err = float((one - full[i:i + 1]).abs().sum() / (full[i:i + 1].abs().sum() + 1e-30))
            assert err <= (1e-5 if dtype == torch.float32 else 2e-2), (i, err)


def test_fused_step_is_deterministic_and_tail_matches_torch():
    from audio_depth_estimation_amd.engine import FusedTrainer
    audio, gt = _batch()
    finals, losses = [], []
    for _ in range(2):
        model = _model(torch.bfloat16).train()
        tr = FusedTrainer(model.engine(), 'Combined', 0.5, 0.5, 0.5, max_depth=30.0, optimizer='AdamW', lr=2e-3, clip_norm=1.0)
        for _ in range(2):
            loss, pred = tr.step(audio, gt)
        losses.append(float(loss))
        finals.append(model.engine().flat_p.clone())
    assert losses[0] == losses[1] and torch.equal(finals[0], finals[1])
    # loss / clip / AdamW tail vs torch on the same prediction and gradients (first step of a fresh model)
    model = _model(torch.bfloat16).train()
    eng = model.engine()
    tr = FusedTrainer(eng, 'Combined', 0.5, 0.5, 0.5, max_depth=30.0, optimizer='AdamW', lr=2e-3, clip_norm=1.0)
    eng.bind_parameters()
    p0 = eng.flat_p.clone()
    loss, pred = tr.step(audio, gt)
    p = pred.detach().float().clone().requires_grad_(True)
    mask = gt != 0
    pp, gg = p[mask], gt[mask]
    d = torch.log(pp.clamp_min(1e-6)) - torch.log(gg.clamp_min(1e-6))
    ref = 0.5 * (pp - gg).abs().mean() + 0.5 * torch.sqrt((d ** 2).mean() - 0.5 * d.mean() ** 2)
    assert abs(float(loss) - float(ref)) <= 1e-5 * abs(float(ref))
    ref.backward()
    assert _rel(tr.loss_gradient(), p.grad) <= 1e-5
    target = eng.dz_target()
    if target is not None:   # the loss kernel wrote d loss / d pre-activation directly (adn_loss_finish_dz): gout * act'(pred)
        dz, _, final_act = target
        dact = p.detach() * (1 - p.detach()) if final_act == 1 else (p.detach() > 0).float()
        assert _rel(dz, p.grad * dact) <= 1e-5
    gflat = eng.flat_g.clone()                                   # gradients of this step (pre-clip values)
    norm = float(gflat.double().norm())
    gcl = gflat * min(1.0, 1.0 / (norm + 1e-6))
    m = 0.1 * gcl
    vv = 0.001 * gcl * gcl
    want = p0 * (1 - 2e-3 * 0.01) - 2e-3 * (m / (1 - 0.9)) / ((vv / (1 - 0.999)).sqrt() + 1e-8)
    err = (eng.flat_p - want).abs()
    big = gcl.abs() > 1e-7                                       # where Adam's step is well conditioned
    assert float(err[big].max()) <= 0.02 * 2e-3 and float(err.max()) <= 2.02 * 2e-3


def _attn_buffers(B2, N, dqk, dv, seed):
    T = torch.bfloat16
    ld = (2 * dqk + dv + 63) // 64 * 64
    g = torch.Generator().manual_seed(seed)
    qkv = (0.5 * torch.randn(B2, N, ld, generator=g)).to(T).to(DEV)
    return qkv, qkv[:, :, :dqk], qkv[:, :, dqk:2 * dqk], qkv[:, :, 2 * dqk:2 * dqk + dv]


def test_attention_full_length_properties():
    """Level-2 cross-attention at its real length (16384 tokens, d_qk 16, d_v 128, MFMA kernels): softmax rows sum to 1
    (constant V comes back unchanged whatever Q and K are), the key/value batch shift pairs entry b with entry
    (b + shift) % B2, and the backward is linear in dO with dQ = dK = 0 for a constant V... (dP - delta vanishes)."""
    from audio_depth_estimation_amd import kernels as K
    B2, N, dqk, dv = 4, 16384, 16, 128
    T = torch.bfloat16
    qkv, q, k, v = _attn_buffers(B2, N, dqk, dv, 21)
    const = torch.linspace(-2, 2, dv, device=DEV).to(T)
    vals = torch.stack([const * (b + 1) for b in range(B2)])                 # entry b holds (b+1) * const in every row
    v.copy_(vals[:, None, :].expand(B2, N, dv))
    o = torch.empty(B2, N, dv, dtype=T, device=DEV)
    lse = torch.empty(B2, N, device=DEV)
    scale = 1.0 / dv ** 0.5
    K.attn_fwd(q, k, v, o, lse, dqk, dv, B2 // 2, scale)
    for b in range(B2):
        want = vals[(b + B2 // 2) % B2].float()
        assert float((o[b].float() - want).abs().max()) <= 2e-2 * float(want.abs().max()), b
    assert bool(torch.isfinite(lse).all())
    # backward with a constant V: dP = dO . V^T is the same for every key, so dS = P (dP - delta) = 0 -> dQ = dK = 0
    g = torch.Generator().manual_seed(22)
    do = torch.randn(B2, N, dv, generator=g).to(T).to(DEV)
    dqkv = torch.zeros_like(qkv)
    dq, dk, dvg = dqkv[:, :, :dqk], dqkv[:, :, dqk:2 * dqk], dqkv[:, :, 2 * dqk:2 * dqk + dv]
    ws = torch.empty(B2 * N, device=DEV)
    K.attn_bwd(q, k, v, o, lse, dqk, dv, B2 // 2, scale, do, dq, dk, dvg, ws)
    ref_scale = float(do.float().abs().mean())
    assert float(dq.float().abs().max()) <= 5e-2 * ref_scale and float(dk.float().abs().max()) <= 5e-2 * ref_scale
    # column sums: sum_k dV[kb][k] = sum_q dO[qb][q] (each softmax row sums to 1), queries of entry b use keys of b + shift
    for kb in range(B2):
        qb = (kb - B2 // 2) % B2
        got, want = dvg[kb].float().sum(0), do[qb].float().sum(0)
        assert float((got - want).abs().max()) <= 2e-2 * float(want.abs().max()) + 0.5, kb


def test_attention_backward_linearity_full_length():
    from audio_depth_estimation_amd import kernels as K
    B2, N, dqk, dv = 2, 16384, 16, 128
    T = torch.bfloat16
    qkv, q, k, v = _attn_buffers(B2, N, dqk, dv, 31)
    o = torch.empty(B2, N, dv, dtype=T, device=DEV)
    lse = torch.empty(B2, N, device=DEV)
    scale = 1.0 / dv ** 0.5
    K.attn_fwd(q, k, v, o, lse, dqk, dv, 1, scale)
    g = torch.Generator().manual_seed(32)
    u = torch.randn(B2, N, dv, generator=g).to(T).to(DEV)
    w = torch.randn(B2, N, dv, generator=g).to(T).to(DEV)
    ws = torch.empty(B2 * N, device=DEV)
    outs = []
    for do in (u, w, (u.float() + w.float()).to(T)):
        d = torch.zeros_like(qkv)
        K.attn_bwd(q, k, v, o, lse, dqk, dv, 1, scale, do.contiguous(), d[:, :, :dqk], d[:, :, dqk:2 * dqk],
                   d[:, :, 2 * dqk:2 * dqk + dv], ws)
        outs.append(d.float())
    want = outs[0] + outs[1]
    for lo, hi, name in ((0, dqk, 'dq'), (dqk, 2 * dqk, 'dk'), (2 * dqk, 2 * dqk + dv, 'dv')):
        assert _rel(outs[2][:, :, lo:hi], want[:, :, lo:hi]) <= 3e-2, name


@pytest.mark.parametrize('kind', ['rgb', 'binaural'])
def test_doubleconv_nets_full_size_linearity_and_determinism(kind):
    """BASELINE configs 3 / 5 shapes at 256^2, batch 32, bf16: RGBDepthNet and BinauralAttentionDepthNet (base 64)."""
    from audio_depth_estimation_amd.engine import FusedTrainer

    def make():
        torch.manual_seed(0)
        if kind == 'rgb':
            from audio_depth_estimation_amd.models.rgb_depth_model import RGBDepthNet
            m = RGBDepthNet(64, True, S, 30.0)
        else:
            from audio_depth_estimation_amd.models.binaural_attention_model import BinauralAttentionDepthNet
            m = BinauralAttentionDepthNet(64, True, S, 30.0)
            with torch.no_grad():
                for a in m.attention_modules.values():
                    a.gamma.fill_(0.5)
        m.compute_dtype = torch.bfloat16
        return m.to(DEV).train()

    g = torch.Generator().manual_seed(77)
    x = torch.rand(B, 3 if kind == 'rgb' else 2, S, S, generator=g).to(DEV)
    gt = (30 * torch.rand(B, 1, S, S, generator=g)).to(DEV)
    model = make()
    eng = model.engine()
    pred = eng.forward(x, True)
    u = torch.randn(pred.shape, generator=g).to(DEV) / pred.numel()
    v = torch.randn(pred.shape, generator=g).to(DEV) / pred.numel()
    grads = []
    for up in (u, v, 1.3 * u + 0.6 * v):
        eng.backward(up)
        grads.append(eng.flat_g.clone())
    # (bf16 rounding of every stored dz / dS; the binaural net adds four attention stages: measured 3.6e-2)
    assert _rel(grads[2], 1.3 * grads[0] + 0.6 * grads[1]) <= (3e-2 if kind == 'rgb' else 6e-2)
    del grads
    finals = []
    for _ in range(2):
        m = make()
        crit = dict(criterion='DepthLoss', l1_weight=1.0, silog_weight=0.1) if kind == 'rgb' else \
            dict(criterion='L1', mask_mode='gt0')
        tr = FusedTrainer(m.engine(), optimizer='AdamW', lr=1e-3, weight_decay=0.01, clip_norm=None, **crit)
        for _ in range(2):
            loss, _ = tr.step(x, gt)
        finals.append((float(loss), m.engine().flat_p.clone()))
        del m, tr
    assert finals[0][0] == finals[1][0] and torch.equal(finals[0][1], finals[1][1])


@pytest.mark.parametrize('kind', ['adabins', 'baseres'])
def test_distillation_trainers_full_size_determinism(kind):
    """BASELINE config 4 shape (AdaBins distillation, teacher forward + student step) and the Base+Residual sibling at
    batch 32, 256^2, bf16: two fresh runs of two fused steps are bit-identical and finite."""
    g = torch.Generator().manual_seed(78)
    audio, rgb = torch.rand(B, 2, S, S, generator=g).to(DEV), torch.rand(B, 3, S, S, generator=g).to(DEV)
    gt = 30 * torch.rand(B, 1, S, S, generator=g)
    gt[gt < 3] = 0
    gt = gt.to(DEV)
    finals = []
    for _ in range(2):
        torch.manual_seed(0)
        if kind == 'adabins':
            from audio_depth_estimation_amd.adabins_engine import AdaBinsTrainer
            from audio_depth_estimation_amd.models.adabins_distillation_model import AdaBinsDistillationModel
            m = AdaBinsDistillationModel(128, 64, S, 30.0)
            m.compute_dtype = torch.bfloat16
            m = m.to(DEV).train()
            tr = AdaBinsTrainer(m.engine(), lr=1e-4)
            step = lambda: tr.step(audio, rgb, gt)
        else:
            from audio_depth_estimation_amd.base_residual_engine import BaseResidualTrainer
            from audio_depth_estimation_amd.models.base_residual_model import BaseResidualDepthNet
            m = BaseResidualDepthNet(2, 64, True, S, 30.0)
            m.compute_dtype = torch.bfloat16
            m = m.to(DEV).train()
            tr = BaseResidualTrainer(m.engine(), use_silog=True, lr=1e-4)
            step = lambda: tr.step(audio, gt)
        for _ in range(2):
            loss, terms = step()
        assert bool(torch.isfinite(terms).all())
        finals.append((float(loss), m.engine().flat_p.clone()))
        del m, tr
    assert finals[0][0] == finals[1][0] and torch.equal(finals[0][1], finals[1][1])


# ---- reference-generated numbers at the headline size (tests/golden/make_golden_unet64.py b32) -----------------------
def _hash_key(key):
    h = 0
    for ch in key:
        h = (h * 131 + ord(ch)) % (2 ** 31 - 1)
    return h


def _sample_idx(numel, key, ns=512):
    g = torch.Generator().manual_seed(_hash_key(key))
    return torch.randint(0, numel, (min(ns, numel),), generator=g)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_b32_train_step_against_the_reference(dtype):
    """ONE full train step of unet_256 ngf 64 at B = 32 (BASELINE configs[1]) against numbers produced by the reference's
    own define_G / SIlogLoss / clip_grad_norm_ / AdamW on the CPU (unet256_ngf64_b32.npz): prediction and d loss / d pred
    samples, loss, every gradient tensor (norm + 512 samples), clipped norm, BatchNorm running statistics, AdamW step.
    This is "right", not just "linear and repeatable", at the tilings / split-K factors / slab sums that are benchmarked.
    f32: prediction rel-L1 <= 1e-4 (north_star; measured 1.3e-7), gradients <= 5e-3 of the tensor norm (sampled rel-L2 <= 2e-2:
    the reference's own fp32 noise floor at this depth; measured <= 5.8e-3).  bf16, against the reference too: prediction
    rel-L1 <= 3e-3 (5.1e-4), loss 1e-4, d loss / d pred 5e-3 (4.6e-4), every gradient tensor: sampled rel-L2 <= 0.4 (worst
    0.31, the innermost BatchNorm affine; <= 0.08 on the four outermost tensors) and norm within 3 % (<= 1.4 %).
    The fixture's output bias is 4.0: with 2 M predicted pixels and the bias at 1.0 a few land in (0, 1e-3), where SIlog's
    1 / pred lets ONE pixel set every gradient norm -- two bf16 runs one rounding apart then differ 5x."""
    import os

    import numpy as np
    from audio_depth_estimation_amd.engine import FusedTrainer
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'unet256_ngf64_b32.npz'))
    lr, max_depth, l1w, sw, lam, bias0 = [float(v) for v in z['hyper']]
    f32 = dtype == torch.float32
    model = _model(dtype)
    with torch.no_grad():
        model.model.model[3].bias.fill_(bias0)        # keeps every prediction away from SIlog's 1 / pred singularity (generator script)
    model.train()
    eng = model.engine()
    audio, gt = _batch()                                     # same seed / construction as the generator's synth_batch(32, 256, 1234)
    tr = FusedTrainer(eng, 'Combined', l1w, sw, lam, max_depth=max_depth, optimizer='AdamW', lr=lr, clip_norm=1.0)
    loss, pred = tr.step(audio, gt)
    idx = torch.from_numpy(z['idx'])
    got = pred.reshape(-1).float().cpu()[idx]
    ref = torch.from_numpy(z['pred_val'])
    relp = float((got - ref).abs().sum() / ref.abs().sum())
    assert relp <= (1e-4 if f32 else 3e-3), relp
    lrel = abs(loss.item() - float(z['loss'])) / abs(float(z['loss']))
    assert lrel <= (1e-5 if f32 else 1e-4), lrel
    dg = tr.loss_gradient().reshape(-1).float().cpu()[idx]
    dref = torch.from_numpy(z['pred_grad_val'])
    reld = float((dg - dref).abs().sum() / dref.abs().sum())
    assert reld <= (1e-4 if f32 else 5e-3), reld
    names = [k for k, _ in model.named_parameters()]
    worst = {}
    for k, prm in model.named_parameters():
        gflat = eng.grad_view(prm).detach().float().cpu().reshape(-1)
        gn_ref = float(z['gnorm/' + k])
        if gn_ref < 1e-12:
            continue
        si = _sample_idx(gflat.numel(), k)
        rms = gn_ref / (gflat.numel() ** 0.5)
        err = float((gflat[si] - torch.from_numpy(z['gsample/' + k])).norm() / (len(si) ** 0.5)) / rms
        nerr = abs(float(gflat.double().norm()) - gn_ref) / gn_ref
        worst[k] = (err, nerr)
    print(f'{dtype}: pred rel-L1 {relp:.3e} loss rel {lrel:.3e} dloss/dpred rel-L1 {reld:.3e}; worst gradients',
          sorted(worst.items(), key=lambda kv: -kv[1][0])[:8])
    for k, (err, nerr) in worst.items():
        if f32:
            assert err <= 2e-2 and nerr <= 5e-3, (k, err, nerr)
        else:
            assert err <= (0.08 if k in names[-4:] else 0.4) and nerr <= 3e-2, (k, err, nerr)
    gn = float(z['grad_norm'])
    assert abs(tr.state[3].item() - gn) <= (2e-3 if f32 else 2e-2) * gn
    if f32:
        for k, prm in model.named_parameters():
            si = _sample_idx(prm.numel(), k)
            gs = torch.from_numpy(z['gsample/' + k]).abs()
            m = gs > 1e-2 * gs.max()                       # Adam's sign-like step is ill-conditioned where g ~ 0
            d = (prm.detach().cpu().reshape(-1)[si] - torch.from_numpy(z['p1sample/' + k])).abs()[m]
            assert float(d.max()) <= 0.1 * lr, (k, float(d.max()) / lr)
    sd = model.state_dict()
    for k in z.files:
        if not k.startswith('sd1/'):
            continue
        ref_v = torch.from_numpy(z[k])
        gotv = sd[k[4:]].cpu()
        if ref_v.dtype == torch.int64:
            assert int(gotv) == int(ref_v), k
        else:
            assert float((gotv - ref_v).abs().max()) <= (1e-4 if f32 else 2e-2) * float(ref_v.abs().max()) + 1e-6, k


def test_bf16_training_tracks_the_exact_f32_path_over_300_steps():
    """the headline dtype must TRAIN like the exact path, not only match it for one step.  300 fused steps
    (unet_256 ngf 64, B = 32, Combined loss, clip 1.0, AdamW lr 2e-3 -- conf/mode/train.yaml) on a fixed synthetic stream of
    300 DIFFERENT batches (seeded on the device; depth = a smooth function of the spectrogram + 1 m of noise, ~10 % invalid
    pixels), once in bf16 and once on the exact-f32 MFMA path, from the same initial weights.  Stated band: the window means
    of the loss (12 windows of 25 steps) agree within 1 % at every window, and both runs descend by more than 4x
    (train.py:633-691).  Measured: <= 0.33 % (1.5617 -> 0.2342 in both).  (On a stream that repeats 8 batches the loss falls
    to 0.015 and the f32 run shows a loss spike at step ~200 that the bf16 run does not: the trajectories of an over-fitted
    run decorrelate, which says nothing about the arithmetic -- hence 300 different batches.)"""
    from audio_depth_estimation_amd.engine import FusedTrainer

    def batch(it):
        g = torch.Generator(device=DEV).manual_seed(4321 + it)
        audio = torch.rand(B, 2, S, S, generator=g, device=DEV)
        noise = torch.randn(B, 1, S, S, generator=g, device=DEV)
        drop = torch.rand(B, 1, S, S, generator=g, device=DEV) < 0.1
        gt = 3 + 24 * torch.nn.functional.avg_pool2d(audio.mean(1, keepdim=True), 9, 1, 4) + noise
        return audio, torch.where(drop, torch.zeros_like(gt), gt.clamp(3.0, 30.0))

    curves = {}
    for dtype in (torch.bfloat16, torch.float32):
        model = _model(dtype)
        with torch.no_grad():
            model.model.model[3].bias.fill_(4.0)
        model.train()
        tr = FusedTrainer(model.engine(), 'Combined', 0.237, 0.637, 0.869, max_depth=30.0, optimizer='AdamW', lr=0.002,
                          clip_norm=1.0)
        losses = torch.zeros(300, device=DEV)
        for it in range(300):
            loss, _ = tr.step(*batch(it))
            losses[it] = loss
        curves[dtype] = losses.cpu().view(12, 25).mean(1)
        del tr, model
        torch.cuda.empty_cache()
    a, b = curves[torch.bfloat16], curves[torch.float32]
    rel = ((a - b).abs() / b.abs()).tolist()
    print('bf16 windows', [round(v, 4) for v in a.tolist()])
    print('f32  windows', [round(v, 4) for v in b.tolist()])
    print('relative difference per window', [round(v, 4) for v in rel])
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
    assert float(b[-1]) < 0.25 * float(b[0]) and float(a[-1]) < 0.25 * float(a[0])      # both trained
    assert max(rel) <= 0.01, rel
