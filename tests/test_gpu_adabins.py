"""GPU parity of the AdaBins distillation model on libadn (models.adabins_distillation_model / adabins_engine).

  * csrc/adabins.hip kernels against torch-CPU fp32 references of the reference's formulas
    (adabins_distillation_model.py:127-149, 198-201; utils_distillation_loss.py:48-143): <= 2e-5 of max|ref| (f32),
    bf16-stored tensors <= 6e-3;
  * the whole model + fused distillation step against the golden vectors produced by the REFERENCE at its only
    valid width (base_channels 64, tests/golden/adabins32_bc64.npz): f32 compute; forward outputs <= 2e-4, loss terms
    <= 2e-4, per-parameter gradient norm <= 5e-3 and sampled gradient entries <= 5e-3 of the tensor max, decoder
    BatchNorm running statistics (updated TWICE per step like the reference's double decoder pass) <= 1e-4.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_oracle_golden import _sample, adabins_initial_state

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
DEV = 'cuda'


def K():
    from audio_depth_estimation_amd import kernels
    return kernels


def rel_err(a, b):
    a, b = a.detach().float().cpu(), torch.as_tensor(b).detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def nhwc(x, dtype):
    return x.permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV)


def test_bin_predictor_fwd_bwd():
    torch.manual_seed(0)
    B, Cb, Hd, nb, maxd = 3, 512, 256, 128, 30.0
    g = torch.randn(B, Cb).requires_grad_(True)
    W1, b1 = (torch.randn(Hd, Cb) * 0.05).requires_grad_(True), (torch.randn(Hd) * 0.1).requires_grad_(True)
    W2, b2 = (torch.randn(nb, Hd) * 0.1).requires_grad_(True), (torch.randn(nb) * 0.1).requires_grad_(True)
    mask = (torch.rand(B, Hd) > 0.1)
    h = F.relu(F.linear(g, W1, b1)) * mask / 0.9
    w = torch.softmax(F.linear(h, W2, b2), 1)
    edges = torch.cat([torch.zeros(B, 1), torch.cumsum(w, 1)], 1) * maxd
    cent = (edges[:, :-1] + edges[:, 1:]) / 2
    dc = torch.randn(B, nb)
    cent.backward(dc)
    k = K()
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=DEV)
    h1, wd, cd = f(B, Hd), f(B, nb), f(B, nb)
    dev = lambda t: t.detach().to(DEV)
    k.binpred_fwd(dev(g), dev(W1), dev(b1), dev(W2), dev(b2), mask.to(torch.uint8).to(DEV), 0.1, maxd, h1, wd, cd)
    assert rel_err(wd, w) <= 2e-5 and rel_err(cd, cent) <= 2e-5 and rel_err(h1, h) <= 2e-5
    dW2p, db2p, dW1p, db1p, dg = f(B, nb * Hd), f(B, nb), f(B, Hd * Cb), f(B, Hd), f(B, Cb)
    k.binpred_bwd(dc.to(DEV), wd, h1, dev(g), dev(W1), dev(W2), True, 0.1, maxd, dW2p, db2p, dW1p, db1p, dg)
    assert rel_err(dW2p.sum(0).view(nb, Hd), W2.grad) <= 5e-5
    assert rel_err(db2p.sum(0), b2.grad) <= 5e-5
    assert rel_err(dW1p.sum(0).view(Hd, Cb), W1.grad) <= 5e-5
    assert rel_err(db1p.sum(0), b1.grad) <= 5e-5
    assert rel_err(dg, g.grad) <= 5e-5
    # no dropout
    k.binpred_fwd(dev(g), dev(W1), dev(b1), dev(W2), dev(b2), None, 0.1, maxd, h1, wd, cd)
    w0 = torch.softmax(F.linear(F.relu(F.linear(g, W1, b1)), W2, b2), 1)
    assert rel_err(wd, w0) <= 2e-5


def test_dropout_mask_statistics():
    k = K()
    m = torch.empty(1 << 20, dtype=torch.uint8, device=DEV)
    k.dropout_mask(m, 0.1, 12345)
    keep = float(m.float().mean())
    assert abs(keep - 0.9) < 2e-3
    m2 = torch.empty_like(m)
    k.dropout_mask(m2, 0.1, 12346)
    assert float((m != m2).float().mean()) > 0.1          # a different seed gives a different draw
    k.dropout_mask(m2, 0.1, 12345)
    assert torch.equal(m, m2)                             # same seed: same draw (replayable)
    cnt = torch.zeros(1, dtype=torch.float64, device=DEV)
    k.dropout_mask(m, 0.1, 12345, cnt)
    cnt += 1                                              # device-side step count: the next (graph) replay differs
    k.dropout_mask(m2, 0.1, 12345, cnt)
    assert float((m != m2).float().mean()) > 0.1 and abs(float(m2.float().mean()) - 0.9) < 2e-3


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_soft_binning_fwd_bwd(dtype):
    torch.manual_seed(1)
    B, nb, H, W = 2, 128, 9, 7
    logits = (torch.randn(B, nb, H, W) * 2).to(dtype).float().requires_grad_(True)
    cent = (torch.rand(B, nb).cumsum(1)).requires_grad_(True)
    p = torch.softmax(logits, 1)
    base = (p * cent[:, :, None, None]).sum(1, keepdim=True)
    dbase = torch.randn_like(base)
    dmean = torch.randn(B, nb)
    (base * dbase).sum().backward(retain_graph=True)
    (logits.mean((2, 3)) * dmean).sum().backward()
    k = K()
    ld = nhwc(logits.detach(), dtype)
    bd = torch.empty(B * H * W, dtype=torch.float32, device=DEV)
    k.bins_fwd(ld, cent.detach().to(DEV), bd)
    assert rel_err(bd.view(B, 1, H, W), base) <= 2e-5
    dl = torch.full((B, H, W, nb), float('nan'), dtype=dtype, device=DEV)
    dc = torch.empty(B, nb, dtype=torch.float32, device=DEV)
    ws = torch.empty(k.bins_bwd_workspace_bytes(B, H * W, nb) // 4, dtype=torch.float32, device=DEV)
    k.bins_bwd(ld, cent.detach().to(DEV), bd, dbase.reshape(-1).to(DEV), dmean.to(DEV), dl, dc, ws)
    assert rel_err(dl.float().cpu().permute(0, 3, 1, 2), logits.grad) <= (2e-5 if dtype == torch.float32 else 6e-3)
    assert rel_err(dc, cent.grad) <= 2e-5


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_pool_and_feature_cosine(dtype):
    torch.manual_seed(2)
    B, C, H, W = 2, 24, 6, 10
    a = torch.randn(B, C, H, W).to(dtype).float().requires_grad_(True)
    r = torch.randn(B, C, H, W).to(dtype).float()
    k = K()
    ad, rd = nhwc(a.detach(), dtype), nhwc(r, dtype)
    ws = torch.empty(k.pool_workspace_bytes(B, H * W, C, 3) // 4, dtype=torch.float32, device=DEV)
    mean = torch.empty(B, C, dtype=torch.float32, device=DEV)
    k.pool(ad, None, B, H * W, C, 1, 1.0 / (H * W), mean, ws)
    assert rel_err(mean, a.detach().mean((2, 3))) <= 2e-5
    st = torch.empty(B, 3, C, dtype=torch.float32, device=DEV)
    k.pool(ad, rd, B, H * W, C, 3, 1.0, st, ws)
    cos = (F.normalize(a.flatten(2), dim=2) * F.normalize(r.flatten(2), dim=2)).sum(2)
    loss = 1 - cos.mean()
    loss.backward()
    stc = st.cpu()
    got_cos = stc[:, 2] / (stc[:, 0].sqrt() * stc[:, 1].sqrt())
    assert rel_err(got_cos, cos) <= 2e-5
    base = torch.randn(B, C, H, W).to(dtype).float()
    ga = nhwc(base, dtype)
    k.featcos_grad(ad, rd, st, -1.0 / (B * C), ga)
    want = base + a.grad
    assert rel_err(ga.float().cpu().permute(0, 3, 1, 2), want) <= (2e-5 if dtype == torch.float32 else 6e-3)


@pytest.mark.parametrize('teacher', [True, False])
def test_distillation_pixel_and_small_terms(teacher):
    """Pixel terms (masked L1 / MSE to teacher / |residual|), KL of mean logits, centre MSE, total and gradients."""
    from oracle import dcnet_oracle
    torch.manual_seed(3)
    B, nb, H, W, maxd = 2, 16, 8, 8, 30.0
    base = (torch.rand(B, 1, H, W) * 32 - 1).requires_grad_(True)
    resid = (torch.randn(B, 1, H, W) * 0.5).requires_grad_(True)
    gt = torch.rand(B, 1, H, W) * 30
    gt[gt < 5] = 0
    tfinal = torch.rand(B, 1, H, W) * 30
    ms = torch.randn(B, nb).requires_grad_(True)
    mt = torch.randn(B, nb)
    cs = torch.rand(B, nb).cumsum(1).requires_grad_(True)
    ct = torch.rand(B, nb).cumsum(1)
    lam = (1.0, 0.5, 0.3, 0.2, 0.1)
    final = torch.clamp(base + resid, 0, maxd)
    # oracle's loss on a hand-built output dict: logits whose spatial mean is ms / mt, no feature term (checked above)
    feats = {f'x{i}': torch.ones(B, 4, 2, 2) for i in range(1, 6)}
    out = {'audio': {'final_depth': final, 'features': feats, 'bin_logits': ms[:, :, None, None].expand(B, nb, 2, 2),
                     'bin_centers': cs, 'residual': resid},
           'rgb': {'final_depth': tfinal, 'features': feats, 'bin_logits': mt[:, :, None, None].expand(B, nb, 2, 2),
                   'bin_centers': ct} if teacher else None}
    total, parts = dcnet_oracle.distillation_loss(out, gt, gt > 0, *lam, 4.0)
    total.backward()
    k = K()
    dev = lambda t: t.detach().contiguous().to(DEV)
    n = B * H * W
    stats = torch.zeros(4, dtype=torch.float64, device=DEV)
    ws = torch.empty(8192, dtype=torch.float32, device=DEV)
    fo = torch.empty(n, dtype=torch.float32, device=DEV)
    tf = dev(tfinal).view(-1) if teacher else None
    k.distill_pix_stats(dev(base).view(-1), dev(resid).view(-1), dev(gt).view(-1), tf, maxd, fo, stats, ws)
    assert rel_err(fo.view(B, 1, H, W), final) <= 1e-6
    fst = [torch.tensor([[[4.0] * 4, [4.0] * 4, [4.0] * 4]] * B, device=DEV) for _ in range(5)]     # cos = 1 everywhere
    terms = torch.zeros(8, dtype=torch.float32, device=DEV)
    dmean, dcent = torch.empty(B, nb, device=DEV), torch.empty(B, nb, device=DEV)
    k.distill_small(dev(ms), dev(mt) if teacher else None, dev(cs), dev(ct) if teacher else None, fst, [4] * 5, stats, 4.0,
                    lam, terms, dmean, dcent)
    want = [float(parts[q]) for q in ('task', 'response', 'feature', 'bin', 'bin_centers', 'sparse')] + [float(total)]
    np.testing.assert_allclose(terms[:7].cpu().numpy(), np.array(want), rtol=2e-5, atol=2e-6)
    db, dr = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    k.distill_pix_grad(dev(base).view(-1), dev(resid).view(-1), dev(gt).view(-1), tf, maxd, stats, lam[0],
                       lam[1] if teacher else 0.0, lam[4], db, dr)
    assert rel_err(db.view(B, 1, H, W), base.grad) <= 2e-5
    assert rel_err(dr.view(B, 1, H, W), resid.grad) <= 2e-5
    if teacher:
        assert rel_err(dmean, ms.grad) <= 2e-5
        assert rel_err(dcent, cs.grad) <= 2e-5
    else:
        assert float(dmean.abs().max()) == 0.0 and float(dcent.abs().max()) == 0.0


def test_adabins_golden_reference_parity_f32():
    from audio_depth_estimation_amd.adabins_engine import AdaBinsTrainer
    z = np.load(os.path.join(GOLDEN, 'adabins32_bc64.npz'))
    lr, max_depth, lt, lr_, lf, lb, ls, temp = [float(v) for v in z['hyper']]
    model = adabins_initial_state(z)
    model.compute_dtype = torch.float32
    model = model.to(DEV)
    audio, rgb, gt = [torch.from_numpy(z[k]).to(DEV) for k in ('audio', 'rgb', 'gt')]
    model.eval()
    o = model(audio, rgb=None, mode='inference')
    assert o['rgb'] is None
    assert rel_err(o['audio']['final_depth'], z['eval/final_depth']) <= 2e-4
    assert rel_err(o['audio']['bin_centers'], z['eval/bin_centers']) <= 2e-4
    model.train()
    tr = AdaBinsTrainer(model.engine(), lt, lr_, lf, lb, ls, temp, optimizer='AdamW', lr=lr, clip_norm=1.0)
    total, terms = tr.step(audio, rgb, gt)
    eng = model.engine()
    for side in ('audio', 'rgb'):
        br = eng.branches[side]
        shp = (audio.shape[0], 1, audio.shape[2], audio.shape[3])
        got = {'bin_centers': br.centers, 'bin_widths': br.widths, 'base_depth': br.base.view(shp),
               'residual': br.head.result.view(shp), 'final_depth': br.final, 'logits_mean': br.mean_logits}
        for k, v in got.items():
            assert rel_err(v, z[f'train/{side}/{k}']) <= 2e-4, (side, k, rel_err(v, z[f'train/{side}/{k}']))
    np.testing.assert_allclose(terms[:6].cpu().numpy(), z['loss_parts'], rtol=2e-4, atol=1e-6)
    assert abs(float(total) - float(z['loss'])) <= 2e-4 * abs(float(z['loss']))
    named = dict(model.named_parameters())
    gkeys = [k[len('gnorm/'):] for k in z.files if k.startswith('gnorm/')]
    tot = 0.0
    for k in gkeys:
        g = eng.grad_view(named[k])
        tot += float(g.double().norm()) ** 2
        assert abs(float(g.double().norm()) - float(z['gnorm/' + k])) <= 5e-3 * float(z['gnorm/' + k]) + 1e-7, k
        ref = z['gs/' + k]
        assert float(np.abs(_sample(g.cpu().contiguous()) - ref).max()) <= 1e-6 + 5e-3 * float(np.abs(ref).max()), k
    assert abs(tot ** 0.5 - float(z['grad_norm'])) <= 2e-3 * float(z['grad_norm'])
    assert abs(float(tr.state[3]) - float(z['grad_norm'])) <= 2e-3 * float(z['grad_norm'])
    sd1 = model.state_dict()
    for k, v in sd1.items():
        ref = z['sd1s/' + k]
        if not v.is_floating_point():
            assert int(v) == int(ref), k                   # decoder BNs: num_batches_tracked advanced by 2
        elif 'running_' in k:
            assert float(np.abs(_sample(v.cpu()) - ref).max()) <= 1e-4 * float(np.abs(ref).max()) + 1e-5, k
        elif k.startswith('rgb_'):
            np.testing.assert_array_equal(_sample(v.cpu()), ref, err_msg=k)       # the teacher is not touched
        else:
            assert float(np.abs(_sample(v.cpu()) - ref).max()) <= 2.02 * lr, k   # within one AdamW step everywhere


def test_distillation_loss_module_and_adaptive_schedule():
    """utils_distillation_loss mirror: forward values from a model output dict vs the reference's loss terms; the
    adaptive weight schedule vs the reference formulas (utils_distillation_loss.py:268-304)."""
    from audio_depth_estimation_amd.utils_distillation_loss import AdaptiveDistillationLoss, DistillationLoss
    from audio_depth_estimation_amd.adabins_engine import AdaBinsTrainer
    z = np.load(os.path.join(GOLDEN, 'adabins32_bc64.npz'))
    lr, max_depth, lt, lr_, lf, lb, ls, temp = [float(v) for v in z['hyper']]
    model = adabins_initial_state(z)
    model.compute_dtype = torch.float32
    model = model.to(DEV).train()
    audio, rgb, gt = [torch.from_numpy(z[k]).to(DEV) for k in ('audio', 'rgb', 'gt')]
    out = model(audio, rgb=rgb, mode='train')
    assert set(out['audio'].keys()) == {'features', 'bin_centers', 'bin_widths', 'bin_logits', 'base_depth', 'residual',
                                        'final_depth'}
    assert out['audio']['bin_logits'].shape == (2, 128, 32, 32) and out['audio']['features']['x5'].shape == (2, 512, 2, 2)
    assert rel_err(out['audio']['features']['x5'], z['train/audio/x5']) <= 2e-4
    assert rel_err(out['rgb']['features']['x5'], z['train/rgb/x5']) <= 2e-4
    crit = DistillationLoss(lt, lr_, lf, lb, ls, temp)
    total, parts = crit(out, gt, gt > 0)
    got = np.array([parts[k] for k in ('task', 'response', 'feature', 'bin', 'bin_centers', 'sparse')])
    np.testing.assert_allclose(got, z['loss_parts'], rtol=2e-4, atol=1e-6)
    assert abs(float(total) - float(z['loss'])) <= 2e-4 * abs(float(z['loss']))
    ad = AdaptiveDistillationLoss(max_epochs=200)
    for epoch, want in ((0, (2.0, 0.1, 0.05, 0.05)), (100, (2.5, 0.1 + 0.4 * 0.4 / 0.9, 0.3, 0.035)),
                        (200, (3.0, 0.5, 0.2, 0.02))):
        ad.set_epoch(epoch)
        w = ad.get_adaptive_weights()
        np.testing.assert_allclose([w['task'], w['response'], w['feature'], w['bin']], want, rtol=1e-6)
    tr = AdaBinsTrainer.from_criterion(model.engine(), ad, lr=1e-4)
    assert tr.lambdas[0] == 3.0 and tr.temperature == 4.0
    with pytest.raises(RuntimeError):
        model(audio.cpu(), rgb=None, mode='inference')


def test_adabins_graph_step_matches_eager():
    """hipGraph replay of the fused distillation step == eager launches from the same state (dropout p = 0)."""
    from audio_depth_estimation_amd.adabins_engine import AdaBinsTrainer
    from audio_depth_estimation_amd.models.adabins_distillation_model import AdaBinsDistillationModel
    g = torch.Generator().manual_seed(5)
    audio, rgb = torch.rand(2, 2, 32, 32, generator=g).to(DEV), torch.rand(2, 3, 32, 32, generator=g).to(DEV)
    gt = (30 * torch.rand(2, 1, 32, 32, generator=g)).to(DEV)
    finals = []
    for graph in (False, True):
        torch.manual_seed(0)
        model = AdaBinsDistillationModel(128, 64, 32, 30.0)
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        model.compute_dtype = torch.bfloat16
        model = model.to(DEV).train()
        tr = AdaBinsTrainer(model.engine(), lr=1e-3)
        if graph:
            tr.enable_graph(after_steps=2)
        for _ in range(5):
            loss, _ = tr.step(audio, rgb, gt)
        torch.cuda.synchronize()
        finals.append((float(loss), model.engine().flat_p.detach().clone()))
    assert abs(finals[1][0] - finals[0][0]) <= 1e-6 * abs(finals[0][0])
    assert torch.equal(finals[0][1], finals[1][1])


@pytest.mark.parametrize('out_size', [64, 24])
def test_adabins_forward_with_output_size_different_from_the_input(out_size):
    """output_size != input size (adabins_distillation_model.py:196-198, 334-337, 383-386: F.interpolate(mode='nearest') of the
    logits / raw residual): the forward dict of both branches against the float64 oracle that resizes exactly where the
    reference does -- up (32 -> 64) and down (32 -> 24, non-integer ratio).  The engine resizes the per-pixel RESULTS instead
    (softmax expectation, tanh and clamp commute with a nearest resize).  The fused training step refuses the combination."""
    from audio_depth_estimation_amd.adabins_engine import AdaBinsTrainer
    from audio_depth_estimation_amd.models.adabins_distillation_model import create_adabins_distillation_model
    from oracle import dcnet_oracle
    torch.manual_seed(3)
    model = create_adabins_distillation_model(n_bins=128, base_channels=64, output_size=out_size, max_depth=30.0)
    model.compute_dtype = torch.float32
    sd = {k: (v.detach().double() if v.is_floating_point() else v.detach().clone()) for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(4)
    audio, rgb = torch.rand(2, 2, 32, 32, generator=g), torch.rand(2, 3, 32, 32, generator=g)
    model = model.to(DEV).eval()
    with torch.no_grad():
        out = model(audio.to(DEV), rgb.to(DEV), mode='train')
        ref, _ = dcnet_oracle.adabins_forward(sd, audio.double(), rgb.double(), 30.0, training=False, output_size=out_size)
    for br in ('audio', 'rgb'):
        for key in ('bin_logits', 'base_depth', 'residual', 'final_depth'):
            got, want = out[br][key], ref[br][key]
            assert tuple(got.shape[-2:]) == (out_size, out_size) and got.shape == want.shape, (br, key, got.shape)
            assert rel_err(got, want) <= 2e-4, (br, key, rel_err(got, want))
        assert rel_err(out[br]['bin_centers'], ref[br]['bin_centers']) <= 2e-4
        assert out[br]['features']['x1'].shape[-1] == 32          # features stay at the input resolution
    model.train()
    tr = AdaBinsTrainer(model.engine(), lr=1e-4)
    total, terms = tr.step(audio.to(DEV), rgb.to(DEV), 30 * torch.rand(2, 1, out_size, out_size, device=DEV))
    assert bool(torch.isfinite(total)) and bool(torch.isfinite(model.engine().flat_g).all())     # (values: the tests below)
    with pytest.raises(RuntimeError, match='the model output is'):
        tr.step(audio.to(DEV), rgb.to(DEV), torch.rand(2, 1, 32, 32, device=DEV))


@pytest.mark.parametrize('teacher', [True, False])
def test_reference_style_autograd_loop_matches_the_fused_trainer(teacher):
    """train_adabins_distillation.py:445-456 as written -- outputs = model(audio, rgb, mode='train'); loss, _ =
    criterion(outputs, gt, gt > 0); loss.backward(); clip_grad_norm_; optimizer.step() -- on the mirror modules: the
    student's .grad equal the fused trainer's gradients (f32 compute: <= 2e-5 of each tensor's max), the teacher gets
    none, and two steps with torch.optim.AdamW land within 0.25 * lr of the fused trainer's parameters."""
    from audio_depth_estimation_amd.adabins_engine import AdaBinsTrainer
    from audio_depth_estimation_amd.models.adabins_distillation_model import AdaBinsDistillationModel
    from audio_depth_estimation_amd.utils_distillation_loss import DistillationLoss
    g = torch.Generator().manual_seed(23)
    audio, rgb = torch.rand(2, 2, 32, 32, generator=g).to(DEV), torch.rand(2, 3, 32, 32, generator=g).to(DEV)
    gt = (30 * torch.rand(2, 1, 32, 32, generator=g)).to(DEV)
    gt[:, :, :3] = 0.0
    lr = 1e-3

    def make():
        torch.manual_seed(4)
        m = AdaBinsDistillationModel(128, 64, 32, 30.0)
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0                                  # (the two paths draw their masks from different counters)
        m.compute_dtype = torch.float32
        m.freeze_rgb()
        return m.to(DEV).train()

    ma, mb = make(), make()
    crit = DistillationLoss(1.0, 0.5, 0.3, 0.2, 0.1, 4.0)
    opt = torch.optim.AdamW([p for p in ma.parameters() if p.requires_grad], lr=lr)
    tr = AdaBinsTrainer.from_criterion(mb.engine(), crit, lr=lr, clip_norm=1.0)
    r = rgb if teacher else None
    for it in range(2):
        opt.zero_grad()
        out = ma(audio, r, mode='train')
        assert out['audio']['final_depth'].requires_grad and not out['audio']['bin_widths'].requires_grad
        assert (out['rgb'] is None) == (not teacher)
        if teacher:
            assert not out['rgb']['final_depth'].requires_grad
        loss, parts = crit(out, gt, gt > 0)
        loss.backward()
        lt, terms = tr.step(audio, r, gt)
        assert abs(float(loss) - float(lt)) <= 1e-5 * abs(float(lt)), (it, float(loss), float(lt))
        if it == 0:
            off = mb.engine().train_offset
            for (k, p), (_, q) in zip(ma.named_parameters(), mb.named_parameters()):
                if not p.requires_grad:
                    assert p.grad is None, k
                    continue
                gb = mb.engine().grad_view(q)
                assert p.grad is not None, k
                assert float((p.grad - gb).abs().max()) <= 2e-5 * float(gb.abs().max()) + 1e-12, k
        torch.nn.utils.clip_grad_norm_([p for p in ma.parameters() if p.requires_grad], 1.0)
        opt.step()
    # (AdamW moves every element by about lr per step whatever its gradient's size, so elements whose tiny gradient
    #  differs in the last bits may differ by a step; the loss of step 2 above already agreed to 1e-5)
    for (k, p), (_, q) in zip(ma.named_parameters(), mb.named_parameters()):
        d = (p - q).abs()
        assert float(d.max()) <= 4.1 * lr and float(d.mean()) <= 0.02 * lr, (k, float(d.max()), float(d.mean()))
    with torch.no_grad():
        assert not ma(audio, None, mode='inference')['audio']['final_depth'].requires_grad


def test_autograd_reaches_every_returned_leaf():
    """Hand-made linear losses over the returned leaves (f32 compute).  (a) exact identity on one forward state:
    back-propagating w through final_depth equals back-propagating mask * w through base_depth and residual, mask =
    the clamp's pass band (adabins_distillation_model.py:389-391): <= 1e-5.  (b) per leaf (x3, x4, x5, bin_centers,
    bin_logits) the analytic directional derivative along an encoder weight direction against central differences
    (h = 5e-3; the ReLU / max-pool kinks and f32 noise leave 2-7 % at this step, a missing path would be 100 %)."""
    from audio_depth_estimation_amd.models.adabins_distillation_model import AdaBinsDistillationModel
    g = torch.Generator().manual_seed(29)
    audio = torch.rand(2, 2, 32, 32, generator=g).to(DEV)
    torch.manual_seed(6)
    m = AdaBinsDistillationModel(128, 64, 32, 30.0)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    m.compute_dtype = torch.float32
    m.freeze_rgb()
    m = m.to(DEV).train()
    student = [p for p in m.parameters() if p.requires_grad]

    def leaves():
        o = m(audio, None, mode='train')['audio']
        return {'x3': o['features']['x3'], 'x4': o['features']['x4'], 'x5': o['features']['x5'], 'centers': o['bin_centers'],
                'logits': o['bin_logits'], 'base': o['base_depth'], 'residual': o['residual'], 'final': o['final_depth']}

    def rnd(t, seed):
        return torch.randn(t.shape, generator=torch.Generator().manual_seed(seed)).to(DEV) / t.numel() ** 0.5

    # (a)
    lv = leaves()
    w = rnd(lv['final'], 1)
    s = lv['base'].detach() + lv['residual'].detach()
    mask = ((s >= 0) & (s <= m.max_depth)).float()
    assert 0 < float(mask.mean())
    ga = torch.autograd.grad((lv['final'] * w).sum(), student, allow_unused=True)
    ga = [t.clone() if t is not None else None for t in ga]
    lv = leaves()
    gb = torch.autograd.grad((lv['base'] * (mask * w)).sum() + (lv['residual'] * (mask * w)).sum(), student, allow_unused=True)
    for p, a, b in zip(student, ga, gb):
        assert (a is None) == (b is None)
        if a is not None:
            assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()) + 1e-12
    # (b)
    name = 'audio_encoder.down2.maxpool_conv.1.double_conv.0.weight'
    p = dict(m.named_parameters())[name]
    d = torch.randn(p.shape, generator=torch.Generator().manual_seed(7)).to(DEV)
    d /= d.norm()
    h = 5e-3
    for i, key in enumerate(('x3', 'x4', 'x5', 'centers', 'logits')):
        lv = leaves()
        wk = rnd(lv[key], 100 + i)
        gp, = torch.autograd.grad((lv[key] * wk).sum(), [p])
        analytic = float((gp.double() * d.double()).sum())
        with torch.no_grad():
            p.add_(h * d)
            up = float((leaves()[key] * wk).sum())
            p.sub_(2 * h * d)
            dn = float((leaves()[key] * wk).sum())
            p.add_(h * d)
        numeric = (up - dn) / (2 * h)
        assert abs(analytic - numeric) <= 0.15 * max(abs(numeric), abs(analytic)) + 1e-3, (key, analytic, numeric)


@pytest.mark.parametrize('shape,S', [((3, 8, 8), 16), ((2, 32, 32), 24), ((1, 12, 20), 7), ((2, 16, 16), 40)])
def test_resize_nearest_backward(shape, S):
    """adn_resize_nearest_bwd against torch's autograd of F.interpolate(mode='nearest') (exact: a sum of copies)."""
    g = torch.Generator().manual_seed(31)
    x = torch.randn(1, *shape, generator=g, requires_grad=True)
    y = F.interpolate(x, size=(S, S), mode='nearest')
    go = torch.randn(y.shape, generator=g)
    y.backward(go)
    got = K().resize_nearest_bwd(go[0].to(DEV), shape[1], shape[2])
    np.testing.assert_allclose(got.cpu().numpy(), x.grad[0].numpy(), rtol=1e-6, atol=1e-6)


def _adabins_pair(out_size, seed=11):
    from audio_depth_estimation_amd.models.adabins_distillation_model import AdaBinsDistillationModel
    torch.manual_seed(seed)
    m = AdaBinsDistillationModel(128, 64, out_size, 30.0)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    m.compute_dtype = torch.float32
    m.freeze_rgb()
    return m.to(DEV).train()


def test_fused_step_with_output_size_twice_the_input_equals_the_native_step():
    """output_size = 2 x input: every source pixel is read by exactly four output pixels, so with the target repeated
    2 x 2 every loss term, every gradient and the AdamW update equal those of the output_size == input step (f32)."""
    from audio_depth_estimation_amd.adabins_engine import AdaBinsTrainer
    g = torch.Generator().manual_seed(37)
    audio, rgb = torch.rand(2, 2, 32, 32, generator=g).to(DEV), torch.rand(2, 3, 32, 32, generator=g).to(DEV)
    gt = (30 * torch.rand(2, 1, 32, 32, generator=g)).to(DEV)
    gt[:, :, :5] = 0.0
    gt2 = gt.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
    res = []
    for S, target in ((32, gt), (64, gt2)):
        m = _adabins_pair(S)
        tr = AdaBinsTrainer(m.engine(), lr=1e-3, clip_norm=1.0)
        total, terms = tr.step(audio, rgb, target)
        eng = m.engine()
        res.append((terms.detach().cpu().numpy().copy(), eng.flat_g[eng.train_offset:].detach().clone(),
                    eng.flat_p.detach().clone()))
    np.testing.assert_allclose(res[1][0][:7], res[0][0][:7], rtol=2e-5, atol=1e-7)
    ga, gb = res[0][1], res[1][1]
    assert float((ga - gb).abs().max()) <= 2e-5 * float(ga.abs().max())
    assert float((res[0][2] - res[1][2]).abs().max()) <= 2.1e-3            # (elements with ~0 gradient may step the other way)
    assert float((res[0][2] - res[1][2]).abs().mean()) <= 2e-5


def test_fused_step_and_autograd_loop_agree_for_a_fractional_resize():
    """input 32 x 32, output_size 24 (a 0.75 gather: some source pixels are never read): the fused step's loss terms equal
    DistillationLoss on the model's 24 x 24 outputs, and its gradients equal those of the reference-style autograd loop
    (criterion gradients at 24 x 24, back through the gather)."""
    from audio_depth_estimation_amd.adabins_engine import AdaBinsTrainer
    from audio_depth_estimation_amd.utils_distillation_loss import DistillationLoss
    g = torch.Generator().manual_seed(41)
    audio, rgb = torch.rand(2, 2, 32, 32, generator=g).to(DEV), torch.rand(2, 3, 32, 32, generator=g).to(DEV)
    gt = (30 * torch.rand(2, 1, 24, 24, generator=g)).to(DEV)
    gt[:, :, :2] = 0.0
    crit = DistillationLoss(1.0, 0.5, 0.3, 0.2, 0.1, 4.0)
    ma, mb = _adabins_pair(24), _adabins_pair(24)
    out = ma(audio, rgb, mode='train')
    assert out['audio']['final_depth'].shape == (2, 1, 24, 24) and out['audio']['final_depth'].requires_grad
    loss, parts = crit(out, gt, gt > 0)
    loss.backward()
    tr = AdaBinsTrainer.from_criterion(mb.engine(), crit, lr=1e-3, clip_norm=1.0)
    total, terms = tr.step(audio, rgb, gt)
    assert abs(float(loss) - float(total)) <= 1e-5 * abs(float(total))
    for (k, p), (_, q) in zip(ma.named_parameters(), mb.named_parameters()):
        if not p.requires_grad:
            continue
        gb = mb.engine().grad_view(q)
        assert float((p.grad - gb).abs().max()) <= 2e-5 * float(gb.abs().max()) + 1e-12, k
