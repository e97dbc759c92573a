"""GPU parity of the Base + Residual model on libadn (models.base_residual_model / base_residual_engine).

  * the structural low-pass target (avg_pool2d k, stride 1, pad k/2 + bilinear resize back) against torch-CPU:
    <= 2e-6 of max|ref|;
  * the model and the fused BaseResidualLoss step against the golden vectors produced by the REFERENCE at its only
    valid width (base_channels 64, tests/golden/baseres32_bc64.npz), f32 compute, both reconstruction variants
    (SIlog -- the trainer's default -- and L1): outputs <= 2e-4, loss terms <= 2e-4, per-parameter gradient norm
    <= 5e-3 and sampled gradient entries <= 5e-3 of the tensor max.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_oracle_golden import _sample, base_residual_initial_state

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
DEV = 'cuda'


def rel_err(a, b):
    a, b = a.detach().float().cpu(), torch.as_tensor(b).detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize('shape,k', [((2, 32, 32), 16), ((1, 40, 56), 16), ((2, 64, 64), 7), ((1, 256, 256), 16)])
def test_lowpass_struct(shape, k):
    from audio_depth_estimation_amd import kernels as K
    B, H, W = shape
    torch.manual_seed(0)
    gt = torch.rand(B, 1, H, W) * 30
    gt[gt < 3] = 0
    s = F.avg_pool2d(gt, kernel_size=k, stride=1, padding=k // 2)
    ref = F.interpolate(s, size=(H, W), mode='bilinear', align_corners=False) if s.shape != gt.shape else s
    out = torch.empty(B, 1, H, W, device=DEV)
    ws = torch.empty(K.lowpass_workspace_bytes(B, H, W, k) // 4 + 4, device=DEV)
    K.lowpass(gt.to(DEV), k, out, ws)
    assert rel_err(out, ref) <= 2e-6


@pytest.mark.parametrize('tag,use_silog', [('silog', True), ('l1', False)])
def test_base_residual_golden_reference_parity_f32(tag, use_silog):
    from audio_depth_estimation_amd.base_residual_engine import BaseResidualTrainer
    z = np.load(os.path.join(GOLDEN, 'baseres32_bc64.npz'))
    lr, max_depth, lrec, lbase, lsp, k, slam = [float(v) for v in z['hyper']]
    model = base_residual_initial_state(z)
    model.compute_dtype = torch.float32
    model = model.to(DEV)
    audio, gt = torch.from_numpy(z['audio']).to(DEV), torch.from_numpy(z['gt']).to(DEV)
    model.eval()
    b, r, f = model(audio)
    for t, name in ((b, 'base'), (r, 'residual'), (f, 'final')):
        assert rel_err(t, z['eval/' + name]) <= 2e-4, name
    model.train()
    tr = BaseResidualTrainer(model.engine(), lrec, lbase, lsp, int(k), use_silog=use_silog, silog_lambda=slam,
                             optimizer='AdamW', lr=lr, clip_norm=1.0)
    total, terms = tr.step(audio, gt)
    eng = model.engine()
    if use_silog:
        for t, name in ((eng.head_base.result, 'base'), (eng.head_res.result, 'residual'), (eng.final, 'final')):
            assert rel_err(t, z['train/' + name]) <= 2e-4, name
        assert rel_err(tr.struct, z['struct']) <= 2e-6
    parts = z[tag + '/parts']
    got = terms.cpu().numpy()
    np.testing.assert_allclose([got[0] / lrec, got[1], got[2]], parts, rtol=2e-4, atol=1e-6)
    assert abs(float(total) - float(z[tag + '/loss'])) <= 2e-4 * abs(float(z[tag + '/loss']))
    named = dict(model.named_parameters())
    for kk in [q[len(tag + '/gnorm/'):] for q in z.files if q.startswith(tag + '/gnorm/')]:
        g = eng.grad_view(named[kk])
        want = float(z[f'{tag}/gnorm/' + kk])
        assert abs(float(g.double().norm()) - want) <= 5e-3 * want + 1e-7, (kk, float(g.double().norm()), want)
        ref = z[f'{tag}/gs/' + kk]
        assert float(np.abs(_sample(g.cpu().contiguous()) - ref).max()) <= 1e-6 + 5e-3 * float(np.abs(ref).max()), kk
    with pytest.raises(RuntimeError):
        model(audio.cpu())


def test_base_residual_loss_module_and_curriculum():
    """utils_base_residual_loss mirror: forward values from model outputs vs the reference's terms; the warm-up
    curriculum of AdaptiveBaseResidualLoss (utils_base_residual_loss.py:210-229)."""
    from audio_depth_estimation_amd.utils_base_residual_loss import AdaptiveBaseResidualLoss, BaseResidualLoss
    z = np.load(os.path.join(GOLDEN, 'baseres32_bc64.npz'))
    lr, max_depth, lrec, lbase, lsp, k, slam = [float(v) for v in z['hyper']]
    dev = lambda a: torch.from_numpy(a).to(DEV)
    gt = dev(z['gt'])
    for tag, use_silog in (('silog', True), ('l1', False)):
        crit = BaseResidualLoss(lrec, lbase, lsp, int(k), use_silog=use_silog, silog_lambda=slam)
        total, d = crit(dev(z['train/base']), dev(z['train/residual']), dev(z['train/final']), gt, gt > 0)
        np.testing.assert_allclose([d['recon'], d['base'], d['sparse']], z[tag + '/parts'], rtol=2e-4, atol=1e-6)
        assert abs(float(total) - float(z[tag + '/loss'])) <= 2e-4 * abs(float(z[tag + '/loss']))
    ad = AdaptiveBaseResidualLoss(lambda_recon_init=0.5, lambda_base_init=2.4, warmup_epochs=50)
    for epoch, want in ((0, (0.5, 2.4)), (25, (0.75, 1.35)), (50, (1.0, 0.3)), (80, (1.0, 0.3))):
        ad.set_epoch(epoch)
        w = ad.get_current_weights()
        np.testing.assert_allclose([w['lambda_recon'], w['lambda_base']], want, rtol=1e-6)


def test_base_residual_trainer_resume_roundtrip():
    """state_dict() / load_state_dict() of the fused trainer: a restored trainer + model continues bit-identically."""
    import copy
    from audio_depth_estimation_amd.base_residual_engine import BaseResidualTrainer
    from audio_depth_estimation_amd.models.base_residual_model import BaseResidualDepthNet
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(9)
    x = torch.rand(2, 2, 32, 32, generator=g).to('cuda')
    gt = (30 * torch.rand(2, 1, 32, 32, generator=g)).to('cuda')

    def make():
        m = BaseResidualDepthNet(2, 64, True, 32, 30.0)
        m.compute_dtype = torch.float32
        return m.to('cuda').train()

    ma = make()
    ta = BaseResidualTrainer(ma.engine(), use_silog=True, lr=1e-3)
    for _ in range(2):
        ta.step(x, gt)
    sd_model = copy.deepcopy({k: v.detach().clone() for k, v in ma.state_dict().items()})
    sd_opt = ta.state_dict()
    assert float(sd_opt['state'][0]['step']) == 2 and 'param_groups' in sd_opt      # torch.optim format
    la, _ = ta.step(x, gt)
    la = float(la)
    mb = make()
    mb.load_state_dict(sd_model)
    tb = BaseResidualTrainer(mb.engine(), use_silog=True, lr=1e-3)
    tb.load_state_dict(sd_opt, 'cuda')
    lb, _ = tb.step(x, gt)
    assert float(lb) == la
    for (k, a), (_, b) in zip(ma.state_dict().items(), mb.state_dict().items()):
        assert torch.equal(a, b), k


def test_graph_step_survives_a_ragged_train_step():
    """The trainer's loss / gradient scratch is kept per batch shape (round-2 advisor finding): a ragged trainer.step between
    two replays of the captured step leaves the run bit-identical to the eager one."""
    from audio_depth_estimation_amd.base_residual_engine import BaseResidualTrainer
    from audio_depth_estimation_amd.models.base_residual_model import BaseResidualDepthNet
    g = torch.Generator().manual_seed(10)
    x = torch.rand(4, 2, 32, 32, generator=g).to('cuda')
    gt = (30 * torch.rand(4, 1, 32, 32, generator=g)).to('cuda')
    finals = []
    for mode in ('eager', 'graph'):
        torch.manual_seed(0)
        m = BaseResidualDepthNet(2, 64, True, 32, 30.0)
        m.compute_dtype = torch.bfloat16
        m = m.to('cuda').train()
        tr = BaseResidualTrainer(m.engine(), use_silog=True, lr=1e-3)
        if mode == 'graph':
            tr.enable_graph(after_steps=1)
        for it in range(6):
            tr.step(x, gt)
            if it in (2, 4):
                tr.step(x[:3], gt[:3])
                junk = [torch.full((3, 1, 32, 32), float('nan'), device='cuda') for _ in range(8)]
                del junk
        torch.cuda.synchronize()
        finals.append(m.engine().flat_p.detach().clone())
    assert torch.isfinite(finals[1]).all()
    assert torch.equal(finals[0], finals[1])


def test_mse_reconstruction_variant_matches_reference_formula():
    """BaseResidualLoss(use_l1=False, use_silog=False): the reference's recon term is F.mse_loss over the valid pixels
    (utils_base_residual_loss.py:60-65, 118-131).  Checked against that formula in torch (value and d loss/d final)."""
    from audio_depth_estimation_amd import kernels as K
    from audio_depth_estimation_amd.utils_base_residual_loss import BaseResidualLoss, recon_criterion
    torch.manual_seed(3)
    B, H, W = 2, 48, 40
    gt = torch.rand(B, 1, H, W) * 30
    gt[gt < 3] = 0
    base = gt + torch.randn_like(gt)
    resid = 0.3 * torch.randn_like(gt)
    final = (base + resid).requires_grad_(True)
    valid = gt > 0
    k = 16
    s = F.avg_pool2d(gt, kernel_size=k, stride=1, padding=k // 2)
    struct = F.interpolate(s, size=(H, W), mode='bilinear', align_corners=False)
    rec = F.mse_loss(final[valid], gt[valid])
    ref_total = 0.7 * rec + 1.2 * F.l1_loss(base[valid], struct[valid]) + 0.05 * resid[valid].abs().mean()
    gref, = torch.autograd.grad(0.7 * rec, final)
    crit = BaseResidualLoss(lambda_recon=0.7, lambda_base=1.2, lambda_sparse=0.05, lowpass_kernel=k, use_l1=False,
                            use_silog=False)
    total, d = crit(base.to(DEV), resid.to(DEV), final.detach().to(DEV), gt.to(DEV), valid.to(DEV))
    assert abs(float(total) - float(ref_total)) <= 2e-5 * abs(float(ref_total))
    assert abs(d['recon'] - float(rec)) <= 2e-5 * float(rec)
    # gradient of the weighted reconstruction term, as the fused trainer asks for it
    code, mm = recon_criterion(False, False)
    stats = torch.zeros(4, dtype=torch.float64, device=DEV)
    ws = torch.empty(4096 + 8, dtype=torch.float64, device=DEV)
    lo, gr = torch.zeros(1, device=DEV), torch.empty(B, 1, H, W, device=DEV)
    fd, gd = final.detach().to(DEV), gt.to(DEV)
    K.loss_stats(fd, gd, 1.0, mm, 1e-6, stats, ws)
    K.loss_finish(fd, gd, 1.0, mm, 1e-6, stats, code, 0.7, 0.0, 0.5, lo, gr)
    assert rel_err(gr, gref) <= 1e-5
    # loss-only call (grad = None): the MSE value must still be written (round-2 advisor finding: it was skipped)
    lo2 = torch.full((1,), -1.0, device=DEV)
    K.loss_finish(fd, gd, 1.0, mm, 1e-6, stats, code, 0.7, 0.0, 0.5, lo2, None)
    assert float(lo2) == float(lo) and abs(float(lo2) - 0.7 * float(rec)) <= 2e-5 * 0.7 * float(rec)
    with pytest.raises(RuntimeError, match='criterion 4'):
        K.loss_finish(fd, gd, 1.0, 1, 1e-6, stats, 4, 0.7, 0.0, 0.5, lo, gr)


def test_output_size_resize_inside_the_decoders():
    """Input 32x32 with output_size 64: both heads resize their activated maps (base_residual_model.py:185-211) -- forward
    and every gradient against the oracle (pinned by the reference fixture at equal sizes) with torch autograd."""
    from audio_depth_estimation_amd.base_residual_engine import BaseResidualTrainer
    from audio_depth_estimation_amd.models.base_residual_model import BaseResidualDepthNet
    from oracle import dcnet_oracle as O
    torch.manual_seed(0)
    m = BaseResidualDepthNet(2, 64, True, 64, 30.0)
    m.compute_dtype = torch.float32
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(11)
    x = torch.rand(2, 2, 32, 32, generator=g)
    gt = 30 * torch.rand(2, 1, 64, 64, generator=g)
    gt[gt < 3] = 0
    pk = [k for k, _ in m.named_parameters()]
    sdo = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    for k in pk:
        sdo[k].requires_grad_(True)
    base, res, fin, _ = O.base_residual_forward(sdo, x.double(), 30.0, True, output_size=64)
    assert tuple(fin.shape[-2:]) == (64, 64)
    loss, _ = O.base_residual_loss(base, res, fin, gt.double(), gt > 0, 1.0, 1.2, 0.05, 16, use_silog=True, silog_lambda=0.5)
    loss.backward()
    m = m.to(DEV).train()
    tr = BaseResidualTrainer(m.engine(), 1.0, 1.2, 0.05, 16, use_silog=True, silog_lambda=0.5, lr=1e-3, clip_norm=1.0)
    total, _ = tr.step(x.to(DEV), gt.to(DEV))
    eng = m.engine()
    assert tuple(eng.final.shape[-2:]) == (64, 64)
    assert rel_err(eng.head_base.result, base) <= 1e-4 and rel_err(eng.head_res.result, res) <= 1e-4
    assert rel_err(eng.final, fin) <= 1e-4
    assert abs(float(total) - float(loss)) <= 1e-4 * abs(float(loss))
    named = dict(m.named_parameters())
    for k in pk:
        want = sdo[k].grad
        if float(want.abs().max()) < 1e-9:
            continue
        got = eng.grad_view(named[k]).detach().double().cpu()
        assert float((got - want).norm() / want.norm()) <= 2e-2, k      # (ReLU flips)


@pytest.mark.parametrize('variant', ['silog', 'l1', 'mse'])
def test_reference_style_autograd_loop_matches_the_fused_trainer(variant):
    """train_base_residual.py's loop as written -- base, res, final = model(x); loss, _ = criterion(base, res, final, gt,
    gt > 0); loss.backward(); clip_grad_norm_; optimizer.step() -- on the mirror modules: the gradients that reach
    .grad equal the fused trainer's (f32 compute: <= 1e-5 of each tensor's max), and two steps with torch.optim.AdamW
    stay on the fused trainer's trajectory (same loss at step 2 to 1e-5, mean parameter distance <= 0.02 * lr)."""
    from audio_depth_estimation_amd.base_residual_engine import BaseResidualTrainer
    from audio_depth_estimation_amd.models.base_residual_model import BaseResidualDepthNet
    from audio_depth_estimation_amd.utils_base_residual_loss import BaseResidualLoss
    g = torch.Generator().manual_seed(21)
    x = torch.rand(2, 2, 32, 32, generator=g).to('cuda')
    gt = (30 * torch.rand(2, 1, 32, 32, generator=g)).to('cuda')
    gt[:, :, :4] = 0.0                                   # some invalid pixels
    kw = dict(use_l1=variant == 'l1', use_silog=variant == 'silog')
    lr = 1e-3

    def make():
        torch.manual_seed(3)
        m = BaseResidualDepthNet(2, 64, True, 32, 30.0)
        m.compute_dtype = torch.float32
        return m.to('cuda').train()

    ma, mb = make(), make()
    crit = BaseResidualLoss(1.0, 1.2, 0.05, 16, **kw)
    opt = torch.optim.AdamW(ma.parameters(), lr=lr)
    tr = BaseResidualTrainer.from_criterion(mb.engine(), crit, lr=lr, clip_norm=1.0)
    for it in range(2):
        opt.zero_grad()
        base, res, final = ma(x)
        assert final.requires_grad and base.requires_grad
        loss, parts = crit(base, res, final, gt, gt > 0)
        loss.backward()
        lt, terms = tr.step(x, gt)
        assert abs(float(loss) - float(lt)) <= 1e-5 * abs(float(lt)), (it, float(loss), float(lt))
        if it == 0:
            for (k, p), (_, q) in zip(ma.named_parameters(), mb.named_parameters()):
                gb = mb.engine().grad_view(q)
                assert p.grad is not None, k
                assert float((p.grad - gb).abs().max()) <= 1e-5 * float(gb.abs().max()) + 1e-12, k
        torch.nn.utils.clip_grad_norm_(ma.parameters(), 1.0)
        opt.step()
    # (AdamW moves every element by about lr per step whatever its gradient's size, so elements whose tiny gradient
    #  differs in the last bits may differ by a step; the loss of step 2 above already agreed to 1e-5)
    for (k, p), (_, q) in zip(ma.named_parameters(), mb.named_parameters()):
        d = (p - q).abs()
        assert float(d.max()) <= 4.1 * lr and float(d.mean()) <= 0.02 * lr, (k, float(d.max()), float(d.mean()))
    with torch.no_grad():                                 # no graph outside training / under no_grad
        assert not ma(x)[2].requires_grad
    assert not ma.eval()(x)[2].requires_grad
