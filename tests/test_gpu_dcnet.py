"""GPU parity of the DoubleConv-family networks on libadn (models.rgb_depth_model / dc_engine).

  * against the committed golden vectors produced by the REFERENCE (tests/golden/rgb64_bc8.npz; base_channels=8,
    so the f32 path runs the generic kernels for C%8 != 0 layers and MFMA elsewhere) -- f32 compute, tolerance:
    prediction relative L1 <= 1e-4, gradients <= 2e-3 of the per-tensor max, one AdamW step <= 2 % of lr;
  * against the CPU oracle in float64 at full width (base_channels=64, MFMA kernels).
    f32: prediction relative L1 <= 1e-5; gradients: relative L2 error <= 2e-2 per tensor and cosine >= 0.9999.
    The gradient bound is NOT the kernels' accuracy (1e-6, see the layers behind the last ReLU flip in
    tools/diag_rgb.py) but the price of ONE ReLU whose pre-activation is within rounding of 0 (f64 says -6e-7,
    f32 says +2e-6, measured) and carries a large gradient: everything upstream of it moves by ~1e-3.
    bf16: the oracle applies bf16 rounding at the engine's storage points (dcnet_oracle.QUANT: conv operands,
    stored z, stored activations) with straight-through gradients.  A freshly initialised BatchNorm+ReLU conv
    stack amplifies any perturbation by ~1.7x per conv (measured, and reproduced by the emulation on CPU), so
    after 18 convs even accumulation-order differences reach 1e-2: prediction relative L1 <= 3e-2, gradient
    cosine >= 0.9 (measured 0.946..1.0).  The tight bf16 bounds are the per-kernel tests.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
DEV = 'cuda'


def rel_l1(a, b):
    a, b = a.detach().float().cpu(), torch.as_tensor(b).float()
    return float((a - b).abs().sum() / (b.abs().sum() + 1e-30))


def max_rel(a, b):
    a, b = a.detach().float().cpu(), torch.as_tensor(b).float()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _rgb(bc, S, dtype, sd=None, max_depth=30.0, bilinear=True):
    from audio_depth_estimation_amd.models.rgb_depth_model import RGBDepthNet
    if sd is not None:
        bilinear = 'up1.up.weight' not in sd            # Up(bilinear=False) owns a ConvTranspose2d
    model = RGBDepthNet(base_channels=bc, bilinear=bilinear, output_size=S, max_depth=max_depth)
    model.compute_dtype = dtype
    if sd is not None:
        model.load_state_dict(sd)
    return model.to(DEV)


def _check_sd1(sd_now, z, lr):
    """One AdamW step against the reference.  Adam's update lr * g / (|g| + 1e-8) is ill-conditioned where the
    gradient is at the 1e-8 epsilon scale (a 1e-9 gradient difference moves the step by several % of lr), so
    the 2 %-of-lr bound is applied where |g_ref| > 1e-6 and the trivial bound (one full step) elsewhere."""
    for k in sd_now:
        ref = torch.from_numpy(z['sd1/' + k])
        if ref.dtype == torch.int64:
            assert int(sd_now[k]) == int(ref), k
            continue
        err = (sd_now[k].cpu() - ref).abs()
        if 'grad/' + k in z.files:
            g = torch.from_numpy(z['grad/' + k]).abs()
            tol = torch.where(g > 1e-6, torch.full_like(g, 0.02 * lr), torch.full_like(g, 1.01 * lr))
            assert bool((err <= tol + 1e-6 * ref.abs()).all()), (k, float(err.max()))
        else:
            assert float(err.max()) <= 1e-5 * float(ref.abs().max()) + 1e-6, k


# bilinear=True / ConvTranspose2d upsampling / 32x32 input resized to output_size 64 before the clamp
@pytest.mark.parametrize('fixture', ['rgb64_bc8', 'rgbconvt64_bc8', 'rgbresize32to64_bc8'])
def test_rgb_golden_reference_parity_f32(fixture):
    from audio_depth_estimation_amd.engine import FusedTrainer
    z = np.load(os.path.join(GOLDEN, fixture + '.npz'))
    bc, S, B = [int(v) for v in z['meta']]
    lr, wd, max_depth, l1w, sw = [float(v) for v in z['hyper']]
    sd0 = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('sd0/')}
    model = _rgb(bc, S, torch.float32, sd0, max_depth)
    assert list(model.state_dict().keys()) == list(sd0.keys())
    image, gt = torch.from_numpy(z['image']).to(DEV), torch.from_numpy(z['gt']).to(DEV)

    model.eval()
    with torch.no_grad():
        pe = model(image)
    assert rel_l1(pe, z['pred_eval']) <= 1e-4

    # --- path 1: torch autograd + torch optimizer driving the engine (train_rgb_depth.py:355-362 semantics)
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=wd)
    opt.zero_grad()
    pred, feats = model(image, return_features=True)
    assert rel_l1(pred, z['pred_train']) <= 1e-4
    for k in ('x1', 'x5', 'd4', 'd1'):
        assert rel_l1(feats[k], z['feat/' + k]) <= 1e-4, k
    loss = l1w * (pred - gt).abs().mean() + sw * ((pred[:, :, :, :-1] - pred[:, :, :, 1:]).abs().mean() +
                                                  (pred[:, :, :-1, :] - pred[:, :, 1:, :]).abs().mean())
    assert abs(loss.item() - float(z['loss'])) <= 1e-4 * abs(float(z['loss']))
    loss.backward()
    for k, prm in model.named_parameters():
        assert prm.grad is not None, k
        # (3e-3 for the ConvTranspose2d variant: one near-zero pre-activation flips a ReLU, as in the binaural case)
        assert max_rel(prm.grad, z['grad/' + k]) <= (2e-3 if fixture == 'rgb64_bc8' else 3e-3), k
    opt.step()
    _check_sd1(model.state_dict(), z, lr)

    # --- path 2: fully fused step (DepthLoss + AdamW kernels, no clipping) from the same start point
    model2 = _rgb(bc, S, torch.float32, sd0, max_depth)
    model2.train()
    tr = FusedTrainer(model2.engine(), 'DepthLoss', l1w, sw, optimizer='AdamW', lr=lr, weight_decay=wd,
                      clip_norm=None)
    loss2, pred2 = tr.step(image, gt)
    assert abs(loss2.item() - float(z['loss'])) <= 1e-4 * abs(float(z['loss']))
    assert max_rel(tr.gout, z['pred_grad']) <= 1e-5
    _check_sd1(model2.state_dict(), z, lr)


def _oracle_step(sd, image, gt, max_depth):
    from oracle import dcnet_oracle
    sd = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    pkeys = [k for k, v in sd.items() if v.is_floating_point() and 'running_' not in k]
    for k in pkeys:
        sd[k].requires_grad_(True)
    pred, stats = dcnet_oracle.rgb_forward(sd, image.double(), max_depth, training=True)
    pred.retain_grad()
    loss = dcnet_oracle.depth_loss(pred, gt.double())
    loss.backward()
    return pred.detach(), loss.item(), {k: sd[k].grad for k in pkeys}, stats, pred.grad


@pytest.mark.parametrize('bilinear', [True, False])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_rgb_full_width_against_oracle(dtype, bilinear):
    """base_channels=64 at 64x64, B=2: every conv runs the MFMA S1 kernels (thin first layer included); with
    bilinear=False the four ConvTranspose2d(k 2, s 2) upsamplers run as 1x1 MFMA GEMMs + pixel shuffle."""
    from audio_depth_estimation_amd.engine import FusedTrainer
    torch.manual_seed(0)
    S = 64
    model = _rgb(64, S, dtype, bilinear=bilinear)
    with torch.no_grad():
        model.outc.bias.fill_(2.0)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(1234)
    image = torch.rand(2, 3, S, S, generator=g)
    gt = 30 * torch.rand(2, 1, S, S, generator=g)
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    from oracle import dcnet_oracle
    if dtype == torch.bfloat16:
        dcnet_oracle.QUANT = lambda t: t.float().bfloat16().to(t.dtype)
    try:
        pred_ref, loss_ref, grads_ref, stats_ref, pred_grad_ref = _oracle_step(sd, image, gt, 30.0)
    finally:
        dcnet_oracle.QUANT = None

    model.train()
    eng = model.engine()
    f32 = dtype == torch.float32
    # forward + backward from the ORACLE's d loss / d pred: the L1 + TV loss has a discontinuous gradient
    # (sign of near-equal neighbours), so feeding both sides the same upstream gradient tests the network
    # kernels rather than the sign flips of a 1e-7 prediction difference
    pred = eng.forward(image.to(DEV), True).clone()
    assert rel_l1(pred, pred_ref) <= (1e-5 if f32 else 3e-2)
    eng.backward(pred_grad_ref.float().to(DEV))
    for k, prm in model.named_parameters():
        got = eng.grad_view(prm).detach().float().cpu().reshape(-1)
        ref = grads_ref[k].reshape(-1).float()
        cos = float(torch.dot(got, ref) / (got.norm() * ref.norm() + 1e-30))
        if f32:
            rl2 = float((got - ref).norm() / (ref.norm() + 1e-30))
            assert rl2 <= 2e-2 and cos >= 0.9999, (k, rl2, cos)
        else:
            assert cos >= 0.9, (k, cos)
    for k, v in stats_ref.items():
        got = model.state_dict()[k].cpu()
        v = v.float()
        assert float((got - v).abs().max()) <= (1e-5 if f32 else 3e-2) * float(v.abs().max()) + 1e-6, k
    # the fused loss on the engine's own prediction
    from audio_depth_estimation_amd import kernels as K
    stats = torch.zeros(4, dtype=torch.float64, device=DEV)
    ws = torch.empty(K.l1tv_workspace_bytes(pred.numel()) // 8, dtype=torch.float64, device=DEV)
    lo = torch.zeros(1, dtype=torch.float32, device=DEV)
    gout = torch.empty_like(pred)
    K.l1tv_stats(pred, gt.to(DEV), stats, ws)
    K.l1tv_finish(pred, gt.to(DEV), stats, 1, 1.0, 0.1, lo, gout)
    assert abs(float(lo) - loss_ref) <= (1e-5 if f32 else 3e-2) * abs(loss_ref)


def test_rgb_graph_and_plan_match_eager():
    """hipGraph replay and launch-plan replay of the fused RGB step give the eager parameters."""
    from audio_depth_estimation_amd.engine import FusedTrainer
    g = torch.Generator().manual_seed(3)
    image = torch.rand(2, 3, 64, 64, generator=g).to(DEV)
    gt = (30 * torch.rand(2, 1, 64, 64, generator=g)).to(DEV)
    finals = []
    for mode in ('eager', 'graph', 'plan'):
        torch.manual_seed(0)
        model = _rgb(32, 64, torch.bfloat16)
        model.train()
        tr = FusedTrainer(model.engine(), 'DepthLoss', 1.0, 0.1, optimizer='AdamW', lr=1e-3, weight_decay=0.01,
                          clip_norm=None)
        if mode == 'graph':
            tr.enable_graph(after_steps=2)
        elif mode == 'plan':
            tr.enable_launch_plan(after_steps=2)
        for _ in range(5):
            loss, _ = tr.step(image, gt)
        torch.cuda.synchronize()
        finals.append((float(loss), model.engine().flat_p.detach().clone()))
    for lossv, flat in finals[1:]:
        assert abs(lossv - finals[0][0]) <= 1e-6 * abs(finals[0][0])
        assert torch.equal(flat, finals[0][1])


def test_rgb_error_behaviour():
    model = _rgb(8, 64, torch.float32)
    with pytest.raises(RuntimeError):
        model(torch.rand(1, 3, 64, 64))                  # CPU tensor: no CPU path
    with pytest.raises(RuntimeError):
        model(torch.rand(1, 2, 64, 64, device=DEV))      # wrong channel count
    with pytest.raises(RuntimeError):
        model.inc(torch.rand(1, 3, 64, 64, device=DEV))  # inner blocks are not callable on their own


# ---- BinauralAttentionDepthNet --------------------------------------------------------------------------
def _binaural(bc, S, dtype, sd=None, max_depth=30.0, levels=(2, 3, 4, 5), bilinear=True):
    from audio_depth_estimation_amd.models.binaural_attention_model import BinauralAttentionDepthNet
    model = BinauralAttentionDepthNet(base_channels=bc, bilinear=bilinear, output_size=S, max_depth=max_depth,
                                      attention_levels=list(levels))
    model.compute_dtype = dtype
    if sd is not None:
        model.load_state_dict(sd)
    return model.to(DEV)


def _noise_bias(k):
    """Parameters whose true gradient is identically 0, where the reference holds float noise ~1e-9 (and Adam turns
    that noise into +-lr steps):
      * fusion conv bias: sits in front of BatchNorm (libadn writes the exact zero);
      * key bias: shifts every score of a query by the same amount, which softmax ignores;
      * value / out bias: add a per-channel constant to the attention branch, which passes linearly through the
        1x1 fusion conv and is removed by its BatchNorm.
    For the last three libadn, like the reference, ends up with rounding noise."""
    return ((k.startswith('fusion_layers') and k.endswith('.0.bias')) or
            (k.startswith('attention_modules') and k.endswith(('.key.bias', '.value.bias', '.out.bias'))))


def _check_noise_grad(k, got, scale, rel=1e-4):
    if k.startswith('fusion_layers'):
        assert float(got.abs().max()) == 0.0, k
    else:
        assert float(got.abs().max()) <= rel * scale, (k, float(got.abs().max()), scale)


def _qbias(k):
    return k.rsplit('.', 2)[0] + '.query.bias'


def test_binaural_golden_reference_parity_f32():
    """tests/golden/binaural64_bc8.npz: two encoders, cross-attention at levels 2-5 (gamma = 0.5), fusion, decoder,
    sigmoid head; masked (gt > 0) Combined L1 + SIlog loss; AdamW step (train_binaural_attention.py:399-433)."""
    from audio_depth_estimation_amd.engine import FusedTrainer
    z = np.load(os.path.join(GOLDEN, 'binaural64_bc8.npz'))
    bc, S, B = [int(v) for v in z['meta']]
    lr, wd, max_depth, l1w, sw, lam = [float(v) for v in z['hyper']]
    sd0 = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('sd0/')}
    model = _binaural(bc, S, torch.float32, sd0, max_depth)
    assert list(model.state_dict().keys()) == list(sd0.keys())
    audio, gt = torch.from_numpy(z['audio']).to(DEV), torch.from_numpy(z['gt']).to(DEV)
    model.eval()
    with torch.no_grad():
        pe = model(audio)
    assert rel_l1(pe, z['pred_eval']) <= 1e-4
    model.train()
    tr = FusedTrainer(model.engine(), 'Combined', l1w, sw, lam, max_depth=max_depth, optimizer='AdamW', lr=lr,
                      weight_decay=wd, clip_norm=None, mask_mode='gt0')
    loss, pred = tr.step(audio, gt)
    assert rel_l1(pred, z['pred_train']) <= 1e-4
    assert abs(loss.item() - float(z['loss'])) <= 1e-4 * abs(float(z['loss']))
    assert max_rel(tr.gout, z['pred_grad']) <= 1e-4
    eng = model.engine()
    for k, prm in model.named_parameters():
        if _noise_bias(k):
            _check_noise_grad(k, eng.grad_view(prm),
                              float(np.abs(z['grad/' + _qbias(k)]).max()) if k.startswith('attention') else 1.0)
            continue
        assert max_rel(eng.grad_view(prm), z['grad/' + k]) <= 3e-3, (k, max_rel(eng.grad_view(prm), z['grad/' + k]))
    sd1 = model.state_dict()
    for k in sd1:
        ref = torch.from_numpy(z['sd1/' + k])
        if ref.dtype == torch.int64:
            assert int(sd1[k]) == int(ref), k
        elif _noise_bias(k):
            assert float((sd1[k].cpu() - ref).abs().max()) <= 2.02 * lr
        elif 'grad/' + k in z.files:
            g = torch.from_numpy(z['grad/' + k]).abs()
            # |g| at Adam's epsilon scale: the step direction is noise on both sides (up to 2 lr apart)
            tol = torch.where(g > 1e-6, torch.full_like(g, 0.03 * lr), torch.full_like(g, 2.02 * lr))
            err = (sd1[k].cpu() - ref).abs()
            assert bool((err <= tol + 1e-6 * ref.abs()).all()), (k, float(err.max()))
        else:
            assert float((sd1[k].cpu() - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) + 1e-6, k


@pytest.mark.parametrize('dtype,bilinear', [(torch.float32, True), (torch.bfloat16, True), (torch.float32, False)])
def test_binaural_full_width_against_oracle(dtype, bilinear):
    """(bilinear=False: ConvTranspose2d upsampling and a 1024-channel level 5 -- generic attention kernels.)
    base_channels=64 at 64x64, B=2 (attention over 1024 / 256 / 64 / 16 tokens): MFMA GEMMs + attention kernels
    against the float64 oracle.  f32: as for RGBDepthNet.  bf16: the oracle emulates bf16 storage in the conv
    stacks; this net is ~30 conv/attention stages deep and a freshly initialised BN+ReLU stack amplifies
    perturbations ~1.7x per stage (see the RGB test), so the end-to-end bf16 bounds can only be loose: prediction
    relative L1 <= 5e-2 (measured 2.6e-2), gradient cosine >= 0.6 (measured 0.65..1.0).  The f32 path runs the
    very same templated kernels at 1e-6, and the bf16 kernels are individually bounded in
    test_gpu_dcnet_kernels.py."""
    from oracle import dcnet_oracle, loss_oracle
    torch.manual_seed(0)
    S = 64
    model = _binaural(64, S, dtype, bilinear=bilinear)
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for m in model.attention_modules.values():
            m.gamma.fill_(0.5)
            for conv in (m.query, m.key, m.value, m.out):
                conv.bias.copy_(0.1 * torch.randn(conv.bias.shape, generator=g))
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    audio = torch.rand(2, 2, S, S, generator=g)
    gt = 30 * torch.rand(2, 1, S, S, generator=g)
    gt[gt < 3] = 0
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    pkeys = [k for k, v in sd64.items() if v.is_floating_point() and 'running_' not in k]
    for k in pkeys:
        sd64[k].requires_grad_(True)
    if dtype == torch.bfloat16:
        dcnet_oracle.QUANT = lambda t: t.float().bfloat16().to(t.dtype)
    try:
        pred_ref, stats_ref = dcnet_oracle.binaural_forward(sd64, audio.double(), 30.0, training=True)
    finally:
        dcnet_oracle.QUANT = None
    pred_ref.retain_grad()
    loss_ref = loss_oracle.masked_loss(pred_ref, gt.double(), 'L1', mask_mode='gt0')
    loss_ref.backward()
    model.train()
    eng = model.engine()
    f32 = dtype == torch.float32
    pred = eng.forward(audio.to(DEV), True).clone()
    assert rel_l1(pred, pred_ref.detach()) <= (1e-5 if f32 else 5e-2)
    eng.backward(pred_ref.grad.float().to(DEV))
    for k, prm in model.named_parameters():
        got = eng.grad_view(prm).detach().float().cpu().reshape(-1)
        ref = sd64[k].grad.reshape(-1).float()
        if _noise_bias(k):
            _check_noise_grad(k, got, float(sd64[_qbias(k)].grad.abs().max()) if k.startswith('attention') else 1.0,
                              1e-4 if f32 else 5e-2)      # bf16: rounding noise of the summed dk / dv / G rows
            continue
        cos = float(torch.dot(got, ref) / (got.norm() * ref.norm() + 1e-30))
        if f32:
            rl2 = float((got - ref).norm() / (ref.norm() + 1e-30))
            # (gamma is ONE number, a sum over every pixel with heavy cancellation: 5e-2)
            assert rl2 <= (5e-2 if got.numel() == 1 else 2e-2) and cos >= 0.9999, (k, rl2, cos)
        elif got.numel() == 1:
            # gamma: ONE number that is a sum over every pixel with heavy cancellation -- in bf16 its value (even its
            # sign) is dominated by the amplified rounding noise of this freshly initialised net (generic and MFMA
            # attention kernels disagree with the oracle and with each other by factors of 2-5 here; exact in f32)
            assert bool(torch.isfinite(got).all()), k
        else:
            assert cos >= 0.6, (k, cos)


def test_binaural_final_resize_against_oracle():
    """32x32 input, output_size 64: sigmoid * max_depth -> F.interpolate(bilinear, align_corners=False) -> clamp
    (binaural_attention_model.py:322-337), forward and every parameter gradient against the float64 oracle."""
    from oracle import dcnet_oracle, loss_oracle
    torch.manual_seed(0)
    model = _binaural(16, 64, torch.float32, levels=(4, 5))
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for m in model.attention_modules.values():
            m.gamma.fill_(0.5)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    audio = torch.rand(2, 2, 32, 32, generator=g)
    gt = 30 * torch.rand(2, 1, 64, 64, generator=g)
    gt[gt < 3] = 0
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    pkeys = [k for k, v in sd64.items() if v.is_floating_point() and 'running_' not in k]
    for k in pkeys:
        sd64[k].requires_grad_(True)
    pred_ref, _ = dcnet_oracle.binaural_forward(sd64, audio.double(), 30.0, attention_levels=(4, 5), training=True,
                                                output_size=64)
    assert pred_ref.shape == (2, 1, 64, 64)
    pred_ref.retain_grad()
    loss_oracle.masked_loss(pred_ref, gt.double(), 'L1', mask_mode='gt0').backward()
    model.train()
    eng = model.engine()
    pred = eng.forward(audio.to(DEV), True).clone()
    assert pred.shape == (2, 1, 64, 64)
    assert rel_l1(pred, pred_ref.detach()) <= 1e-5
    eng.backward(pred_ref.grad.float().to(DEV))
    for k, prm in model.named_parameters():
        if _noise_bias(k):
            continue
        got = eng.grad_view(prm).detach().float().cpu().reshape(-1)
        ref = sd64[k].grad.reshape(-1).float()
        cos = float(torch.dot(got, ref) / (got.norm() * ref.norm() + 1e-30))
        rl2 = float((got - ref).norm() / (ref.norm() + 1e-30))
        assert rl2 <= (5e-2 if got.numel() == 1 else 2e-2) and cos >= 0.9999, (k, rl2, cos)


def test_resume_from_a_checkpoint_written_by_the_reference():
    """tests/golden/ref_ckpt_binaural_bc4.pth was written by the reference's own code path (model.state_dict() +
    torch.optim.AdamW.state_dict() in the dict layout of train_binaural_attention.py:563-571 after two of its steps).
    The mirror loads it the way the reference resumes (:358-361) -- model.load_state_dict + the fused trainer's
    load_state_dict on the torch-format optimizer state -- and its next step must match the reference's third step."""
    import numpy as np
    from audio_depth_estimation_amd.engine import FusedTrainer
    from audio_depth_estimation_amd.models.binaural_attention_model import create_binaural_attention_model
    ck = torch.load(os.path.join(GOLDEN, 'ref_ckpt_binaural_bc4.pth'), map_location='cpu')
    z = np.load(os.path.join(GOLDEN, 'ref_ckpt_binaural_bc4_next.npz'))
    assert set(ck) >= {'epoch', 'model_state_dict', 'optimizer_state_dict'} and ck['epoch'] == 2
    lr, wd = [float(v) for v in z['hyper']]
    torch.manual_seed(123)
    model = create_binaural_attention_model(base_channels=4, bilinear=True, output_size=64, max_depth=30.0,
                                            attention_levels=[2, 3, 4, 5])
    model.compute_dtype = torch.float32
    model.load_state_dict(ck['model_state_dict'])
    model = model.to(DEV).train()
    tr = FusedTrainer(model.engine(), 'L1', optimizer='AdamW', lr=lr, weight_decay=wd, clip_norm=None, mask_mode='gt0')
    tr.load_state_dict(ck['optimizer_state_dict'], DEV)
    assert int(tr.state[0].item()) == 2
    before = {k: p.detach().cpu().clone() for k, p in model.named_parameters()}
    loss, _ = tr.step(torch.from_numpy(z['audio']).to(DEV), torch.from_numpy(z['gt']).to(DEV))
    assert abs(float(loss) - float(z['loss'])) <= 1e-4 * abs(float(z['loss']))
    worst = ('', 0.0)
    for k, p in model.named_parameters():
        d = (p.detach().cpu() - before[k]).reshape(-1)[:256].numpy()
        ref = z['delta/' + k]
        # an AdamW step with restored moments: per-element change <= lr; compare in units of lr
        err = float(np.abs(d - ref).max()) / lr
        if _noise_bias(k):                     # zero-gradient parameters: Adam steps on float noise on both sides
            assert err <= 2.02, (k, err)
        elif err > worst[1]:
            worst = (k, err)
    assert worst[1] <= 0.05, worst
    # and the state written back is again something the reference's torch optimizer loads
    sd = tr.state_dict()
    probe = torch.optim.AdamW([torch.nn.Parameter(torch.zeros_like(p, device='cpu')) for p in model.parameters()], lr=lr,
                              weight_decay=wd)
    probe.load_state_dict(sd)
    assert float(sd['state'][0]['step']) == 3


# ---- full width, 256 x 256, against numbers produced by the reference (tests/golden/make_golden_fullsize_dc.py) ------------
def _hash_key(key):
    h = 0
    for ch in key:
        h = (h * 131 + ord(ch)) % (2 ** 31 - 1)
    return h


def _perturb_by_name(model):
    """The generator's `perturb`: BatchNorm affine, attention gate and attention biases as a function of the tensor NAME."""
    with torch.no_grad():
        for k, v in model.state_dict().items():
            g = torch.Generator().manual_seed(_hash_key(k))
            if k.endswith('.gamma'):
                v.fill_(0.5)
            elif 'attention_modules' in k and k.endswith('.bias'):
                v.copy_(0.1 * torch.randn(v.shape, generator=g))
            elif v.dim() == 1 and k.endswith('.weight'):
                v.copy_(1.0 + 0.2 * torch.randn(v.shape, generator=g))
            elif v.dim() == 1 and k.endswith('.bias') and ('double_conv' in k or 'fusion' in k):
                v.copy_(0.1 * torch.randn(v.shape, generator=g))


def _fullsize_check(z, model, tr, loss, pred, f32, pred_tol, grad_tol, skip=lambda k: False):
    idx = torch.from_numpy(z['idx'])
    ref = torch.from_numpy(z['pred_val'])
    relp = float((pred.reshape(-1).float().cpu()[idx] - ref).abs().sum() / ref.abs().sum())
    lrel = abs(loss.item() - float(z['loss'])) / abs(float(z['loss']))
    dref = torch.from_numpy(z['pred_grad_val'])
    reld = float((tr.gout.reshape(-1).float().cpu()[idx] - dref).abs().sum() / dref.abs().sum())
    eng, worst = model.engine(), {}
    for k, prm in model.named_parameters():
        gn_ref = float(z['gnorm/' + k])
        if skip(k) or gn_ref < 1e-12:
            continue
        gflat = eng.grad_view(prm).detach().float().cpu().reshape(-1)
        g = torch.Generator().manual_seed(_hash_key(k))
        si = torch.randint(0, gflat.numel(), (min(512, gflat.numel()),), generator=g)
        rms = gn_ref / (gflat.numel() ** 0.5)
        err = float((gflat[si] - torch.from_numpy(z['gsample/' + k])).norm() / (len(si) ** 0.5)) / rms
        worst[k] = (err, abs(float(gflat.double().norm()) - gn_ref) / gn_ref)
    top = sorted(worst.items(), key=lambda kv: -kv[1][0])[:4]
    print(f'pred rel-L1 {relp:.3e} loss rel {lrel:.3e} dloss/dpred rel-L1 {reld:.3e}; worst gradients {top}')
    assert relp <= pred_tol, relp
    assert lrel <= (1e-5 if f32 else 2e-3), lrel
    assert reld <= (1e-4 if f32 else 8e-2), reld
    for k, (err, nerr) in worst.items():
        scalar = 8.0 if k.endswith('.gamma') else 1.0      # one scalar = a 4 M-term sum of mixed signs (measured 1.5 % in f32)
        assert err <= scalar * grad_tol[0] and nerr <= scalar * grad_tol[1], (k, err, nerr)


def _seed_weights_match(model, z):
    for k, v in model.state_dict().items():
        if v.dtype.is_floating_point:
            assert abs(float(v.double().sum()) - float(z['init_sum/' + k])) <= 1e-6 * max(1.0, float(z['init_abs/' + k])), k


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_rgb_full_width_256_against_the_reference(dtype):
    """RGBDepthNet base 64 at 256 x 256, B = 2 (rgb_bc64_256.npz: the reference's create_rgb_depth_model + DepthLoss run on
    the CPU): the patch-staged 3 x 3 MFMA kernels at their real tilings against reference NUMBERS.
    f32: prediction rel-L1 <= 1e-4 (measured 6e-6), every gradient tensor sampled rel-L2 <= 2e-2 (1.0e-2) and norm within
    5e-3 (1.6e-3).  bf16 (18 conv stages in a freshly initialised BatchNorm stack amplify every rounding ~1.7x per stage,
    DESIGN section 2): prediction <= 8e-2 (4.7e-2), d loss / d pred <= 8e-2 (3.4e-2), gradient norms within 15 % (<= 6.7 %)
    -- the per-ELEMENT gradient error of the deep tensors is of the order of the gradient itself (sampled rel-L2 0.69, bound
    1.0): what bf16 keeps at this depth and width is the norm and the training behaviour (test_gpu_config5's descent tests),
    the tight statement about the kernels is the f32 row."""
    from audio_depth_estimation_amd.engine import FusedTrainer
    z = np.load(os.path.join(GOLDEN, 'rgb_bc64_256.npz'))
    bc, S, B = [int(v) for v in z['meta']]
    max_depth, l1w, sw = [float(v) for v in z['hyper']]
    torch.manual_seed(0)
    model = _rgb(bc, S, dtype, None, max_depth)
    _seed_weights_match(model, z)
    _perturb_by_name(model)
    with torch.no_grad():
        model.outc.bias.fill_(2.0)
    g = torch.Generator().manual_seed(1234)
    image = torch.rand(B, 3, S, S, generator=g)
    gt = max_depth * torch.rand(B, 1, S, S, generator=g)
    gt[gt < 0.1 * max_depth] = 0.0
    model.train()
    tr = FusedTrainer(model.engine(), 'DepthLoss', l1w, sw, 0.0, max_depth=max_depth, optimizer='AdamW', lr=1e-6,
                      clip_norm=None)
    loss, pred = tr.step(image.to(DEV), gt.to(DEV))
    f32 = dtype == torch.float32
    _fullsize_check(z, model, tr, loss, pred, f32, 1e-4 if f32 else 8e-2, (2e-2, 5e-3) if f32 else (1.0, 0.15))


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_binaural_level2_attention_16384_tokens_against_the_reference(dtype):
    """BinauralAttentionDepthNet base 64 at 256 x 256 with the LEVEL-2 cross-attention (C = 128, 128 x 128 = 16 384 tokens per
    direction, gamma = 0.5), B = 1, against binaural_l2_bc64_256.npz -- numbers from the reference's own module, which
    materialises the two 16 384 x 16 384 score matrices (models/binaural_attention_model.py:106-153).  Replaces "constant V /
    linear in dO" at this size with "equal to the reference" (VERDICT r2 item 2b): prediction, loss, d loss / d pred and every
    gradient (query / key / value / out / gamma of the attention module included).  f32 (generic exact attention kernels):
    prediction rel-L1 <= 1e-4 (measured 1.6e-6), gradients sampled rel-L2 <= 2e-2 (<= 1e-2; the scalar gamma 1.5 %, bound
    8x).  bf16 (the MFMA attention kernels): prediction <= 4e-2 (1.5e-2), d loss / d pred <= 8e-2 (2.2e-2), gradient norms
    within 20 % (<= 13 %), per-element error of the deepest (first-layer) tensors of the order of the gradient (0.86, bound
    1.0) -- see the RGB test above for what that bound does and does not say."""
    from audio_depth_estimation_amd.engine import FusedTrainer
    z = np.load(os.path.join(GOLDEN, 'binaural_l2_bc64_256.npz'))
    bc, S, B = [int(v) for v in z['meta']]
    max_depth, l1w, sw, lam = [float(v) for v in z['hyper']]
    torch.manual_seed(0)
    model = _binaural(bc, S, dtype, None, max_depth, levels=(2,))
    _seed_weights_match(model, z)
    _perturb_by_name(model)
    with torch.no_grad():
        model.outc[0].weight.mul_(0.1)          # keeps the predictions away from SIlog's 1 / pred singularity (generator script)
    g = torch.Generator().manual_seed(1234)
    audio = torch.rand(B, 2, S, S, generator=g)
    gt = max_depth * torch.rand(B, 1, S, S, generator=g)
    gt[gt < 0.1 * max_depth] = 0.0
    model.train()
    tr = FusedTrainer(model.engine(), 'Combined', l1w, sw, lam, max_depth=max_depth, optimizer='AdamW', lr=1e-6,
                      clip_norm=None, mask_mode='gt0')
    loss, pred = tr.step(audio.to(DEV), gt.to(DEV))
    f32 = dtype == torch.float32
    _fullsize_check(z, model, tr, loss, pred, f32, 1e-4 if f32 else 4e-2, (2e-2, 5e-3) if f32 else (1.0, 0.2),
                    skip=_noise_bias)
