"""Pin oracle/ against the golden vectors generated from the reference (CPU only)."""
import os

import numpy as np
import pytest
import torch

from oracle import loss_oracle, metrics_oracle, optim_oracle, unet_oracle

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def _load(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def _sd(z, prefix):
    return {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}


@pytest.mark.parametrize('name,netG', [('unet256_ngf4', 'unet_256'), ('unet128_ngf4_dn', 'unet_128')])
def test_unet_forward_backward_matches_reference(name, netG):
    z = _load(name)
    ngf, S, depth_norm, B = [int(v) for v in z['meta']]
    lr, max_depth, l1w, sw, lam = [float(v) for v in z['hyper']]
    nd = unet_oracle.num_downs_of(netG)
    sd = _sd(z, 'sd0/')
    audio, gt = torch.from_numpy(z['audio']), torch.from_numpy(z['gt'])

    # eval-mode forward
    with torch.no_grad():
        pe, _ = unet_oracle.unet_forward(sd, audio, nd, bool(depth_norm), training=False)
    np.testing.assert_allclose(pe.numpy(), z['pred_eval'], rtol=1e-5, atol=1e-6)

    # train-mode forward + loss + backward
    pkeys = unet_oracle.param_keys(nd)
    assert pkeys == [k[len('grad/'):] for k in z.files if k.startswith('grad/')]
    for k in pkeys:
        sd[k] = sd[k].clone().requires_grad_(True)
    pred, new_stats = unet_oracle.unet_forward(sd, audio, nd, bool(depth_norm), training=True)
    np.testing.assert_allclose(pred.detach().numpy(), z['pred_train'], rtol=1e-5, atol=1e-6)
    scale = max_depth if depth_norm else 1.0
    loss = loss_oracle.masked_loss(pred, gt, 'Combined', l1w, sw, lam, scale=scale)
    assert abs(loss.item() - float(z['loss'])) <= 1e-5 * abs(float(z['loss']))
    loss.backward()
    for k in pkeys:
        ref = z['grad/' + k]
        got = sd[k].grad.numpy()
        np.testing.assert_allclose(got, ref, rtol=2e-3, atol=1e-6 + 1e-4 * np.abs(ref).max())
    for k, v in new_stats.items():
        np.testing.assert_allclose(v.numpy(), z['sd1/' + k], rtol=1e-5, atol=1e-6)

    # clip + AdamW
    grads = [sd[k].grad.numpy() for k in pkeys]
    total, coef = optim_oracle.clip_coef(grads, 1.0)
    assert abs(total - float(z['grad_norm'])) <= 1e-4 * float(z['grad_norm'])
    for k in pkeys:
        p1, _, _ = optim_oracle.adamw_step(sd[k].detach().numpy(), sd[k].grad.numpy(),
                                           np.zeros(sd[k].shape), np.zeros(sd[k].shape), 1, lr,
                                           grad_scale=coef)
        np.testing.assert_allclose(p1, z['sd1/' + k], rtol=1e-4, atol=2e-6)


def test_param_key_layout_unet256():
    keys = unet_oracle.param_keys(8)
    assert len(keys) == 43
    assert keys[0] == 'model.model.0.weight'
    assert keys[-2:] == ['model.model.3.weight', 'model.model.3.bias']
    chans = unet_oracle.level_channels(8, 64, 2, 1)
    assert chans[0] == (2, 64, 128, 1) and chans[7] == (512, 512, 512, 512) and chans[3] == (256, 512, 1024, 256)


def test_conv_definitions_pin_torch_ops():
    rng = np.random.default_rng(0)
    x = rng.normal(size=(2, 3, 8, 8)).astype(np.float32)
    w = rng.normal(size=(5, 3, 4, 4)).astype(np.float32)
    y = torch.nn.functional.conv2d(torch.from_numpy(x), torch.from_numpy(w), stride=2, padding=1).numpy()
    np.testing.assert_allclose(y, unet_oracle.conv2d_direct_numpy(x, w, 2, 1), rtol=1e-4, atol=1e-4)
    wt = rng.normal(size=(3, 5, 4, 4)).astype(np.float32)
    yt = torch.nn.functional.conv_transpose2d(torch.from_numpy(x), torch.from_numpy(wt), stride=2, padding=1).numpy()
    np.testing.assert_allclose(yt, unet_oracle.conv_transpose2d_direct_numpy(x, wt, 2, 1), rtol=1e-4, atol=1e-4)


def test_loss_cases():
    z = _load('loss_cases')
    pred, gt = z['pred'], z['gt']
    for lam in (0.5, 0.869, 1.0):
        v = loss_oracle.masked_loss(torch.from_numpy(pred), torch.from_numpy(gt), 'SIlog', silog_lambda=lam)
        assert abs(v.item() - float(z[f'silog_{lam}'])) < 1e-5
        g = loss_oracle.masked_loss_grad_numpy(pred, gt, 'SIlog', silog_lambda=lam)
        np.testing.assert_allclose(g, z[f'silog_grad_{lam}'], rtol=1e-3, atol=1e-8)
    g = loss_oracle.masked_loss_grad_numpy(pred, gt, 'Combined', 0.237, 0.637, 0.869)
    np.testing.assert_allclose(g, z['combined_grad'], rtol=1e-3, atol=1e-8)
    v = loss_oracle.masked_loss(torch.from_numpy(pred), torch.from_numpy(gt), 'Combined', 0.237, 0.637, 0.869)
    assert abs(v.item() - float(z['combined'])) < 1e-5


def test_metrics_cases():
    z = _load('metrics_cases')
    names = sorted({k.split('/')[0] for k in z.files})
    assert 'pred_all_negative' in names and 'batched' in names
    for n in names:
        got = metrics_oracle.compute_errors(z[n + '/gt'], z[n + '/pred'])
        np.testing.assert_allclose(np.array(got, dtype=np.float64), z[n + '/ref'], rtol=1e-6, atol=1e-7,
                                   err_msg=n)


@pytest.mark.parametrize('opt', ['AdamW', 'Adam', 'SGD', 'Adam_wd'])
def test_optim_cases(opt):
    z = _load('optim_cases')
    ps = [z[f'{opt}/p0/{i}'].astype(np.float64) for i in range(3)]
    ms = [np.zeros_like(p) for p in ps]
    vs = [np.zeros_like(p) for p in ps]
    for step in range(3):
        gs = [z[f'{opt}/g{step}/{i}'] for i in range(3)]
        total, coef = optim_oracle.clip_coef(gs, 1.0)
        assert abs(total - float(z[f'{opt}/norm{step}'])) < 1e-4 * total
        for i in range(3):
            if opt == 'SGD':
                ps[i] = optim_oracle.sgd_step(ps[i], gs[i], 0.002, coef)
            else:
                lr = 0.001 if opt == 'Adam_wd' else 0.002
                wd = {'AdamW': 0.01, 'Adam': 0.0, 'Adam_wd': 0.01}[opt]
                ps[i], ms[i], vs[i] = optim_oracle.adamw_step(ps[i], gs[i], ms[i], vs[i], step + 1, lr,
                                                               weight_decay=wd, decoupled=(opt == 'AdamW'),
                                                               grad_scale=coef)
            np.testing.assert_allclose(ps[i], z[f'{opt}/p{step + 1}/{i}'], rtol=2e-5, atol=2e-6)


# ---- DoubleConv family (RGBDepthNet, BinauralAttentionDepthNet) -----------------------------------------
def _dc_param_keys(z):
    return [k[len('grad/'):] for k in z.files if k.startswith('grad/')]


# bilinear=True / ConvTranspose2d upsampling / 32x32 input resized to output_size 64 before the clamp
@pytest.mark.parametrize('fixture', ['rgb64_bc8', 'rgbconvt64_bc8', 'rgbresize32to64_bc8'])
def test_rgbdepthnet_oracle_matches_reference(fixture):
    """oracle.dcnet_oracle.rgb_forward + depth_loss vs reference RGBDepthNet / DepthLoss / AdamW step."""
    from oracle import dcnet_oracle
    z = _load(fixture)
    lr, wd, max_depth, l1w, sw = [float(v) for v in z['hyper']]
    S = int(z['meta'][1])
    sd = _sd(z, 'sd0/')
    image, gt = torch.from_numpy(z['image']), torch.from_numpy(z['gt'])
    with torch.no_grad():
        pe, _ = dcnet_oracle.rgb_forward(sd, image, max_depth, training=False, output_size=S)
    np.testing.assert_allclose(pe.numpy(), z['pred_eval'], rtol=1e-5, atol=1e-5)
    pkeys = _dc_param_keys(z)
    for k in pkeys:
        sd[k] = sd[k].clone().requires_grad_(True)
    pred, new_stats, feats = dcnet_oracle.rgb_forward(sd, image, max_depth, training=True, return_features=True,
                                                      output_size=S)
    np.testing.assert_allclose(pred.detach().numpy(), z['pred_train'], rtol=1e-5, atol=1e-5)
    for k in ('x1', 'x5', 'd4', 'd1'):
        np.testing.assert_allclose(feats[k].detach().numpy(), z['feat/' + k], rtol=1e-5, atol=1e-5)
    pred.retain_grad()
    loss = dcnet_oracle.depth_loss(pred, gt, l1w, sw)
    assert abs(loss.item() - float(z['loss'])) <= 1e-6 * abs(float(z['loss']))
    loss.backward()
    np.testing.assert_allclose(pred.grad.numpy(), z['pred_grad'], rtol=1e-6, atol=1e-9)
    for k in pkeys:
        ref = z['grad/' + k]
        np.testing.assert_allclose(sd[k].grad.numpy(), ref, rtol=2e-3, atol=1e-6 + 1e-4 * np.abs(ref).max())
    for k, v in new_stats.items():
        np.testing.assert_allclose(v.numpy(), z['sd1/' + k], rtol=1e-5, atol=1e-6)
    for k in pkeys:                                             # AdamW(lr, weight_decay), no clipping
        p1, _, _ = optim_oracle.adamw_step(sd[k].detach().numpy(), sd[k].grad.numpy(), np.zeros(sd[k].shape),
                                           np.zeros(sd[k].shape), 1, lr, weight_decay=wd)
        np.testing.assert_allclose(p1, z['sd1/' + k], rtol=1e-4, atol=2e-6)


def test_binaural_oracle_matches_reference():
    """oracle.dcnet_oracle.binaural_forward (two encoders, cross-attention, fusion, decoder) vs the reference."""
    from oracle import dcnet_oracle
    z = _load('binaural64_bc8')
    lr, wd, max_depth, l1w, sw, lam = [float(v) for v in z['hyper']]
    sd = _sd(z, 'sd0/')
    audio, gt = torch.from_numpy(z['audio']), torch.from_numpy(z['gt'])
    with torch.no_grad():
        pe, _ = dcnet_oracle.binaural_forward(sd, audio, max_depth, training=False)
    np.testing.assert_allclose(pe.numpy(), z['pred_eval'], rtol=1e-5, atol=1e-5)
    pkeys = _dc_param_keys(z)
    for k in pkeys:
        sd[k] = sd[k].clone().requires_grad_(True)
    pred, new_stats = dcnet_oracle.binaural_forward(sd, audio, max_depth, training=True)
    np.testing.assert_allclose(pred.detach().numpy(), z['pred_train'], rtol=1e-5, atol=1e-5)
    loss = loss_oracle.masked_loss(pred, gt, 'Combined', l1w, sw, lam, mask_mode='gt0')
    assert abs(loss.item() - float(z['loss'])) <= 1e-5 * abs(float(z['loss']))
    loss.backward()
    for k in pkeys:
        ref = z['grad/' + k]
        if k.startswith('fusion_layers') and k.endswith('.0.bias'):
            continue        # a bias in front of BatchNorm: the true gradient is 0, the reference holds float noise
        np.testing.assert_allclose(sd[k].grad.numpy(), ref, rtol=5e-3, atol=1e-6 + 2e-4 * np.abs(ref).max())
    for k, v in new_stats.items():
        np.testing.assert_allclose(v.numpy(), z['sd1/' + k], rtol=1e-5, atol=1e-6)


# ---- AdaBins distillation model ---------------------------------------------------------------------------------
def _sample(t, n=512):
    f = t.detach().reshape(-1)
    if f.numel() <= n:
        return f.clone().numpy()
    return f[torch.linspace(0, f.numel() - 1, n).long()].clone().numpy()


def adabins_initial_state(z):
    """Rebuild the fixture's initial weights: same seed + same construction order as the reference (checked against the
    stored samples), then the perturbations of make_golden_dcnet.py (perturb_bn(3), perturb_biases(9), dropout p=0)."""
    from audio_depth_estimation_amd.models.adabins_distillation_model import create_adabins_distillation_model
    bc, nb, S, B = [int(v) for v in z['meta']]
    torch.manual_seed(0)
    model = create_adabins_distillation_model(n_bins=nb, base_channels=bc, output_size=S, max_depth=float(z['hyper'][1]))
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(1.0 + 0.2 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
                m.running_mean.copy_(0.1 * torch.randn(m.running_mean.shape, generator=g))
                m.running_var.copy_(1.0 + 0.5 * torch.rand(m.running_var.shape, generator=g))
        g = torch.Generator().manual_seed(9)
        for m in model.modules():
            if isinstance(m, (torch.nn.Linear, torch.nn.Conv2d)) and m.bias is not None:
                m.bias.copy_(0.2 * torch.randn(m.bias.shape, generator=g))
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    sd = model.state_dict()
    assert len(sd) == 230
    for k, v in sd.items():
        ref = z['sd0s/' + k]
        np.testing.assert_array_equal(_sample(v) if v.is_floating_point() else v.numpy(), ref, err_msg=k)
    return model


def test_adabins_oracle_matches_reference():
    """oracle.dcnet_oracle.adabins_forward + distillation_loss vs AdaBinsDistillationModel / DistillationLoss (one
    decoder evaluation standing for the reference's two, double BN running-stat update)."""
    from oracle import dcnet_oracle
    z = _load('adabins32_bc64')
    lr, max_depth, lt, lr_, lf, lb, ls, temp = [float(v) for v in z['hyper']]
    model = adabins_initial_state(z)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    audio, rgb, gt = [torch.from_numpy(z[k]) for k in ('audio', 'rgb', 'gt')]
    torch.set_num_threads(8)
    with torch.no_grad():
        o, _ = dcnet_oracle.adabins_forward(sd, audio, None, max_depth, training=False)
    np.testing.assert_allclose(o['audio']['final_depth'].numpy(), z['eval/final_depth'], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(o['audio']['bin_centers'].numpy(), z['eval/bin_centers'], rtol=1e-4, atol=1e-4)
    pkeys = [k[len('gnorm/'):] for k in z.files if k.startswith('gnorm/')]
    assert all(k.startswith(('audio_', 'residual_head')) for k in pkeys)         # the teacher gets no gradients
    for k in pkeys:
        sd[k] = sd[k].clone().requires_grad_(True)
    o, new_stats = dcnet_oracle.adabins_forward(sd, audio, rgb, max_depth, training=True)
    for side in ('audio', 'rgb'):
        for k in ('bin_centers', 'bin_widths', 'base_depth', 'residual', 'final_depth'):
            np.testing.assert_allclose(o[side][k].detach().numpy(), z[f'train/{side}/{k}'], rtol=2e-4, atol=2e-4,
                                       err_msg=f'{side}/{k}')
        np.testing.assert_allclose(o[side]['bin_logits'].detach().mean((2, 3)).numpy(), z[f'train/{side}/logits_mean'],
                                   rtol=2e-4, atol=2e-4)
    loss, parts = dcnet_oracle.distillation_loss(o, gt, gt > 0, lt, lr_, lf, lb, ls, temp)
    got = np.array([float(parts[k]) for k in ('task', 'response', 'feature', 'bin', 'bin_centers', 'sparse')])
    np.testing.assert_allclose(got, z['loss_parts'], rtol=2e-4, atol=1e-6)
    assert abs(loss.item() - float(z['loss'])) <= 2e-4 * abs(float(z['loss']))
    loss.backward()
    for k in pkeys:
        g = sd[k].grad
        assert abs(float(g.double().norm()) - float(z['gnorm/' + k])) <= 5e-3 * float(z['gnorm/' + k]) + 1e-7, k
        ref = z['gs/' + k]
        np.testing.assert_allclose(_sample(g), ref, rtol=5e-3, atol=1e-6 + 5e-3 * np.abs(ref).max(), err_msg=k)
    for k, v in new_stats.items():           # decoder BN statistics carry the double update
        ref = z['sd1s/' + k]
        np.testing.assert_allclose(_sample(v), ref, rtol=1e-4, atol=1e-5, err_msg=k)


# ---- Base + Residual model ------------------------------------------------------------------------------------------
def base_residual_initial_state(z):
    from audio_depth_estimation_amd.models.base_residual_model import create_base_residual_model
    bc, S, B = [int(v) for v in z['meta']]
    torch.manual_seed(0)
    model = create_base_residual_model(input_channels=2, base_channels=bc, output_size=S, max_depth=float(z['hyper'][1]))
    g = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(1.0 + 0.2 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
                m.running_mean.copy_(0.1 * torch.randn(m.running_mean.shape, generator=g))
                m.running_var.copy_(1.0 + 0.5 * torch.rand(m.running_var.shape, generator=g))
    for k, v in model.state_dict().items():
        np.testing.assert_array_equal(_sample(v) if v.is_floating_point() else v.numpy(), z['sd0s/' + k], err_msg=k)
    return model


def test_base_residual_oracle_matches_reference():
    from oracle import dcnet_oracle
    z = _load('baseres32_bc64')
    lr, max_depth, lrec, lbase, lsp, k, slam = [float(v) for v in z['hyper']]
    model = base_residual_initial_state(z)
    sd = {kk: v.detach().clone() for kk, v in model.state_dict().items()}
    audio, gt = torch.from_numpy(z['audio']), torch.from_numpy(z['gt'])
    torch.set_num_threads(8)
    np.testing.assert_allclose(dcnet_oracle.lowpass_struct(gt, int(k)).numpy(), z['struct'], rtol=1e-6, atol=1e-6)
    with torch.no_grad():
        b, r, f, _ = dcnet_oracle.base_residual_forward(sd, audio, max_depth, training=False)
    for t, name in ((b, 'base'), (r, 'residual'), (f, 'final')):
        np.testing.assert_allclose(t.numpy(), z['eval/' + name], rtol=2e-4, atol=2e-4)
    pkeys = [kk[len('silog/gnorm/'):] for kk in z.files if kk.startswith('silog/gnorm/')]
    for tag, use_silog in (('silog', True), ('l1', False)):
        sdg = {kk: (v.clone().requires_grad_(True) if kk in pkeys else v) for kk, v in sd.items()}
        b, r, f, _ = dcnet_oracle.base_residual_forward(sdg, audio, max_depth, training=True)
        if use_silog:
            for t, name in ((b, 'base'), (r, 'residual'), (f, 'final')):
                np.testing.assert_allclose(t.detach().numpy(), z['train/' + name], rtol=2e-4, atol=2e-4)
        loss, parts = dcnet_oracle.base_residual_loss(b, r, f, gt, gt > 0, lrec, lbase, lsp, int(k), use_silog, slam)
        np.testing.assert_allclose([float(p) for p in parts], z[tag + '/parts'], rtol=2e-4, atol=1e-6)
        assert abs(loss.item() - float(z[tag + '/loss'])) <= 2e-4 * abs(float(z[tag + '/loss']))
        loss.backward()
        for kk in pkeys:
            gr, ref = sdg[kk].grad, z[f'{tag}/gs/' + kk]
            assert abs(float(gr.double().norm()) - float(z[f'{tag}/gnorm/' + kk])) <= 5e-3 * float(z[f'{tag}/gnorm/' + kk]) + 1e-7, kk
            np.testing.assert_allclose(_sample(gr), ref, rtol=5e-3, atol=1e-6 + 5e-3 * np.abs(ref).max(), err_msg=kk)


def test_edge_loss_oracle_matches_reference_golden():
    """oracle/edge_loss_oracle.py against the reference's BinauralAttentionLoss (values + autograd gradients)."""
    from oracle import edge_loss_oracle as E
    z = np.load(os.path.join(GOLDEN, 'edge_loss_cases.npz'))
    for name in ('random', 'box', 'all_invalid'):
        for tag in ('default', 'heavy'):
            k = f'{name}/{tag}/'
            pred = torch.from_numpy(z[name + '/pred']).double().requires_grad_(True)
            gt = torch.from_numpy(z[name + '/gt']).double()
            total, (r, e, s) = E.edge_loss(pred, gt, *[float(v) for v in z[k + 'lambdas']])
            got = np.array([float(r), float(e), float(s), float(total)])
            np.testing.assert_allclose(got, z[k + 'terms'], rtol=2e-5, atol=1e-7)
            if total.requires_grad and name != 'all_invalid':
                total.backward()
                ref = z[k + 'grad']
                assert float(np.abs(pred.grad.numpy() - ref).max()) <= 2e-5 * float(np.abs(ref).max()) + 1e-9
