"""Generate the DoubleConv-family golden vectors by IMPORTING THE REFERENCE (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_dcnet.py

Needs /root/reference (read-only mount).  Only numeric inputs/outputs are saved (.npz next to this script).

Reference entry points exercised (file:line in /root/reference):
  models/rgb_depth_model.py:225          create_rgb_depth_model   (base_channels=8, 64x64)
  train_rgb_depth.py:43-87               create_depth_loss()      (executed from the file's AST: the module
                                         itself imports wandb, which is not installed)
  train_rgb_depth.py:355-362, 263-268    forward, loss, backward, AdamW(lr, weight_decay).step()
  models/binaural_attention_model.py:347 create_binaural_attention_model (base_channels=8, 64x64)
  train_binaural_attention.py:399-433    mask gt>0, Combined L1 + SIlog, AdamW step (no clipping)
  utils_loss.py:9                        SIlogLoss
"""
import ast
import contextlib
import io
import os
import sys

import numpy as np
import torch

sys.path.insert(0, '/root/reference')
from models.rgb_depth_model import create_rgb_depth_model                   # noqa: E402
from models.binaural_attention_model import create_binaural_attention_model  # noqa: E402
from utils_loss import SIlogLoss                                             # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def reference_depth_loss():
    """create_depth_loss() of train_rgb_depth.py, compiled from the reference file without importing the
    script (it needs wandb / cv2 at import time)."""
    src = open('/root/reference/train_rgb_depth.py').read()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == 'create_depth_loss')
    ns = {'torch': torch, 'nn': torch.nn}
    exec(compile(ast.Module([fn], []), 'train_rgb_depth.py', 'exec'), ns)
    return ns['create_depth_loss']()


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def save(name, out):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print(name, 'loss', float(out['loss']), 'bytes', os.path.getsize(path))


def perturb_bn(model, seed):
    """Non-trivial BN affine parameters / running stats so that eval mode and the BN gradients are exercised."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(1.0 + 0.2 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
                m.running_mean.copy_(0.1 * torch.randn(m.running_mean.shape, generator=g))
                m.running_var.copy_(1.0 + 0.5 * torch.rand(m.running_var.shape, generator=g))


def rgb_case(name='rgb64_bc8', bc=8, S=64, B=2, lr=1e-4, wd=0.01, max_depth=30.0, bilinear=True, in_size=None):
    torch.manual_seed(0)
    model = quiet(create_rgb_depth_model, base_channels=bc, bilinear=bilinear, output_size=S, max_depth=max_depth)
    perturb_bn(model, 1)
    with torch.no_grad():
        model.outc.bias.fill_(2.0)          # most pixels inside the clamp range; some still hit clamp(0)
    out = {'sd0/' + k: v.detach().clone().numpy() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(1234)
    Si = in_size or S                   # in_size != S: the model resizes its head output to S x S (reference :200-206)
    image = torch.rand(B, 3, Si, Si, generator=g)
    gt = max_depth * torch.rand(B, 1, S, S, generator=g)
    gt[gt < 0.1 * max_depth] = 0.0
    out['image'], out['gt'] = image.numpy(), gt.numpy()
    model.eval()
    with torch.no_grad():
        out['pred_eval'] = model(image).numpy()
    model.train()
    criterion = reference_depth_loss()
    opt = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=wd)       # train_rgb_depth.py:263-268
    pred, feats = model(image, return_features=True)
    for k in ('x1', 'x5', 'd4', 'd1'):
        out['feat/' + k] = feats[k].detach().numpy()
    loss, parts = criterion(pred, gt)
    opt.zero_grad()
    pred.retain_grad()
    loss.backward()
    out['pred_train'], out['pred_grad'] = pred.detach().numpy(), pred.grad.numpy()
    out['loss'] = np.float64(loss.item())
    out['loss_parts'] = np.array([parts['l1'], parts['smooth']], dtype=np.float64)
    for k, p in model.named_parameters():
        out['grad/' + k] = p.grad.detach().clone().numpy()
    opt.step()
    for k, v in model.state_dict().items():
        out['sd1/' + k] = v.detach().clone().numpy()
    out['meta'] = np.array([bc, S, B], dtype=np.int64)
    out['hyper'] = np.array([lr, wd, max_depth, 1.0, 0.1], dtype=np.float64)
    save(name, out)


def binaural_case(name='binaural64_bc8', bc=8, S=64, B=2, lr=1e-3, wd=0.01, max_depth=30.0, l1_w=0.5, silog_w=0.5,
                  lam=0.5):
    torch.manual_seed(0)
    model = quiet(create_binaural_attention_model, base_channels=bc, bilinear=True, output_size=S,
                  max_depth=max_depth, attention_levels=[2, 3, 4, 5])
    perturb_bn(model, 2)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for k, m in model.attention_modules.items():
            m.gamma.fill_(0.5)              # gamma = 0 (the init) would switch the attention path off
            for conv in (m.query, m.key, m.value, m.out):
                conv.bias.copy_(0.1 * torch.randn(conv.bias.shape, generator=g))
    out = {'sd0/' + k: v.detach().clone().numpy() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(1234)
    audio = torch.rand(B, 2, S, S, generator=g)
    gt = max_depth * torch.rand(B, 1, S, S, generator=g)
    gt[gt < 0.1 * max_depth] = 0.0
    out['audio'], out['gt'] = audio.numpy(), gt.numpy()
    model.eval()
    with torch.no_grad():
        out['pred_eval'] = model(audio).numpy()
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=wd)       # train_binaural_attention.py:320-325
    pred = model(audio)
    valid = gt > 0                                                            # :403
    l1 = torch.nn.L1Loss()
    silog = SIlogLoss(lambda_scale=lam)
    loss = l1_w * l1(pred[valid], gt[valid]) + silog_w * silog(pred[valid], gt[valid])    # :418-420
    opt.zero_grad()
    pred.retain_grad()
    loss.backward()
    out['pred_train'], out['pred_grad'] = pred.detach().numpy(), pred.grad.numpy()
    out['loss'] = np.float64(loss.item())
    for k, p in model.named_parameters():
        out['grad/' + k] = p.grad.detach().clone().numpy()
    opt.step()
    for k, v in model.state_dict().items():
        out['sd1/' + k] = v.detach().clone().numpy()
    out['meta'] = np.array([bc, S, B], dtype=np.int64)
    out['hyper'] = np.array([lr, wd, max_depth, l1_w, silog_w, lam], dtype=np.float64)
    save(name, out)




def perturb_biases(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():                       # non-trivial biases (the init sets all conv biases to 0)
        for m in model.modules():
            if isinstance(m, (torch.nn.Linear, torch.nn.Conv2d)) and m.bias is not None:
                m.bias.copy_(0.2 * torch.randn(m.bias.shape, generator=g))


def sample(t, n=512):
    """Deterministic strided sample of a tensor (whole tensor when small)."""
    f = t.detach().reshape(-1)
    if f.numel() <= n:
        return f.clone().numpy()
    idx = torch.linspace(0, f.numel() - 1, n).long()
    return f[idx].clone().numpy()


def adabins_case(name='adabins32_bc64', bc=64, nb=128, S=32, B=2, lr=2e-3, max_depth=30.0):
    """models/adabins_distillation_model.py:462 create_adabins_distillation_model, utils_distillation_loss.py:20
    DistillationLoss with the trainer's default weights (train_adabins_distillation.py:179-188), one step of
    train_adabins_distillation.py:445-456 (mode='train' with the RGB teacher, gt>0 mask, clip_grad_norm_(1.0),
    AdamW).  The reference decoder hard-codes its channel counts (Up(1024,..), Up(768,..), :186-189), so the model
    only exists at base_channels=64 (42.6 M parameters): the fixture holds inputs, outputs, the loss terms and, per
    parameter, the gradient norm plus a strided 512-element sample of the gradient and of the updated value; the
    initial weights are regenerated from the seed (same-seed-same-weights) and the perturbations below.
    Dropout(0.1) of the bin predictors is set to p=0: its draw comes from torch's global RNG stream and cannot
    be reproduced by another implementation."""
    from models.adabins_distillation_model import create_adabins_distillation_model
    from utils_distillation_loss import DistillationLoss
    torch.manual_seed(0)
    model = create_adabins_distillation_model(n_bins=nb, base_channels=bc, output_size=S, max_depth=max_depth)
    perturb_bn(model, 3)
    perturb_biases(model, 9)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    out = {}
    for k, v in model.state_dict().items():
        out['sd0s/' + k] = sample(v) if v.is_floating_point() else v.clone().numpy()
    g = torch.Generator().manual_seed(1234)
    audio = torch.rand(B, 2, S, S, generator=g)
    rgb = torch.rand(B, 3, S, S, generator=g)
    gt = max_depth * torch.rand(B, 1, S, S, generator=g)
    gt[gt < 0.1 * max_depth] = 0.0
    out['audio'], out['rgb'], out['gt'] = audio.numpy(), rgb.numpy(), gt.numpy()
    model.eval()
    with torch.no_grad():
        o = model(audio, rgb=None, mode='inference')
        out['eval/final_depth'] = o['audio']['final_depth'].numpy()
        out['eval/bin_centers'] = o['audio']['bin_centers'].numpy()
    model.train()
    crit = DistillationLoss(lambda_task=1.0, lambda_response=0.5, lambda_feature=0.3, lambda_bin=0.2,
                            lambda_sparse=0.1, temperature=4.0)
    opt = torch.optim.AdamW(filter(lambda p: p.requires_grad, model.parameters()), lr=lr)
    o = model(audio, rgb=rgb, mode='train')
    for side in ('audio', 'rgb'):
        for k in ('bin_centers', 'bin_widths', 'base_depth', 'residual', 'final_depth'):
            out[f'train/{side}/{k}'] = o[side][k].detach().numpy()
        out[f'train/{side}/logits_mean'] = o[side]['bin_logits'].detach().mean((2, 3)).numpy()
        out[f'train/{side}/x5'] = o[side]['features']['x5'].detach().numpy()
    valid = gt > 0
    loss, parts = crit(o, gt, valid)
    opt.zero_grad()
    loss.backward()
    out['loss'] = np.float64(loss.item())
    out['loss_parts'] = np.array([parts[k] for k in ('task', 'response', 'feature', 'bin', 'bin_centers', 'sparse')],
                                 dtype=np.float64)
    for k, p in model.named_parameters():
        if p.grad is not None:
            out['gnorm/' + k] = np.float64(p.grad.double().norm().item())
            out['gs/' + k] = sample(p.grad)
    tn = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
    out['grad_norm'] = np.float64(tn.item())
    opt.step()
    for k, v in model.state_dict().items():
        out['sd1s/' + k] = sample(v) if v.is_floating_point() else v.clone().numpy()
    out['meta'] = np.array([bc, nb, S, B], dtype=np.int64)
    out['hyper'] = np.array([lr, max_depth, 1.0, 0.5, 0.3, 0.2, 0.1, 4.0], dtype=np.float64)
    save(name, out)


def base_residual_case(name='baseres32_bc64', bc=64, S=32, B=2, lr=2e-3, max_depth=30.0):
    """models/base_residual_model.py:236 create_base_residual_model (its base decoder hard-codes 1024/384/192/96 input
    channels, :113-116, so it only exists at base_channels=64), utils_base_residual_loss.py:28 BaseResidualLoss with
    the trainer's settings (train_base_residual.py:272-281: SIlog reconstruction, lambda 1.0 / 1.2 / 0.05, kernel 16),
    one step of train_base_residual.py:375-388 (gt > 0 mask, clip_grad_norm_(1.0), AdamW).  Sampled fixture as for
    AdaBins; the head biases are shifted so that base + residual stays away from 0 where SIlog's 1/pred is
    ill-conditioned."""
    from models.base_residual_model import create_base_residual_model
    from utils_base_residual_loss import BaseResidualLoss
    torch.manual_seed(0)
    model = create_base_residual_model(input_channels=2, base_channels=bc, output_size=S, max_depth=max_depth)
    perturb_bn(model, 4)
    out = {}
    for k, v in model.state_dict().items():
        out['sd0s/' + k] = sample(v) if v.is_floating_point() else v.clone().numpy()
    g = torch.Generator().manual_seed(1234)
    audio = torch.rand(B, 2, S, S, generator=g)
    gt = max_depth * torch.rand(B, 1, S, S, generator=g)
    gt[gt < 0.1 * max_depth] = 0.0
    out['audio'], out['gt'] = audio.numpy(), gt.numpy()
    model.eval()
    with torch.no_grad():
        b, r, f = model(audio)
        out['eval/base'], out['eval/residual'], out['eval/final'] = b.numpy(), r.numpy(), f.numpy()
    model.train()
    for use_silog, tag in ((True, 'silog'), (False, 'l1')):
        crit = BaseResidualLoss(lambda_recon=1.0, lambda_base=1.2, lambda_sparse=0.05, lowpass_kernel=16,
                                use_silog=use_silog, silog_lambda=0.5)
        model.zero_grad()
        b, r, f = model(audio)
        loss, parts = crit(b, r, f, gt, gt > 0)
        loss.backward()
        out[f'{tag}/loss'] = np.float64(loss.item())
        out[f'{tag}/parts'] = np.array([parts['recon'], parts['base'], parts['sparse']], dtype=np.float64)
        for k, p in model.named_parameters():
            out[f'{tag}/gnorm/' + k] = np.float64(p.grad.double().norm().item())
            out[f'{tag}/gs/' + k] = sample(p.grad)
        if use_silog:
            out['train/base'], out['train/residual'], out['train/final'] = b.detach().numpy(), r.detach().numpy(), f.detach().numpy()
            import torch.nn.functional as F_
            s = F_.avg_pool2d(gt, kernel_size=16, stride=1, padding=8)
            out['struct'] = F_.interpolate(s, size=gt.shape[-2:], mode='bilinear', align_corners=False).numpy()
    out['loss'] = out['silog/loss']
    out['meta'] = np.array([bc, S, B], dtype=np.int64)
    out['hyper'] = np.array([lr, max_depth, 1.0, 1.2, 0.05, 16, 0.5], dtype=np.float64)
    save(name, out)


if __name__ == '__main__':
    import sys as _sys
    torch.set_num_threads(8)
    which = _sys.argv[1:] or ['rgb', 'binaural', 'adabins', 'baseres']
    if 'rgb' in which:
        rgb_case()
    if 'rgbresize' in which:            # 32 x 32 input, output_size 64: final F.interpolate before the clamp
        rgb_case('rgbresize32to64_bc8', in_size=32)
    if 'rgbconvt' in which:             # Up(bilinear=False): ConvTranspose2d(k 2, s 2) upsampling
        rgb_case('rgbconvt64_bc8', bilinear=False)
    if 'binaural' in which:
        binaural_case()
    if 'adabins' in which:
        adabins_case()
    if 'baseres' in which:
        base_residual_case()
