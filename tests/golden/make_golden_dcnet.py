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


def rgb_case(name='rgb64_bc8', bc=8, S=64, B=2, lr=1e-4, wd=0.01, max_depth=30.0):
    torch.manual_seed(0)
    model = quiet(create_rgb_depth_model, base_channels=bc, bilinear=True, output_size=S, max_depth=max_depth)
    perturb_bn(model, 1)
    with torch.no_grad():
        model.outc.bias.fill_(2.0)          # most pixels inside the clamp range; some still hit clamp(0)
    out = {'sd0/' + k: v.detach().clone().numpy() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(1234)
    image = torch.rand(B, 3, S, S, generator=g)
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


if __name__ == '__main__':
    torch.set_num_threads(8)
    rgb_case()
    binaural_case()
