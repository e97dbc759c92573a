"""Reference-generated fixtures of the DoubleConv family at FULL width and 256 x 256 (build container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_fullsize_dc.py

  binaural_l2_bc64_256.npz   models/binaural_attention_model.py:347 create_binaural_attention_model(base_channels=64,
                             output_size=256, attention_levels=[2]), B = 1: the cross-attention of level 2 at its real size
                             (C = 128, 128 x 128 -> 16 384 tokens, :106-153; the reference materialises two 16 384^2 f32
                             score matrices), gamma = 0.5; masked Combined L1 + SIlog loss, backward
                             (train_binaural_attention.py:399-433)
  rgb_bc64_256.npz           models/rgb_depth_model.py:225 create_rgb_depth_model(base_channels=64, output_size=256), B = 2;
                             DepthLoss (train_rgb_depth.py:43-87, compiled from the file's AST), backward

The 29 M / 17 M parameters are not stored: the mirror regenerates them from the seed (checked against per-tensor checksums)
and both sides apply `perturb` below (keyed by parameter name, so module iteration order does not matter).  Stored: 8192
sampled points of the train-mode prediction and of d loss / d pred, the loss, per-tensor gradient norms + 512-element samples.
"""
import ast
import contextlib
import io
import os
import sys

import numpy as np
import torch

sys.path.insert(0, '/root/reference')
from models.binaural_attention_model import create_binaural_attention_model  # noqa: E402
from models.rgb_depth_model import create_rgb_depth_model                    # noqa: E402
from utils_loss import SIlogLoss                                             # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
NS = 512


def hash_key(key):
    h = 0
    for ch in key:
        h = (h * 131 + ord(ch)) % (2 ** 31 - 1)
    return h


def sample_idx(numel, key, ns=NS):
    g = torch.Generator().manual_seed(hash_key(key))
    return torch.randint(0, numel, (min(ns, numel),), generator=g)


def perturb(model):
    """Non-trivial BatchNorm affine parameters, attention gate and attention biases, as a function of the tensor's NAME
    (the test applies the same function to the mirror)."""
    with torch.no_grad():
        for k, v in model.state_dict().items():
            g = torch.Generator().manual_seed(hash_key(k))
            if k.endswith('.gamma'):
                v.fill_(0.5)                                   # gamma = 0 (the init) switches the attention path off
            elif 'attention_modules' in k and k.endswith('.bias'):
                v.copy_(0.1 * torch.randn(v.shape, generator=g))
            elif v.dim() == 1 and k.endswith('.weight'):       # BatchNorm gamma
                v.copy_(1.0 + 0.2 * torch.randn(v.shape, generator=g))
            elif v.dim() == 1 and k.endswith('.bias') and ('double_conv' in k or 'fusion' in k):   # BatchNorm beta
                v.copy_(0.1 * torch.randn(v.shape, generator=g))


def reference_depth_loss():
    src = open('/root/reference/train_rgb_depth.py').read()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == 'create_depth_loss')
    ns = {'torch': torch, 'nn': torch.nn}
    exec(compile(ast.Module([fn], []), 'train_rgb_depth.py', 'exec'), ns)
    return ns['create_depth_loss']()


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def finish(name, model, pred, loss, out):
    g = torch.Generator().manual_seed(77)
    idx = torch.randint(0, pred.numel(), (8192,), generator=g)
    out['idx'] = idx.numpy()
    out['pred_val'] = pred.detach().reshape(-1)[idx].numpy()
    out['pred_grad_val'] = pred.grad.detach().reshape(-1)[idx].numpy()
    out['pred_min'] = np.float64(pred.detach().min().item())
    out['loss'] = np.float64(loss.item())
    for k, prm in model.named_parameters():
        gflat = prm.grad.detach().reshape(-1)
        out['gnorm/' + k] = np.float64(gflat.double().norm().item())
        out['gsample/' + k] = gflat[sample_idx(gflat.numel(), k)].numpy()
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print(name, 'loss', float(loss), 'pred range', float(pred.min()), float(pred.max()), 'bytes', os.path.getsize(path))


def checksums(model, out):
    for k, v in model.state_dict().items():
        if v.dtype.is_floating_point:
            out['init_sum/' + k] = np.float64(v.double().sum().item())
            out['init_abs/' + k] = np.float64(v.double().abs().sum().item())


def binaural_l2(S=256, B=1, max_depth=30.0, l1_w=0.5, silog_w=0.5, lam=0.5):
    torch.manual_seed(0)
    model = quiet(create_binaural_attention_model, base_channels=64, bilinear=True, output_size=S, max_depth=max_depth,
                  attention_levels=[2])
    out = {}
    checksums(model, out)                      # of the seed's weights, BEFORE the perturbation
    perturb(model)
    with torch.no_grad():
        # logits of a freshly initialised net spread over +-20: sigmoid * 30 then reaches 1e-8 m, where SIlog's 1 / pred lets
        # single pixels set every gradient norm (see make_golden_unet64.py main_b32).  A tenth of the head's weights keeps
        # every prediction in [8, 22] m; the attention path is unaffected.
        model.outc[0].weight.mul_(0.1)
    g = torch.Generator().manual_seed(1234)
    audio = torch.rand(B, 2, S, S, generator=g)
    gt = max_depth * torch.rand(B, 1, S, S, generator=g)
    gt[gt < 0.1 * max_depth] = 0.0
    model.train()
    pred = model(audio)
    valid = gt > 0
    loss = l1_w * torch.nn.L1Loss()(pred[valid], gt[valid]) + silog_w * SIlogLoss(lambda_scale=lam)(pred[valid], gt[valid])
    pred.retain_grad()
    loss.backward()
    out['hyper'] = np.array([max_depth, l1_w, silog_w, lam], dtype=np.float64)
    out['meta'] = np.array([64, S, B], dtype=np.int64)
    finish('binaural_l2_bc64_256', model, pred, loss, out)


def rgb_full(S=256, B=2, max_depth=30.0):
    torch.manual_seed(0)
    model = quiet(create_rgb_depth_model, base_channels=64, bilinear=True, output_size=S, max_depth=max_depth)
    out = {}
    checksums(model, out)
    perturb(model)
    with torch.no_grad():
        model.outc.bias.fill_(2.0)             # most pixels inside the clamp range
    g = torch.Generator().manual_seed(1234)
    image = torch.rand(B, 3, S, S, generator=g)
    gt = max_depth * torch.rand(B, 1, S, S, generator=g)
    gt[gt < 0.1 * max_depth] = 0.0
    model.train()
    pred = model(image)
    loss, parts = reference_depth_loss()(pred, gt)
    pred.retain_grad()
    loss.backward()
    out['hyper'] = np.array([max_depth, 1.0, 0.1], dtype=np.float64)
    out['meta'] = np.array([64, S, B], dtype=np.int64)
    finish('rgb_bc64_256', model, pred, loss, out)


if __name__ == '__main__':
    torch.set_num_threads(8)
    which = sys.argv[1:] or ['rgb', 'binaural']
    if 'rgb' in which:
        rgb_full()
    if 'binaural' in which:
        binaural_l2()
