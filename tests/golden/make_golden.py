"""Generate golden vectors by IMPORTING THE REFERENCE (runs only in the build container).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Needs /root/reference (read-only mount).  Nothing of the reference travels: only the
numeric inputs/outputs below are saved as .npz next to this script.  The fixtures pin
oracle/ (tests/test_oracle_golden.py) and, through identical inputs, the HIP path
(tests/test_gpu_*.py).

Reference entry points exercised (file:line in /root/reference):
  models/unetbaseline_model.py:84  define_G            (unet_256 ngf=4, unet_128 ngf=4 depth_norm)
  utils_loss.py:9                  SIlogLoss
  utils_criterion.py:6             compute_errors
  train.py:646-691                 loss assembly, clip_grad_norm_(1.0), AdamW(lr) step
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, '/root/reference')
from models.unetbaseline_model import define_G          # noqa: E402
from utils_loss import SIlogLoss                         # noqa: E402
from utils_criterion import compute_errors               # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
L1_W, SILOG_W, SILOG_LAMBDA = 0.237, 0.637, 0.869       # conf/mode/train.yaml:12-14


def synth_batch(B, C, S, seed, max_depth=30.0, depth_norm=False):
    """SURVEY.md section 8(d) synthetic inputs."""
    g = torch.Generator().manual_seed(seed)
    audio = torch.rand(B, C, S, S, generator=g)
    gt = max_depth * torch.rand(B, 1, S, S, generator=g)
    gt[gt < 0.1 * max_depth] = 0.0
    if depth_norm:
        gt = gt / max_depth
    return audio, gt


def unet_case(name, netG, ngf, S, depth_norm, max_depth, criterion, B=2, lr=0.002, out_bias=None, wscale=1.0):
    cfg = SimpleNamespace(dataset=SimpleNamespace(depth_norm=depth_norm, max_depth=max_depth))
    torch.manual_seed(0)
    model = define_G(cfg, input_nc=2, output_nc=1, ngf=ngf, netG=netG, norm='batch',
                     use_dropout=False, init_type='normal', init_gain=0.02, gpu_ids=[])
    out = {}
    out['sd_init/model.model.0.weight'] = model.state_dict()['model.model.0.weight'].clone().numpy()
    if out_bias is not None:
        # A freshly initialised ReLU head predicts ~0 m, where SIlog's 1/pred makes d loss/d pred
        # ill-conditioned (a 5e-7 change of pred moves the gradient by 1%).  Shift the outermost bias so
        # the vectors test the kernels, not the conditioning of log() near 0.
        with torch.no_grad():
            model.model.model[3].bias.fill_(out_bias)
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    for k, v in sd0.items():
        out['sd0/' + k] = v.numpy()
    audio, gt = synth_batch(B, 2, S, 1234, max_depth, depth_norm)
    out['audio'] = audio.numpy()
    out['gt'] = gt.numpy()

    # eval-mode forward with the initial running stats
    model.eval()
    with torch.no_grad():
        out['pred_eval'] = model(audio).numpy()

    # one training step exactly as train.py:633-691
    model.train()
    optimizer = torch.optim.AdamW(model.parameters(), lr=lr)        # train.py:473-474
    optimizer.zero_grad()
    pred = model(audio)
    valid = gt != 0.0                                               # train.py:646
    scale = max_depth if depth_norm else 1.0                        # train.py:649-652
    p, g = pred[valid] * scale, gt[valid] * scale
    l1 = torch.nn.L1Loss()
    silog = SIlogLoss(lambda_scale=SILOG_LAMBDA)
    if criterion == 'Combined':
        loss = wscale * L1_W * l1(p, g) + wscale * SILOG_W * silog(p, g)   # train.py:656-658
    elif criterion == 'L1':
        loss = l1(p, g)
    else:
        loss = silog(p, g)
    pred.retain_grad()
    loss.backward()
    out['pred_grad'] = pred.grad.detach().clone().numpy()
    out['pred_train'] = pred.detach().numpy()
    out['loss'] = np.float64(loss.item())
    for k, prm in model.named_parameters():
        out['grad/' + k] = prm.grad.detach().clone().numpy()
    total_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)   # train.py:689
    out['grad_norm'] = np.float64(total_norm.item())
    optimizer.step()                                                 # train.py:691
    for k, v in model.state_dict().items():
        out['sd1/' + k] = v.detach().clone().numpy()
    out['meta'] = np.array([ngf, S, int(depth_norm), B], dtype=np.int64)
    out['hyper'] = np.array([lr, max_depth, wscale * L1_W, wscale * SILOG_W, SILOG_LAMBDA], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print(name, 'loss', loss.item(), 'grad_norm', total_norm.item(),
          'bytes', os.path.getsize(os.path.join(HERE, name + '.npz')))


def loss_cases():
    out = {}
    g = torch.Generator().manual_seed(7)
    pred = 30 * torch.rand(2, 1, 32, 32, generator=g) - 2.0          # some negatives -> clamp branch
    gt = 30 * torch.rand(2, 1, 32, 32, generator=g)
    gt[gt < 4] = 0.0
    pred.requires_grad_(True)
    for lam in (0.5, 0.869, 1.0):
        crit = SIlogLoss(lambda_scale=lam)
        m = gt != 0
        v = crit(pred[m], gt[m])
        gr, = torch.autograd.grad(v, pred)
        out[f'silog_{lam}'] = np.float64(v.item())
        out[f'silog_grad_{lam}'] = gr.numpy()
    m = gt != 0
    comb = L1_W * torch.nn.L1Loss()(pred[m], gt[m]) + SILOG_W * SIlogLoss(SILOG_LAMBDA)(pred[m], gt[m])
    gr, = torch.autograd.grad(comb, pred)
    out['combined'] = np.float64(comb.item())
    out['combined_grad'] = gr.numpy()
    out['pred'] = pred.detach().numpy()
    out['gt'] = gt.numpy()
    np.savez_compressed(os.path.join(HERE, 'loss_cases.npz'), **out)
    print('loss_cases ok')


def metrics_cases():
    rng = np.random.default_rng(3)
    cases = {}
    gt = (30 * rng.random((64, 64))).astype(np.float32); gt[gt < 3] = 0
    pred = (gt + rng.normal(0, 2, gt.shape)).astype(np.float32)
    cases['metres'] = (gt, pred)
    gtn = (rng.random((64, 64))).astype(np.float32); gtn[gtn < 0.1] = 0
    cases['normalised'] = (gtn, (gtn * (1 + 0.1 * rng.normal(size=gtn.shape))).astype(np.float32))
    cases['all_invalid_gt'] = (np.zeros((8, 8), np.float32), np.ones((8, 8), np.float32))
    cases['pred_all_negative'] = (gt, -np.abs(pred))
    cases['pred_tiny_positive'] = (gt, np.full_like(gt, 5e-4))
    cases['pred_zero'] = (gt, np.zeros_like(gt))
    bg = np.stack([gt, gt[::-1].copy()])
    cases['batched'] = (bg, np.stack([pred, pred[::-1].copy() * 1.1]).astype(np.float32))
    out = {}
    for k, (g_, p_) in cases.items():
        import io, contextlib
        with contextlib.redirect_stdout(io.StringIO()):
            r = compute_errors(g_.copy(), p_.copy())
        out[k + '/gt'] = g_
        out[k + '/pred'] = p_
        out[k + '/ref'] = np.array([float(x) for x in r], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, 'metrics_cases.npz'), **out)
    print('metrics_cases ok')


def optim_cases():
    out = {}
    for opt_name in ('AdamW', 'Adam', 'SGD', 'Adam_wd'):
        torch.manual_seed(11)
        ps = [torch.nn.Parameter(torch.randn(n)) for n in (1000, 37, 4096)]
        if opt_name == 'AdamW':
            opt = torch.optim.AdamW(ps, lr=0.002)
        elif opt_name == 'Adam':
            opt = torch.optim.Adam(ps, lr=0.002)
        elif opt_name == 'Adam_wd':
            opt = torch.optim.Adam(ps, lr=0.001, weight_decay=0.01)   # train_binaural_attention.py:314-318
        else:
            opt = torch.optim.SGD(ps, lr=0.002)
        for i, p in enumerate(ps):
            out[f'{opt_name}/p0/{i}'] = p.detach().clone().numpy()
        for step in range(3):
            gs = [torch.randn_like(p) * (0.005 if step == 1 else 3.0) for p in ps]   # step 1 below clip
            for p, g in zip(ps, gs):
                p.grad = g.clone()
            for i, g in enumerate(gs):
                out[f'{opt_name}/g{step}/{i}'] = g.numpy()
            tn = torch.nn.utils.clip_grad_norm_(ps, max_norm=1.0)
            out[f'{opt_name}/norm{step}'] = np.float64(tn.item())
            opt.step()
            for i, p in enumerate(ps):
                out[f'{opt_name}/p{step + 1}/{i}'] = p.detach().clone().numpy()
    np.savez_compressed(os.path.join(HERE, 'optim_cases.npz'), **out)
    print('optim_cases ok')


if __name__ == '__main__':
    torch.set_num_threads(8)
    unet_case('unet256_ngf4', 'unet_256', 4, 256, False, 30.0, 'Combined', out_bias=1.0, wscale=10.0)   # 10x loss weights (--l1_weight/--silog_weight) so that the clip branch fires
    unet_case('unet128_ngf4_dn', 'unet_128', 4, 128, True, 12.0, 'Combined')
    loss_cases()
    metrics_cases()
    optim_cases()
