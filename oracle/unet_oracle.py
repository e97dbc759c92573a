"""CPU restatement of the reference U-Net generator (TEST INFRASTRUCTURE ONLY).

Restates /root/reference/models/unetbaseline_model.py:
  * UnetGenerator.__init__            :123-148  (level nesting, channel widths)
  * UnetSkipConnectionBlock.__init__  :157-229  (per-level layer list, bias rules)
  * UnetSkipConnectionBlock.forward   :231-235  (skip concat order [x, model(x)])
as a *functional* over a flat ``state_dict`` with the reference's key names, so the same
tensors can be fed to the reference module, to this oracle and to the HIP engine.

The in-place LeakyReLU(0.2, True) at :189 mutates the skip tensor before the concat at :235;
because every concat is consumed through ReLU (:190 uprelu) the effective skip is ReLU(x)
and forward/backward are identical to the non-aliased form used here (SURVEY.md section 7,
"In-place LeakyReLU aliasing"; pinned by tests/golden/unet_*.npz).

Pinned by: tests/test_oracle_golden.py against tests/golden/unet_*.npz.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

BN_EPS = 1e-5       # nn.BatchNorm2d default (unetbaseline_model.py:69)
BN_MOMENTUM = 0.1   # nn.BatchNorm2d default


def num_downs_of(netG: str) -> int:
    """define_G netG switch (unetbaseline_model.py:114-119)."""
    if netG == 'unet_128':
        return 7
    if netG == 'unet_256':
        return 8
    raise NotImplementedError('Generator model name [%s] is not recognized' % netG)


def level_channels(num_downs: int, ngf: int, input_nc: int, output_nc: int):
    """(Cin_down, Cout_down, Cin_up, Cout_up) per level, outermost = level 0.

    unetbaseline_model.py:141-148: innermost ngf*8->ngf*8, (num_downs-5) middle levels at
    ngf*8, then ngf*4->ngf*8, ngf*2->ngf*4, ngf->ngf*2, outermost input_nc->ngf.
    """
    down_out = [ngf, ngf * 2, ngf * 4, ngf * 8] + [ngf * 8] * (num_downs - 4)
    down_in = [input_nc] + down_out[:-1]
    levels = []
    for i in range(num_downs):
        cin_d, cout_d = down_in[i], down_out[i]
        innermost = i == num_downs - 1
        cin_u = cout_d if innermost else cout_d * 2          # :196-198 / :209-211 / :218-220
        cout_u = output_nc if i == 0 else down_in[i]          # outer_nc of the block
        levels.append((cin_d, cout_d, cin_u, cout_u))
    return levels


def level_prefix(i: int) -> str:
    """state_dict prefix of the nn.Sequential of level i (0 = outermost)."""
    p = 'model.model'
    if i >= 1:
        p += '.1.model'
    for _ in range(max(0, i - 1)):
        p += '.3.model'
    return p


def level_keys(i: int, num_downs: int) -> dict:
    """Key names of the layers of level i (Sequential indices from :199-229)."""
    p = level_prefix(i)
    if i == 0:                       # [downconv, submodule, uprelu, upconv, ReLU|Sigmoid]
        return {'down': p + '.0', 'up': p + '.3', 'bn_d': None, 'bn_u': None}
    if i == num_downs - 1:           # [downrelu, downconv, uprelu, upconv, upnorm]
        return {'down': p + '.1', 'up': p + '.3', 'bn_d': None, 'bn_u': p + '.4'}
    # [downrelu, downconv, downnorm, submodule, uprelu, upconv, upnorm]
    return {'down': p + '.1', 'up': p + '.5', 'bn_d': p + '.2', 'bn_u': p + '.6'}


def _bn(x, sd, key, training, new_stats):
    w, b = sd[key + '.weight'], sd[key + '.bias']
    rm, rv = sd[key + '.running_mean'], sd[key + '.running_var']
    if training:
        rm2, rv2 = rm.clone(), rv.clone()
        y = F.batch_norm(x, rm2, rv2, w, b, True, BN_MOMENTUM, BN_EPS)
        new_stats[key + '.running_mean'] = rm2
        new_stats[key + '.running_var'] = rv2
        return y
    return F.batch_norm(x, rm, rv, w, b, False, BN_MOMENTUM, BN_EPS)


def unet_forward(sd: dict, x: torch.Tensor, num_downs: int, depth_norm: bool,
                 training: bool = True):
    """Forward of define_G(...)'s module.  Returns (out, new_running_stats).

    ``sd`` maps reference key names to tensors (leaf tensors with requires_grad for a
    backward through torch autograd).  Follows the Sequential order of
    unetbaseline_model.py:199-229 level by level.
    """
    new_stats = {}
    skips = []          # x_i = input of level i  (x_0 = network input)
    h = x
    for i in range(num_downs):
        k = level_keys(i, num_downs)
        skips.append(h)
        a = h if i == 0 else F.leaky_relu(h, 0.2)                    # downrelu :189
        h = F.conv2d(a, sd[k['down'] + '.weight'], None, stride=2, padding=1)   # :187
        if k['bn_d'] is not None:
            h = _bn(h, sd, k['bn_d'], training, new_stats)            # downnorm :190
    # h = output of the innermost down conv (no norm)
    u = None
    for i in reversed(range(num_downs)):
        k = level_keys(i, num_downs)
        inp = h if i == num_downs - 1 else torch.cat([skips[i + 1], u], 1)   # :235
        a = F.relu(inp)                                               # uprelu :191
        bias = sd.get(k['up'] + '.bias') if i == 0 else None          # :196-198 bias only outermost
        u = F.conv_transpose2d(a, sd[k['up'] + '.weight'], bias, stride=2, padding=1)
        if k['bn_u'] is not None:
            u = _bn(u, sd, k['bn_u'], training, new_stats)            # upnorm :192
    out = torch.sigmoid(u) if depth_norm else F.relu(u)               # :201-206
    return out, new_stats


def param_keys(num_downs: int):
    """Trainable parameter keys in nn.Module.parameters() order (registration order)."""
    def rec(i):
        k = level_keys(i, num_downs)
        out = [k['down'] + '.weight']
        if k['bn_d'] is not None:
            out += [k['bn_d'] + '.weight', k['bn_d'] + '.bias']
        if i < num_downs - 1:
            out += rec(i + 1)
        out.append(k['up'] + '.weight')
        if i == 0:
            out.append(k['up'] + '.bias')
        if k['bn_u'] is not None:
            out += [k['bn_u'] + '.weight', k['bn_u'] + '.bias']
        return out
    return rec(0)


def conv2d_direct_numpy(x, w, stride, pad):
    """Definition-level conv2d (float64 numpy, tiny sizes) used to pin F.conv2d itself."""
    import numpy as np
    B, C, H, W = x.shape
    K, _, R, S = w.shape
    Ho = (H + 2 * pad - R) // stride + 1
    Wo = (W + 2 * pad - S) // stride + 1
    xp = np.zeros((B, C, H + 2 * pad, W + 2 * pad), dtype=np.float64)
    xp[:, :, pad:pad + H, pad:pad + W] = x
    y = np.zeros((B, K, Ho, Wo), dtype=np.float64)
    for r in range(R):
        for s in range(S):
            patch = xp[:, :, r:r + stride * Ho:stride, s:s + stride * Wo:stride]
            y += np.einsum('bchw,kc->bkhw', patch, w[:, :, r, s].astype(np.float64))
    return y


def conv_transpose2d_direct_numpy(x, w, stride, pad):
    """Definition-level ConvTranspose2d: scatter form out[2i-pad+r] += x[i]*w[:, :, r, s]."""
    import numpy as np
    B, C, H, W = x.shape
    _, K, R, S = w.shape
    Ho = (H - 1) * stride - 2 * pad + R
    Wo = (W - 1) * stride - 2 * pad + S
    full = np.zeros((B, K, (H - 1) * stride + R, (W - 1) * stride + S), dtype=np.float64)
    for r in range(R):
        for s in range(S):
            full[:, :, r:r + stride * H:stride, s:s + stride * W:stride] += np.einsum(
                'bchw,ck->bkhw', x.astype(np.float64), w[:, :, r, s].astype(np.float64))
    return full[:, :, pad:pad + Ho, pad:pad + Wo]
