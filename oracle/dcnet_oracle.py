"""CPU restatement of the reference's DoubleConv U-Net family (TEST INFRASTRUCTURE ONLY).

Restates, as *functionals* over a flat ``state_dict`` with the reference's key names (so the same tensors can
be fed to the reference module, to this oracle and to the HIP engine):
  * DoubleConv / Down / Up            /root/reference/models/rgb_depth_model.py:21-77 (identical copies in
                                      binaural_attention_model.py:22-78, adabins_distillation_model.py:27-82)
  * RGBDepthNet.forward               rgb_depth_model.py:148-218
  * BinauralCrossAttention.forward    binaural_attention_model.py:106-153
  * BinauralEncoder / BinauralAttentionDepthNet.forward   binaural_attention_model.py:171-178, 280-340
  * DepthLoss.forward                 train_rgb_depth.py:43-87 (L1 + 0.1 * total variation, unmasked)
Backward comes from torch autograd over these functionals (leaf tensors with requires_grad), in float64 when a
test needs a ground truth that is tighter than fp32.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(audio-depth-estimation_amd/) never does.
Pinned by: tests/test_oracle_golden.py against tests/golden/rgb64_bc8.npz and tests/golden/binaural64_bc8.npz
(generated from the reference by tests/golden/make_golden_dcnet.py).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


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


TAPS = None     # tests may set this to a dict to capture the raw conv outputs (with retain_grad) by key
QUANT = None    # tests may set this to a rounding function (e.g. to bf16 and back) that is applied at the HIP
#                 engine's storage points -- conv operands, the stored raw conv output, the stored activations --
#                 with a straight-through gradient: the oracle then models "bf16 storage, exact arithmetic",
#                 which is what the bf16 engine computes up to accumulation order.


def _q(t):
    return t if QUANT is None else t + (QUANT(t) - t).detach()


def _tap(key, h):
    if TAPS is not None:
        if h.requires_grad:
            h.retain_grad()
        TAPS[key] = h
    return h


def _conv_bn_relu(sd, conv_key, bn_key, x, training, new_stats, bias=None, padding=1):
    z = _tap(conv_key, F.conv2d(x, _q(sd[conv_key + '.weight']), bias, padding=padding))
    if QUANT is None or not training:
        return _q(F.relu(_bn(z, sd, bn_key, training, new_stats)))      # eval: BN is fused into the GEMM epilogue
    # engine semantics: batch statistics from the f32 accumulators, normalisation of the STORED (rounded) z
    mu, var = z.mean((0, 2, 3)), z.var((0, 2, 3), unbiased=False)
    n = z.numel() // z.shape[1]
    new_stats[bn_key + '.running_mean'] = (1 - BN_MOMENTUM) * sd[bn_key + '.running_mean'] + BN_MOMENTUM * mu.detach()
    new_stats[bn_key + '.running_var'] = ((1 - BN_MOMENTUM) * sd[bn_key + '.running_var'] +
                                          BN_MOMENTUM * var.detach() * n / max(n - 1, 1))
    sc = sd[bn_key + '.weight'] / torch.sqrt(var + BN_EPS)
    y = (_q(z) - mu.view(1, -1, 1, 1)) * sc.view(1, -1, 1, 1) + sd[bn_key + '.bias'].view(1, -1, 1, 1)
    return _q(F.relu(y))


def double_conv(sd, prefix, x, training, new_stats):
    """(conv3x3 no bias -> BN -> ReLU) x 2; ``prefix`` ends in '.double_conv' (rgb_depth_model.py:28-35)."""
    h = _conv_bn_relu(sd, prefix + '.0', prefix + '.1', x, training, new_stats)
    return _conv_bn_relu(sd, prefix + '.3', prefix + '.4', h, training, new_stats)


def down(sd, prefix, x, training, new_stats):
    """MaxPool2d(2) -> DoubleConv (rgb_depth_model.py:45-49)."""
    return double_conv(sd, prefix + '.maxpool_conv.1.double_conv', F.max_pool2d(x, 2), training, new_stats)


def up(sd, prefix, x1, x2, training, new_stats):
    """bilinear x2 (align_corners=True) -> pad to the skip -> cat([skip, up]) -> DoubleConv (:61-77); with
    bilinear=False the upsampling is ConvTranspose2d(C, C // 2, kernel_size=2, stride=2) (:64-67), recognised by
    its parameters in the state dict."""
    if prefix + '.up.weight' in sd:
        x1 = _q(F.conv_transpose2d(x1, sd[prefix + '.up.weight'], sd[prefix + '.up.bias'], stride=2))
    else:
        x1 = _q(F.interpolate(x1, scale_factor=2, mode='bilinear', align_corners=True))
    dy, dx = x2.shape[2] - x1.shape[2], x2.shape[3] - x1.shape[3]
    x1 = F.pad(x1, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
    return double_conv(sd, prefix + '.conv.double_conv', torch.cat([x2, x1], 1), training, new_stats)


def encoder(sd, prefix, x, training, new_stats):
    """inc, down1..down4 -> [x1..x5] (rgb_depth_model.py:166-170 / binaural_attention_model.py:171-178)."""
    feats = [double_conv(sd, prefix + 'inc.double_conv', x, training, new_stats)]
    for i in range(1, 5):
        feats.append(down(sd, f'{prefix}down{i}', feats[-1], training, new_stats))
    return feats


def decoder(sd, feats, training, new_stats):
    """up1..up4 over [x1..x5] -> [d4, d3, d2, d1] (rgb_depth_model.py:184-187)."""
    d, outs = feats[4], []
    for i in range(4):
        d = up(sd, f'up{i + 1}', d, feats[3 - i], training, new_stats)
        outs.append(d)
    return outs


def _final_resize(depth, output_size):
    """F.interpolate(bilinear, align_corners=False) to output_size^2 when the WIDTH differs (rgb_depth_model.py:200-206)."""
    if output_size is not None and depth.shape[-1] != output_size:
        depth = F.interpolate(depth, size=(output_size, output_size), mode='bilinear', align_corners=False)
    return depth


def rgb_forward(sd, x, max_depth=30.0, training=True, return_features=False, output_size=None):
    """RGBDepthNet.forward (rgb_depth_model.py:148-218); output_size=None means output_size == input size.
    Returns (depth, new_running_stats[, features])."""
    new_stats = {}
    feats = encoder(sd, '', _q(x), training, new_stats)
    ds = decoder(sd, feats, training, new_stats)
    depth = _final_resize(F.conv2d(ds[3], sd['outc.weight'], sd['outc.bias']), output_size)
    depth = torch.clamp(depth, 0, max_depth)                                  # :209
    if return_features:
        f = {f'x{i + 1}': feats[i] for i in range(5)}
        f.update({'d4': ds[0], 'd3': ds[1], 'd2': ds[2], 'd1': ds[3]})
        return depth, new_stats, f
    return depth, new_stats


def cross_attention(sd, prefix, left, right):
    """BinauralCrossAttention.forward (binaural_attention_model.py:106-153): both directions share the
    query/key/value/out projections; softmax over keys of Q^T K / sqrt(C); residual scaled by gamma."""
    B, C, H, W = left.shape

    def proj(name, t):
        return F.conv2d(t, sd[f'{prefix}.{name}.weight'], sd[f'{prefix}.{name}.bias'])

    def attend(q_src, kv_src):
        q = proj('query', q_src).reshape(B, -1, H * W)
        k = proj('key', kv_src).reshape(B, -1, H * W)
        v = proj('value', kv_src).reshape(B, C, H * W)
        att = torch.softmax(torch.bmm(q.transpose(1, 2), k) / (C ** 0.5), dim=-1)       # [B, HW(q), HW(k)]
        o = torch.bmm(v, att.transpose(1, 2)).reshape(B, C, H, W)
        return q_src + sd[prefix + '.gamma'] * proj('out', o)

    return attend(left, right), attend(right, left)


def binaural_forward(sd, x, max_depth=30.0, attention_levels=(2, 3, 4, 5), training=True, output_size=None):
    """BinauralAttentionDepthNet.forward (binaural_attention_model.py:280-340), output_size == input size."""
    new_stats = {}
    lf = encoder(sd, 'left_encoder.', x[:, 0:1], training, new_stats)
    rf = encoder(sd, 'right_encoder.', x[:, 1:2], training, new_stats)
    fused = []
    for level in range(1, 6):
        l, r = lf[level - 1], rf[level - 1]
        if level in attention_levels:
            l, r = cross_attention(sd, f'attention_modules.attn_{level}', l, r)
        p = f'fusion_layers.fusion_{level}'
        h = F.conv2d(torch.cat([l, r], 1), sd[p + '.0.weight'], sd[p + '.0.bias'])
        fused.append(F.relu(_bn(h, sd, p + '.1', training, new_stats)))
    ds = decoder(sd, fused, training, new_stats)
    depth = torch.sigmoid(F.conv2d(ds[3], sd['outc.0.weight'], sd['outc.0.bias'])) * max_depth
    return torch.clamp(_final_resize(depth, output_size), 0, max_depth), new_stats       # :326-337


def depth_loss(pred, target, lambda_l1=1.0, lambda_smooth=0.1):
    """DepthLoss.forward (train_rgb_depth.py:53-85): unmasked L1 + lambda_smooth * (mean|dx| + mean|dy|)."""
    l1 = (pred - target).abs().mean()
    dx = (pred[:, :, :, :-1] - pred[:, :, :, 1:]).abs()
    dy = (pred[:, :, :-1, :] - pred[:, :, 1:, :]).abs()
    return lambda_l1 * l1 + lambda_smooth * (dx.mean() + dy.mean())


# ---- AdaBins distillation model (adabins_distillation_model.py) ----------------------------------------------
def bin_predictor(sd, prefix, x5, max_depth, drop_mask=None, drop_p=0.1):
    """AdaBinsBinPredictor.forward (:127-149): avgpool -> Linear -> ReLU -> Dropout -> Linear -> Softmax -> cumsum
    edges * max_depth -> midpoints.  ``drop_mask`` (0/1, [B,256]) replaces the train-mode Dropout draw."""
    g = x5.mean((2, 3))
    h = F.relu(F.linear(g, sd[prefix + '.predictor.0.weight'], sd[prefix + '.predictor.0.bias']))
    if drop_mask is not None:
        h = h * drop_mask / (1.0 - drop_p)
    widths = torch.softmax(F.linear(h, sd[prefix + '.predictor.3.weight'], sd[prefix + '.predictor.3.bias']), 1)
    edges = torch.cat([torch.zeros_like(widths[:, :1]), torch.cumsum(widths, 1)], 1) * max_depth
    return (edges[:, :-1] + edges[:, 1:]) / 2, widths


def adabins_branch(sd, enc, pred, dec, x, max_depth, training, new_stats, drop_mask=None, output_size=None):
    """forward_audio / forward_rgb (:301-399).  The reference runs the decoder twice on identical inputs (once inside
    the decoder module, once for the residual head); the second pass reproduces the first one's activations, so
    it is evaluated once here and only its side effect -- a second BatchNorm running-stat update in train mode -- is
    replayed (``new_stats`` holds the twice-updated statistics)."""
    feats = encoder(sd, enc + '.', _q(x), training, new_stats)
    centers, widths = bin_predictor(sd, pred, feats[4], max_depth, drop_mask)
    dstats = {}
    d, skip = feats[4], [feats[3], feats[2], feats[1], feats[0]]
    for i in range(4):
        d = up(sd, f'{dec}.up{i + 1}', d, skip[i], training, dstats)
    if training:
        for k, v in dstats.items():                     # second update with the same batch statistic
            once = v
            prev = sd[k]
            batch = (once - (1 - BN_MOMENTUM) * prev) / BN_MOMENTUM
            new_stats[k] = (1 - BN_MOMENTUM) * once + BN_MOMENTUM * batch
    logits = F.conv2d(d, sd[dec + '.class_head.weight'], sd[dec + '.class_head.bias'])
    resid_raw = F.conv2d(d, sd['residual_head.weight'], sd['residual_head.bias'])
    if output_size is not None and logits.shape[-1] != output_size:           # :196-198, :334-337, :383-386
        logits = F.interpolate(logits, size=(output_size, output_size), mode='nearest')
        resid_raw = F.interpolate(resid_raw, size=(output_size, output_size), mode='nearest')
    probs = torch.softmax(logits, 1)
    base = (probs * centers[:, :, None, None]).sum(1, keepdim=True)
    residual = torch.tanh(resid_raw) * (max_depth * 0.05)
    final = torch.clamp(base + residual, 0, max_depth)
    return {'features': {f'x{i + 1}': feats[i] for i in range(5)}, 'bin_centers': centers, 'bin_widths': widths,
            'bin_logits': logits, 'base_depth': base, 'residual': residual, 'final_depth': final}


def adabins_forward(sd, audio, rgb=None, max_depth=30.0, training=True, drop_mask_audio=None, drop_mask_rgb=None,
                    output_size=None):
    """AdaBinsDistillationModel.forward (:401-426); teacher under no_grad.  ``output_size``: the nearest resize of the
    logits / raw residual when it differs from the input size."""
    new_stats = {}
    a = adabins_branch(sd, 'audio_encoder', 'audio_bin_predictor', 'audio_decoder', audio, max_depth, training,
                       new_stats, drop_mask_audio, output_size)
    r = None
    if rgb is not None:
        with torch.no_grad():
            r = adabins_branch(sd, 'rgb_encoder', 'rgb_bin_predictor', 'rgb_decoder', rgb, max_depth, training,
                               new_stats, drop_mask_rgb, output_size)
    return {'audio': a, 'rgb': r}, new_stats


def distillation_loss(output, gt, valid, lambda_task=2.0, lambda_response=0.3, lambda_feature=0.2, lambda_bin=0.05,
                      lambda_sparse=0.1, temperature=4.0):
    """DistillationLoss.forward (utils_distillation_loss.py:147-238).  Returns (total, dict of terms)."""
    a, r = output['audio'], output['rgb']
    task = (a['final_depth'][valid] - gt[valid]).abs().mean()
    zero = torch.zeros((), dtype=gt.dtype)
    resp = feat = kl = cm = zero
    if r is not None:
        resp = ((a['final_depth'][valid] - r['final_depth'][valid].detach()) ** 2).mean()
        tot = 0.0
        for lv in ('x1', 'x2', 'x3', 'x4', 'x5'):
            af = F.normalize(a['features'][lv].flatten(2), dim=2)
            rf = F.normalize(r['features'][lv].detach().flatten(2), dim=2)
            tot = tot + (1 - (af * rf).sum(2).mean())
        feat = tot / 5
        al = F.log_softmax(a['bin_logits'].mean((2, 3)) / temperature, 1)
        rl = torch.softmax(r['bin_logits'].detach().mean((2, 3)) / temperature, 1)
        kl = F.kl_div(al, rl, reduction='batchmean')
        cm = ((a['bin_centers'] - r['bin_centers'].detach()) ** 2).mean()
    sparse = a['residual'][valid].abs().mean()
    total = (lambda_task * task + lambda_response * resp + lambda_feature * feat + lambda_bin * (kl + cm) +
             lambda_sparse * sparse)
    return total, {'task': task, 'response': resp, 'feature': feat, 'bin': kl, 'bin_centers': cm, 'sparse': sparse}


# ---- Base + Residual model (base_residual_model.py, utils_base_residual_loss.py) ----------------------------------------
def base_residual_forward(sd, x, max_depth=30.0, training=True, output_size=None):
    """BaseResidualDepthNet.forward (base_residual_model.py:151-211): shared encoder, base decoder -> sigmoid * max_depth,
    residual decoder -> tanh * 0.3 * max_depth, each ACTIVATED map resized (bilinear, align_corners=False) when its size
    differs from output_size (:185-188, :206-209; None = same size), final = clamp(base + residual)."""
    new_stats = {}
    feats = encoder(sd, '', _q(x), training, new_stats)

    def dec(tag):
        d = feats[4]
        for i in range(4):
            d = up(sd, f'{tag}_up{i + 1}', d, feats[3 - i], training, new_stats)
        return d

    b, r = dec('base'), dec('res')
    base = torch.sigmoid(F.conv2d(b, sd['base_head.weight'], sd['base_head.bias'])) * max_depth
    residual = torch.tanh(F.conv2d(r, sd['res_head.weight'], sd['res_head.bias'])) * (max_depth * 0.3)
    if output_size is not None and tuple(base.shape[-2:]) != (output_size, output_size):
        base = F.interpolate(base, size=(output_size, output_size), mode='bilinear', align_corners=False)
        residual = F.interpolate(residual, size=(output_size, output_size), mode='bilinear', align_corners=False)
    return base, residual, torch.clamp(base + residual, 0, max_depth), new_stats


def lowpass_struct(gt, k=16):
    """Structural target (utils_base_residual_loss.py:91-107)."""
    s = F.avg_pool2d(gt, kernel_size=k, stride=1, padding=k // 2)
    if s.shape != gt.shape:
        s = F.interpolate(s, size=gt.shape[-2:], mode='bilinear', align_corners=False)
    return s


def base_residual_loss(base, residual, final, gt, valid, lambda_recon=1.0, lambda_base=1.2, lambda_sparse=0.05, k=16,
                       use_silog=False, silog_lambda=0.5, eps=1e-6):
    """BaseResidualLoss.forward (utils_base_residual_loss.py:72-160): recon (L1 or SIlog on the masked pixels) +
    lambda_base * L1(base, lowpass(gt)) + lambda_sparse * mean|residual|.  Returns (total, (recon, base, sparse))."""
    with torch.no_grad():
        struct = lowpass_struct(gt, k)
    f, g = final[valid], gt[valid]
    if use_silog:
        d = torch.log(torch.clamp(f, min=eps)) - torch.log(torch.clamp(g, min=eps))        # utils_loss.py:37-47
        recon = torch.sqrt(torch.clamp((d * d).mean() - silog_lambda * d.mean() ** 2, min=0.0))
    else:
        recon = (f - g).abs().mean()
    lb = (base[valid] - struct[valid]).abs().mean()
    ls = residual[valid].abs().mean()
    return lambda_recon * recon + lambda_base * lb + lambda_sparse * ls, (recon, lb, ls)
