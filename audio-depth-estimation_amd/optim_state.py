"""Optimizer state of the fused trainers in ``torch.optim`` format.

The reference checkpoints carry ``optimizer.state_dict()`` (train.py:897-909, :1006-1017 key 'optimizer';
train_binaural_attention.py:351-365, :560-586 key 'optimizer_state_dict', restored with ``optimizer.load_state_dict``).
The fused trainers keep Adam's moments as two flat f32 buffers in ``parameters()`` order with conv weights in
channels_last memory (flat.py); these helpers convert between that and the dict a ``torch.optim.Adam / AdamW / SGD``
over the same parameters writes and reads: per-parameter ``exp_avg`` / ``exp_avg_sq`` tensors in the parameter's logical
[X, Y, kh, kw] shape, a float32 ``step`` tensor, and the ``param_groups`` entry of THIS torch version (taken from a real,
never-stepped torch optimizer so every version-specific key is present).
"""
from __future__ import annotations

import torch

KINDS = {0: 'AdamW', 1: 'Adam', 2: 'SGD'}


def _torch_optimizer(kind, params, lr, betas, eps, weight_decay):
    if kind == 0:
        return torch.optim.AdamW(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
    if kind == 1:
        return torch.optim.Adam(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
    return torch.optim.SGD(params, lr=lr)


def export_state(meta, view, exp_avg, exp_avg_sq, step, kind, lr, betas, eps, weight_decay):
    """meta: [(param, flat offset, numel)] of the OPTIMISED parameters, in order; view(flat, off, p) -> logical-shape view.
    exp_avg / exp_avg_sq: flat buffers indexed by the same offsets (None before the first step)."""
    opt = _torch_optimizer(kind, [p for p, _, _ in meta], lr, betas, eps, weight_decay)
    sd = opt.state_dict()
    if kind != 2 and exp_avg is not None and step > 0:
        for i, (p, off, _) in enumerate(meta):
            sd['state'][i] = {'step': torch.tensor(float(step)),
                              'exp_avg': view(exp_avg, off, p).detach().cpu().contiguous().clone(),
                              'exp_avg_sq': view(exp_avg_sq, off, p).detach().cpu().contiguous().clone()}
    return sd


def is_torch_format(sd):
    return isinstance(sd, dict) and 'param_groups' in sd and 'state' in sd


def import_state(sd, meta, view, exp_avg, exp_avg_sq):
    """Fill the flat moment buffers from a torch-format optimizer state; returns (step, param_group dict).
    Raises ValueError when the saved parameter list does not match the optimised parameters."""
    groups = sd['param_groups']
    if len(groups) != 1:
        # the fused optimizer applies ONE set of hyper-parameters to the whole flat buffer (every reference script builds
        # its optimizer from model.parameters(): one group); silently using group 0's settings for all would diverge
        raise ValueError(f'optimizer state has {len(groups)} param_groups; the fused trainers support exactly one')
    ids = [i for g in groups for i in g['params']]
    if len(ids) != len(meta):
        raise ValueError(f'optimizer state holds {len(ids)} parameters, the model optimises {len(meta)}')
    step = 0
    for k, (p, off, _) in zip(ids, meta):
        st = sd['state'].get(k)
        if not st:
            continue
        for name, buf in (('exp_avg', exp_avg), ('exp_avg_sq', exp_avg_sq)):
            if name in st:
                t = st[name]
                if tuple(t.shape) != tuple(p.shape):
                    raise ValueError(f'optimizer state {name} of parameter {k} has shape {tuple(t.shape)}, expected '
                                     f'{tuple(p.shape)}')
                view(buf, off, p).copy_(t.to(buf.device, torch.float32))
        if 'step' in st:
            step = max(step, int(float(st['step'])))
    return step, groups[0]


def adopt_group(trainer, group):
    """``torch.optim.Optimizer.load_state_dict`` restores lr, betas, eps and weight_decay from the checkpoint's param_group
    (the reference resumes that way, train_binaural_attention.py:351-365): do the same instead of keeping the values the
    trainer was constructed with.  Call BEFORE the step counter / bias corrections are set (they depend on betas)."""
    trainer.lr = float(group.get('lr', trainer.lr))
    if 'betas' in group:
        trainer.betas = (float(group['betas'][0]), float(group['betas'][1]))
    if 'eps' in group:
        trainer.eps = float(group['eps'])
    if 'weight_decay' in group:
        trainer.weight_decay = float(group['weight_decay'])


def import_legacy_flat(sd, meta, exp_avg, exp_avg_sq):
    """Round 1 stored the moments as the flat buffers themselves ('exp_avg' / 'exp_avg_sq' / 'step').  The flat layout
    aligned every parameter to 4 elements then and to 8 now (flat.py), so the buffers are re-sliced parameter by parameter;
    anything that fits neither layout is rejected instead of being loaded shifted."""
    def offsets(align):
        out, o = [], 0
        for _, _, n in meta:
            out.append(o)
            o += (n + align - 1) // align * align
        return out, o
    src_m, src_v = sd['exp_avg'].reshape(-1), sd['exp_avg_sq'].reshape(-1)
    for align in (8, 4):
        offs, total = offsets(align)
        if src_m.numel() == total and src_v.numel() == total:
            for (p, off, n), so in zip(meta, offs):
                exp_avg[off:off + n].copy_(src_m[so:so + n].to(exp_avg.device, torch.float32))
                exp_avg_sq[off:off + n].copy_(src_v[so:so + n].to(exp_avg_sq.device, torch.float32))
            return int(sd['step'])
    raise ValueError(f'flat optimizer state of {src_m.numel()} elements matches neither flat layout of this model '
                     f'({offsets(8)[1]} with 8-element alignment, {offsets(4)[1]} with the round-1 4-element alignment)')
