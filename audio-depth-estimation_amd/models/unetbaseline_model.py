"""U-Net baseline generator on libadn (drop-in for the reference's models/unetbaseline_model.py).

Same public surface as /root/reference/models/unetbaseline_model.py:
  define_G(cfg, input_nc, output_nc, ngf, netG, norm, use_dropout, init_type, init_gain, gpu_ids)  (:84)
  init_net (:42), init_weights (:9), get_norm_layer (:59), Identity (:79),
  UnetGenerator (:123), UnetSkipConnectionBlock (:157)
and the module-level names torch / nn / init / functools that train.py picks up through its
star import (train.py:5-7, :421).  The nn.Module tree (and therefore every state_dict key, e.g.
``model.model.1.model.2.running_var``) is identical to the reference so checkpoints interchange;
the arithmetic is NOT torch's: UnetGenerator.forward hands the whole network to the fused HIP
pipeline in engine.py (implicit-GEMM MFMA convolutions with BatchNorm/activation/skip-concat fused
around them).  There is no CPU path: calling the model with CPU tensors raises RuntimeError.
"""
import functools
import os

import torch
import torch.nn as nn
from torch.nn import init

from ..engine import UNetEngine, run_unet

_DTYPES = {'bf16': torch.bfloat16, 'bfloat16': torch.bfloat16, 'f32': torch.float32, 'fp32': torch.float32,
           'float32': torch.float32}


def default_compute_dtype():
    """bf16 MFMA by default; ADN_COMPUTE_DTYPE=f32 selects the exact-f32 MFMA parity path."""
    return _DTYPES[os.environ.get('ADN_COMPUTE_DTYPE', 'bf16').lower()]


class Identity(nn.Module):
    def forward(self, x):
        return x


def get_norm_layer(norm_type='instance'):
    """batch | instance | none -> layer factory (reference :59-77)."""
    if norm_type == 'batch':
        return functools.partial(nn.BatchNorm2d, affine=True, track_running_stats=True)
    if norm_type == 'instance':
        return functools.partial(nn.InstanceNorm2d, affine=False, track_running_stats=False)
    if norm_type == 'none':
        def norm_layer(x):
            return Identity()
        return norm_layer
    raise NotImplementedError('normalization layer [%s] is not found' % norm_type)


def init_weights(net, init_type='normal', init_gain=0.02):
    """Class-name driven initialisation, same traversal (net.apply) and RNG consumption as reference :9-40."""
    def init_func(m):
        cname = m.__class__.__name__
        is_lin = cname.find('Conv') != -1 or cname.find('Linear') != -1
        if hasattr(m, 'weight') and is_lin:
            if init_type == 'normal':
                init.normal_(m.weight.data, 0.0, init_gain)
            elif init_type == 'xavier':
                init.xavier_normal_(m.weight.data, gain=init_gain)
            elif init_type == 'kaiming':
                init.kaiming_normal_(m.weight.data, a=0, mode='fan_in')
            elif init_type == 'orthogonal':
                init.orthogonal_(m.weight.data, gain=init_gain)
            else:
                raise NotImplementedError('initialization method [%s] is not implemented' % init_type)
            if hasattr(m, 'bias') and m.bias is not None:
                init.constant_(m.bias.data, 0.0)
        elif cname.find('BatchNorm2d') != -1:
            init.normal_(m.weight.data, 1.0, init_gain)
            init.constant_(m.bias.data, 0.0)

    print('initialize network with %s' % init_type)
    net.apply(init_func)


class DataParallel(nn.Module):
    """Key-compatible stand-in for torch.nn.DataParallel (reference :52-56).

    The reference wraps the net whenever gpu_ids is non-empty, which prefixes every state_dict key with
    ``module.``.  Multi-GPU execution here is one process per GPU with RCCL gradient all-reduce
    (ddp.py), so this wrapper only keeps the attribute/key layout and forwards the call.
    """

    def __init__(self, module, device_ids=None):
        super().__init__()
        self.module = module
        self.device_ids = list(device_ids or [])

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)


def init_net(net, init_type='normal', init_gain=0.02, gpu_ids=[]):
    """Device placement, (key-compatible) DataParallel wrap and weight init (reference :42-57)."""
    if len(gpu_ids) > 0:
        assert (torch.cuda.is_available())
        net.to(gpu_ids[0])
        net = DataParallel(net, gpu_ids)
    init_weights(net, init_type, init_gain=init_gain)
    return net


def define_G(cfg, input_nc, output_nc, ngf, netG, norm='batch', use_dropout=False, init_type='normal',
             init_gain=0.02, gpu_ids=[]):
    """Create the generator: 'unet_128' (7 downs) or 'unet_256' (8 downs)  (reference :84-120)."""
    norm_layer = get_norm_layer(norm_type=norm)
    if netG == 'unet_128':
        net = UnetGenerator(cfg, input_nc, output_nc, 7, ngf, norm_layer=norm_layer, use_dropout=use_dropout)
    elif netG == 'unet_256':
        net = UnetGenerator(cfg, input_nc, output_nc, 8, ngf, norm_layer=norm_layer, use_dropout=use_dropout)
    else:
        raise NotImplementedError('Generator model name [%s] is not recognized' % netG)
    return init_net(net, init_type, init_gain, gpu_ids)


class UnetSkipConnectionBlock(nn.Module):
    """One U-Net level: |down conv| -> submodule -> |up transposed conv|, skip = concat([x, model(x)]).

    Holds the parameters in the reference's nn.Sequential layout (:199-229); executed by the fused
    engine of the enclosing UnetGenerator, not layer by layer.
    """

    def __init__(self, cfg, outer_nc, inner_nc, input_nc=None, submodule=None, outermost=False, innermost=False,
                 norm_layer=nn.BatchNorm2d, use_dropout=False):
        super().__init__()
        self.outermost = outermost
        self.innermost = innermost
        norm_cls = norm_layer.func if isinstance(norm_layer, functools.partial) else norm_layer
        use_bias = norm_cls == nn.InstanceNorm2d
        if input_nc is None:
            input_nc = outer_nc
        # creation order matters for same-seed-same-weights parity: downconv, downnorm, upnorm, upconv
        downconv = nn.Conv2d(input_nc, inner_nc, kernel_size=4, stride=2, padding=1, bias=use_bias)
        downrelu = nn.LeakyReLU(0.2, True)
        downnorm = norm_layer(inner_nc)
        uprelu = nn.ReLU(True)
        upnorm = norm_layer(outer_nc)
        up_in = inner_nc if innermost else inner_nc * 2
        upconv = nn.ConvTranspose2d(up_in, outer_nc, kernel_size=4, stride=2, padding=1,
                                    bias=True if outermost else use_bias)
        if outermost:
            last = nn.Sigmoid() if cfg.dataset.depth_norm else nn.ReLU()
            layers = [downconv, submodule, uprelu, upconv, last]
        elif innermost:
            layers = [downrelu, downconv, uprelu, upconv, upnorm]
        else:
            layers = [downrelu, downconv, downnorm, submodule, uprelu, upconv, upnorm]
            if use_dropout:
                layers.append(nn.Dropout(0.5))
        self.model = nn.Sequential(*layers)

    def _parts(self):
        mods = list(self.model)
        conv = next(m for m in mods if isinstance(m, nn.Conv2d))
        convt = next(m for m in mods if isinstance(m, nn.ConvTranspose2d))
        sub = next((m for m in mods if isinstance(m, UnetSkipConnectionBlock)), None)
        norms = [m for m in mods if isinstance(m, (nn.BatchNorm2d, nn.InstanceNorm2d, Identity))]
        drop = any(isinstance(m, nn.Dropout) for m in mods)
        if self.outermost:
            bn_d = bn_u = None
        elif self.innermost:
            bn_d, bn_u = None, norms[0]
        else:
            bn_d, bn_u = norms[0], norms[1]
        return dict(down=conv, up=convt, bn_d=bn_d, bn_u=bn_u, sub=sub, dropout=drop)

    def forward(self, x):
        raise RuntimeError('UnetSkipConnectionBlock is executed by the fused libadn pipeline of its '
                           'UnetGenerator; call the generator, not an inner block')


class UnetGenerator(nn.Module):
    """Unet-based generator; constructed innermost -> outermost like the reference (:141-148)."""

    def __init__(self, cfg, input_nc, output_nc, num_downs, ngf=64, norm_layer=nn.BatchNorm2d, use_dropout=False):
        super().__init__()
        mk = functools.partial(UnetSkipConnectionBlock, cfg, norm_layer=norm_layer)
        block = mk(ngf * 8, ngf * 8, input_nc=None, submodule=None, innermost=True)
        for _ in range(num_downs - 5):
            block = mk(ngf * 8, ngf * 8, input_nc=None, submodule=block, use_dropout=use_dropout)
        for mult in (4, 2, 1):
            block = mk(ngf * mult, ngf * mult * 2, input_nc=None, submodule=block)
        self.model = mk(output_nc, ngf, input_nc=input_nc, submodule=block, outermost=True)
        self._num_downs = num_downs
        self._depth_norm = bool(cfg.dataset.depth_norm)
        self._engine = None
        self.compute_dtype = default_compute_dtype()

    def _adn_levels(self):
        levels, blk = [], self.model
        while blk is not None:
            parts = blk._parts()
            for bn in (parts['bn_d'], parts['bn_u']):
                if bn is not None and not isinstance(bn, nn.BatchNorm2d):
                    raise NotImplementedError('the fused libadn pipeline implements norm="batch" (the only '
                                              'setting train.py/test.py use)')
            if parts['dropout']:
                raise NotImplementedError('use_dropout=True is not on the hot path (every reference caller '
                                          'passes False)')
            levels.append({k: parts[k] for k in ('down', 'up', 'bn_d', 'bn_u')})
            blk = parts['sub']
        return levels

    def engine(self):
        if self._engine is None or self._engine.dtype != self.compute_dtype:
            object.__setattr__(self, '_engine', UNetEngine(self, self._num_downs, self._depth_norm,
                                                            self.compute_dtype))
        return self._engine

    def forward(self, input):
        """Standard forward: [B, input_nc, H, W] f32 -> [B, output_nc, H, W] f32."""
        return run_unet(self.engine(), input, self.training)
