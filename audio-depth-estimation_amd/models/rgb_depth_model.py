"""RGB depth estimation model, MI355X-native mirror of /root/reference/models/rgb_depth_model.py.

Same public surface (``DoubleConv``, ``Down``, ``Up``, ``RGBDepthNet``, ``create_rgb_depth_model``), same
constructor arguments, same state_dict keys/shapes and the same initialisation (kaiming_normal fan_out for convs,
BN gamma=1 beta=0, reference :138-146); the forward/backward run as an op tape on libadn (dc_engine.py) instead
of torch ops.  The nn.Modules below only own parameters and buffers.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F  # noqa: F401  (the reference module exports it)
from torch.nn import init

from ..dc_engine import Act, ConvBNReLU, ConvT2x2, DCEngine, Head1x1, MaxPool2, Upsample2x, run_dcnet
from .unetbaseline_model import default_compute_dtype


def _inner(name):
    raise RuntimeError(f'{name} is executed by the fused libadn pipeline of its network; call the network, '
                       'not an inner block')


class DoubleConv(nn.Module):
    """(convolution => [BN] => ReLU) * 2   (reference :21-38)."""

    def __init__(self, in_channels, out_channels, mid_channels=None):
        super().__init__()
        if not mid_channels:
            mid_channels = out_channels
        self.double_conv = nn.Sequential(
            nn.Conv2d(in_channels, mid_channels, kernel_size=3, padding=1, bias=False),
            nn.BatchNorm2d(mid_channels),
            nn.ReLU(inplace=True),
            nn.Conv2d(mid_channels, out_channels, kernel_size=3, padding=1, bias=False),
            nn.BatchNorm2d(out_channels),
            nn.ReLU(inplace=True)
        )

    def forward(self, x):
        _inner('DoubleConv')

    def adn_ops(self, srcs, out_name, H, W):
        """Tape ops of this block over the (virtual concat of) ``srcs``; returns (ops, out Act)."""
        dc = self.double_conv
        mid = Act(out_name + '.mid', dc[0].out_channels, H, W)
        out = Act(out_name, dc[3].out_channels, H, W)
        return [ConvBNReLU(srcs, dc[0], dc[1], mid), ConvBNReLU([mid], dc[3], dc[4], out)], out


class Down(nn.Module):
    """Downscaling with maxpool then double conv (reference :41-52)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.maxpool_conv = nn.Sequential(
            nn.MaxPool2d(2),
            DoubleConv(in_channels, out_channels)
        )

    def forward(self, x):
        _inner('Down')

    def adn_ops(self, src, out_name):
        H, W = src.H // 2, src.W // 2
        if H < 1 or W < 1:
            raise RuntimeError(f'Given input size: ({src.C}x{src.H}x{src.W}). Calculated output size: '
                               f'({src.C}x{H}x{W}). Output size is too small')
        pooled = Act(out_name + '.pool', src.C, H, W)
        ops, out = self.maxpool_conv[1].adn_ops([pooled], out_name, H, W)
        return [MaxPool2(src, pooled)] + ops, out


class Up(nn.Module):
    """Upscaling then double conv (reference :55-77)."""

    def __init__(self, in_channels, out_channels, bilinear=True):
        super().__init__()
        if bilinear:
            self.up = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)
            self.conv = DoubleConv(in_channels, out_channels, in_channels // 2)
        else:
            self.up = nn.ConvTranspose2d(in_channels, in_channels // 2, kernel_size=2, stride=2)
            self.conv = DoubleConv(in_channels, out_channels)

    def forward(self, x1, x2):
        _inner('Up')

    def adn_ops(self, x1, x2, out_name):
        if not isinstance(self.up, nn.Upsample):                         # bilinear=False: ConvTranspose2d(k 2, s 2)
            up = Act(out_name + '.up', x1.C // 2, x2.H, x2.W)
            ops, out = self.conv.adn_ops([x2, up], out_name, x2.H, x2.W)
            return [ConvT2x2(x1, self.up, up)] + ops, out
        if x2.H < 2 * x1.H or x2.W < 2 * x1.W:
            raise NotImplementedError('negative padding (skip smaller than the upsampled tensor) cannot occur with '
                                      'MaxPool2d(2) encoders and is not implemented')
        up = Act(out_name + '.up', x1.C, x2.H, x2.W)
        ops, out = self.conv.adn_ops([x2, up], out_name, x2.H, x2.W)       # torch.cat([x2, x1], dim=1)
        return [Upsample2x(x1, up)] + ops, out


class RGBDepthNet(nn.Module):
    """RGB depth estimation network (reference :80-222); feature sizes match BinauralAttentionDepthNet."""

    FEATURES = ('x1', 'x2', 'x3', 'x4', 'x5', 'd1', 'd2', 'd3', 'd4')

    def __init__(self, base_channels=64, bilinear=True, output_size=256, max_depth=30.0):
        super().__init__()
        self.output_size = output_size
        self.max_depth = max_depth
        self.bilinear = bilinear
        self.inc = DoubleConv(3, base_channels)
        self.down1 = Down(base_channels, base_channels * 2)
        self.down2 = Down(base_channels * 2, base_channels * 4)
        self.down3 = Down(base_channels * 4, base_channels * 8)
        factor = 2 if bilinear else 1
        self.down4 = Down(base_channels * 8, base_channels * 16 // factor)
        self.up1 = Up(base_channels * 16, base_channels * 8 // factor, bilinear)
        self.up2 = Up(base_channels * 8, base_channels * 4 // factor, bilinear)
        self.up3 = Up(base_channels * 4, base_channels * 2 // factor, bilinear)
        self.up4 = Up(base_channels * 2, base_channels, bilinear)
        self.outc = nn.Conv2d(base_channels, 1, kernel_size=1)
        self._init_weights()
        self._engine = None
        self.compute_dtype = default_compute_dtype()

    def _init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
                if m.bias is not None:
                    init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                init.constant_(m.weight, 1)
                init.constant_(m.bias, 0)

    # ---- libadn tape -------------------------------------------------------------------------------
    def _adn_build(self, eng, B, C, H, W):
        if C != 3:
            raise RuntimeError(f'Given groups=1, weight of size {list(self.inc.double_conv[0].weight.shape)}, '
                               f'expected input[{B}, {C}, {H}, {W}] to have 3 channels, but got {C} channels instead')
        x = eng.thin_input('x', 3, H, W)
        ops, x1 = self.inc.adn_ops([x], 'x1', H, W)
        feats = [x1]
        for i, down in enumerate((self.down1, self.down2, self.down3, self.down4)):
            o, f = down.adn_ops(feats[-1], f'x{i + 2}')
            ops += o
            feats.append(f)
        d = feats[4]
        for i, up in enumerate((self.up1, self.up2, self.up3, self.up4)):
            o, d = up.adn_ops(d, feats[3 - i], f'd{4 - i}')
            ops += o
        # (reference :200-206: the head output is resized when its WIDTH differs from output_size)
        head = Head1x1(d, self.outc, 0, self.max_depth, out_size=self.output_size if W != self.output_size else None)
        return [(x, 0, 3)], ops, head

    def engine(self):
        if self._engine is None or self._engine.requested_dtype != self.compute_dtype:
            object.__setattr__(self, '_engine', DCEngine(self, self._adn_build, self.compute_dtype, 'RGBDepthNet'))
        return self._engine

    def forward(self, x, return_features=False):
        """x: [B, 3, H, W] RGB image -> depth [B, 1, H, W]; with return_features also the dict of x1..x5, d1..d4
        (NCHW f32 copies; detached -- distillation through them is the AdaBins model's job)."""
        eng = self.engine()
        depth = run_dcnet(eng, x, self.training)
        if return_features:
            return depth, eng.features(self.FEATURES)
        return depth

    def get_num_params(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)


def create_rgb_depth_model(base_channels=64, bilinear=True, output_size=256, max_depth=30.0):
    """Factory with the reference's signature and printout (reference :225-255)."""
    model = RGBDepthNet(base_channels=base_channels, bilinear=bilinear, output_size=output_size, max_depth=max_depth)
    print("Created RGB Depth Model:")
    print(f"  - Base channels: {base_channels}")
    print("  - Input: RGB (3 channels)")
    print(f"  - Total parameters: {model.get_num_params():,}")
    return model
