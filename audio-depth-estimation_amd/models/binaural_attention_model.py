"""Binaural attention depth model, MI355X-native mirror of /root/reference/models/binaural_attention_model.py.

Same public surface (``DoubleConv``, ``Down``, ``Up``, ``BinauralCrossAttention``, ``BinauralEncoder``,
``BinauralAttentionDepthNet``, ``create_binaural_attention_model``), constructor arguments, attribute names
(``left_encoder``, ``right_encoder``, ``attention_modules['attn_k']``, ``fusion_layers['fusion_k']``,
``attention_levels``), state_dict keys/shapes and initialisation (reference :262-270; gamma = 0).  The
forward/backward run as an op tape on libadn (dc_engine.py): the two encoders' outputs are stacked [left; right]
along the batch so that each level's q|k|v projection, both attention directions and the gated out projection
are single launches; the N x N attention matrices are never materialised (csrc/attn.hip).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F  # noqa: F401  (the reference module exports it)
from torch.nn import init

from ..dc_engine import Act, ConvBNReLU, CrossAttention, DCEngine, Head1x1, run_dcnet
from .rgb_depth_model import DoubleConv, Down, Up, _inner  # identical copies in the reference (:22-78)
from .unetbaseline_model import default_compute_dtype


class BinauralCrossAttention(nn.Module):
    """Cross-attention between left and right channel features (reference :80-153); parameters only."""

    def __init__(self, channels, reduction=8):
        super().__init__()
        self.channels = channels
        self.reduction = reduction
        self.query = nn.Conv2d(channels, channels // reduction, kernel_size=1)
        self.key = nn.Conv2d(channels, channels // reduction, kernel_size=1)
        self.value = nn.Conv2d(channels, channels, kernel_size=1)
        self.out = nn.Conv2d(channels, channels, kernel_size=1)
        self.gamma = nn.Parameter(torch.zeros(1))

    def forward(self, left_feat, right_feat):
        _inner('BinauralCrossAttention')


class BinauralEncoder(nn.Module):
    """Encoder for a single channel (left or right) (reference :155-178)."""

    def __init__(self, base_channels=64, bilinear=True):
        super().__init__()
        self.inc = DoubleConv(1, base_channels)
        self.down1 = Down(base_channels, base_channels * 2)
        self.down2 = Down(base_channels * 2, base_channels * 4)
        self.down3 = Down(base_channels * 4, base_channels * 8)
        factor = 2 if bilinear else 1
        self.down4 = Down(base_channels * 8, base_channels * 16 // factor)

    def forward(self, x):
        _inner('BinauralEncoder')

    def adn_ops(self, x, tag, H, W):
        ops, f = self.inc.adn_ops([x], f'{tag}.x1', H, W)
        feats = [f]
        for i, down in enumerate((self.down1, self.down2, self.down3, self.down4)):
            o, f = down.adn_ops(feats[-1], f'{tag}.x{i + 2}')
            ops += o
            feats.append(f)
        return ops, feats


class BinauralAttentionDepthNet(nn.Module):
    """Binaural attention depth estimation network (reference :181-344)."""

    def __init__(self, base_channels=64, bilinear=True, output_size=256, max_depth=30.0,
                 attention_levels=[2, 3, 4, 5]):
        super().__init__()
        self.output_size = output_size
        self.max_depth = max_depth
        self.bilinear = bilinear
        self.attention_levels = attention_levels
        self.left_encoder = BinauralEncoder(base_channels, bilinear)
        self.right_encoder = BinauralEncoder(base_channels, bilinear)
        self.attention_modules = nn.ModuleDict()
        channel_map = {1: base_channels, 2: base_channels * 2, 3: base_channels * 4, 4: base_channels * 8,
                       5: base_channels * 8 if bilinear else base_channels * 16}
        for level in attention_levels:
            self.attention_modules[f'attn_{level}'] = BinauralCrossAttention(channels=channel_map[level], reduction=8)
        self.fusion_layers = nn.ModuleDict()
        for level in [1, 2, 3, 4, 5]:
            ch = channel_map[level]
            self.fusion_layers[f'fusion_{level}'] = nn.Sequential(
                nn.Conv2d(ch * 2, ch, kernel_size=1),
                nn.BatchNorm2d(ch),
                nn.ReLU(inplace=True)
            )
        factor = 2 if bilinear else 1
        self.up1 = Up(base_channels * 16, base_channels * 8 // factor, bilinear)
        self.up2 = Up(base_channels * 8, base_channels * 4 // factor, bilinear)
        self.up3 = Up(base_channels * 4, base_channels * 2 // factor, bilinear)
        self.up4 = Up(base_channels * 2, base_channels, bilinear)
        self.outc = nn.Sequential(
            nn.Conv2d(base_channels, 1, kernel_size=1),
            nn.Sigmoid()
        )
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
        if C != 2:
            raise RuntimeError(f'expected input[{B}, {C}, {H}, {W}] to have 2 channels (left, right), but got {C} '
                               'channels instead')
        left, right = eng.thin_input('left', 1, H, W), eng.thin_input('right', 1, H, W)
        ops_l, fl = self.left_encoder.adn_ops(left, 'L', H, W)
        ops_r, fr = self.right_encoder.adn_ops(right, 'R', H, W)
        ops = ops_l + ops_r
        attn_ops, fusion_ops, fused = [], [], []
        for level in range(1, 6):
            l, r = fl[level - 1], fr[level - 1]
            if level in self.attention_levels:
                eng.pair(l, r)
                lo, ro = Act(f'L.att{level}', l.C, l.H, l.W), Act(f'R.att{level}', r.C, r.H, r.W)
                eng.pair(lo, ro)
                attn_ops.append(CrossAttention(l, r, self.attention_modules[f'attn_{level}'], lo, ro))
                l, r = lo, ro
            fz = self.fusion_layers[f'fusion_{level}']
            out = Act(f'x{level}', fz[0].out_channels, l.H, l.W)
            fusion_ops.append(ConvBNReLU([l, r], fz[0], fz[1], out))            # torch.cat([left, right], dim=1)
            fused.append(out)
        # all attention blocks, then all fusion layers: a valid order of the reference's per-level loop (:303-320)
        ops += attn_ops + fusion_ops
        d = fused[4]
        for i, up in enumerate((self.up1, self.up2, self.up3, self.up4)):
            o, d = up.adn_ops(d, fused[3 - i], f'd{4 - i}')
            ops += o
        head = Head1x1(d, self.outc[0], 1, self.max_depth,                       # sigmoid * max_depth, [resize,] clamp
                       out_size=self.output_size if W != self.output_size else None)
        return [(left, 0, 1), (right, 1, 1)], ops, head

    def engine(self):
        if self._engine is None or self._engine.requested_dtype != self.compute_dtype:
            object.__setattr__(self, '_engine', DCEngine(self, self._adn_build, self.compute_dtype,
                                                         'BinauralAttentionDepthNet'))
        return self._engine

    def forward(self, x):
        """x: [B, 2, H, W] binaural spectrogram -> depth [B, 1, H, W] in [0, max_depth]."""
        return run_dcnet(self.engine(), x, self.training)

    def get_num_params(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)


def create_binaural_attention_model(base_channels=64, bilinear=True, output_size=256, max_depth=30.0,
                                    attention_levels=[2, 3, 4, 5]):
    """Factory with the reference's signature and printout (reference :347-380)."""
    model = BinauralAttentionDepthNet(base_channels=base_channels, bilinear=bilinear, output_size=output_size,
                                      max_depth=max_depth, attention_levels=attention_levels)
    print("Created Binaural Attention Model:")
    print(f"  - Base channels: {base_channels}")
    print(f"  - Attention levels: {attention_levels}")
    print(f"  - Total parameters: {model.get_num_params():,}")
    return model
