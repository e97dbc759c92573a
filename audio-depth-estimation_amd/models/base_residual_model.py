"""Base + Residual depth model, MI355X-native mirror of /root/reference/models/base_residual_model.py.

Same public surface (``BaseResidualDepthNet``, ``create_base_residual_model``), constructor arguments, construction
order (same-seed-same-weights) and state_dict keys; ``forward(x) -> (base_depth, residual, final_depth)``.  Shared
encoder, a narrow base decoder (hard-coded 1024/384/192/96 input channels in the reference, :113-116: the model only
exists at base_channels=64) with a sigmoid * max_depth head, a residual decoder with a tanh * 0.3 * max_depth head,
final = clamp(base + residual, 0, max_depth).  Execution: one op tape on libadn (base_residual_engine.py).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F  # noqa: F401
from torch.nn import init

from .rgb_depth_model import DoubleConv, Down, Up  # identical copies in the reference (:21-80)
from .unetbaseline_model import default_compute_dtype


class BaseResidualDepthNet(nn.Module):
    """Base + Residual depth estimation network (reference :83-231)."""

    def __init__(self, input_channels=2, base_channels=64, bilinear=True, output_size=256, max_depth=30.0):
        super().__init__()
        self.input_channels = input_channels
        self.output_size = output_size
        self.bilinear = bilinear
        self.max_depth = max_depth
        self.inc = DoubleConv(input_channels, base_channels)
        self.down1 = Down(base_channels, base_channels * 2)
        self.down2 = Down(base_channels * 2, base_channels * 4)
        self.down3 = Down(base_channels * 4, base_channels * 8)
        factor = 2 if bilinear else 1
        self.down4 = Down(base_channels * 8, base_channels * 16 // factor)
        self.base_up1 = Up(1024, 128, bilinear)
        self.base_up2 = Up(384, 64, bilinear)
        self.base_up3 = Up(192, 32, bilinear)
        self.base_up4 = Up(96, 16, bilinear)
        self.base_head = nn.Conv2d(16, 1, kernel_size=1)
        self.res_up1 = Up(base_channels * 16, base_channels * 8 // factor, bilinear)
        self.res_up2 = Up(base_channels * 8, base_channels * 4 // factor, bilinear)
        self.res_up3 = Up(base_channels * 4, base_channels * 2 // factor, bilinear)
        self.res_up4 = Up(base_channels * 2, base_channels, bilinear)
        self.res_head = nn.Conv2d(base_channels, 1, kernel_size=1)
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

    def engine(self):
        from ..base_residual_engine import BaseResidualEngine
        if self._engine is None or self._engine.requested_dtype != self.compute_dtype:
            object.__setattr__(self, '_engine', BaseResidualEngine(self, self.compute_dtype))
        return self._engine

    def forward(self, x):
        """x [B, C, H, W] -> (base_depth, residual, final_depth), each [B, 1, H, W] f32.  In training mode under autograd
        the three hang off one autograd node, so the reference's criterion(...).backward() loop works; the fused
        base_residual_engine.BaseResidualTrainer is the fast path."""
        from ..base_residual_engine import run_base_residual
        return run_base_residual(self.engine(), x, self.training)

    def get_parameters_count(self):
        cnt = lambda mods: sum(p.numel() for m in mods for p in m.parameters())
        enc = cnt([self.inc, self.down1, self.down2, self.down3, self.down4])
        bd = cnt([self.base_up1, self.base_up2, self.base_up3, self.base_up4, self.base_head])
        rd = cnt([self.res_up1, self.res_up2, self.res_up3, self.res_up4, self.res_head])
        return {'encoder': enc, 'base_decoder': bd, 'residual_decoder': rd, 'total': enc + bd + rd}


def create_base_residual_model(input_channels=2, base_channels=64, bilinear=True, output_size=256, max_depth=30.0,
                               gpu_ids=[]):
    """Factory with the reference's signature (reference :234-266); several gpu_ids mean one process per GPU here."""
    model = BaseResidualDepthNet(input_channels=input_channels, base_channels=base_channels, bilinear=bilinear,
                                 output_size=output_size, max_depth=max_depth)
    if len(gpu_ids) > 0 and torch.cuda.is_available():
        model = model.to(f'cuda:{gpu_ids[0]}')
    return model
