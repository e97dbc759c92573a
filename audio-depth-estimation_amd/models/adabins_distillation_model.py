"""AdaBins with knowledge distillation (RGB teacher -> audio student), MI355X-native mirror of
/root/reference/models/adabins_distillation_model.py.

Same public surface (``AdaBinsEncoder``, ``AdaBinsBinPredictor``, ``AdaBinsDecoder``, ``AdaBinsDistillationModel``,
``create_adabins_distillation_model``), constructor arguments, construction order (same-seed-same-weights; 230
state_dict keys), ``forward(audio, rgb=None, mode='train')`` -> ``{'audio': {...}, 'rgb': {...} | None}`` with the
keys ``features, bin_centers, bin_widths, bin_logits, base_depth, residual, final_depth``, ``freeze_rgb()``,
``get_parameters_count()``.  Like the reference's decoder (:186-189 hard-codes 1024/768/384/192 input channels) the
model only exists at ``base_channels=64``.

Execution (adabins_engine.py): each branch is an op tape on libadn.  The reference evaluates the decoder twice on
identical inputs (:322-330 / :369-377, "for simplicity"); the second pass reproduces the first one's activations, so
the tape runs it once, feeds both heads (class head + residual head) from the same tensor, sums their gradients
and replays only the side effect: a second BatchNorm running-statistics update per decoder BN in train mode.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F  # noqa: F401
from torch.nn import init

from .rgb_depth_model import DoubleConv, Down, Up, _inner  # identical copies in the reference (:27-82)
from .unetbaseline_model import default_compute_dtype


class AdaBinsEncoder(nn.Module):
    """Encoder for AdaBins (RGB or audio) (reference :84-103)."""

    def __init__(self, input_channels, base_channels=64):
        super().__init__()
        self.inc = DoubleConv(input_channels, base_channels)
        self.down1 = Down(base_channels, base_channels * 2)
        self.down2 = Down(base_channels * 2, base_channels * 4)
        self.down3 = Down(base_channels * 4, base_channels * 8)
        self.down4 = Down(base_channels * 8, base_channels * 8)

    def forward(self, x):
        _inner('AdaBinsEncoder')

    def adn_ops(self, x, tag, H, W):
        ops, f = self.inc.adn_ops([x], f'{tag}.x1', H, W)
        feats = [f]
        for i, down in enumerate((self.down1, self.down2, self.down3, self.down4)):
            o, f = down.adn_ops(feats[-1], f'{tag}.x{i + 2}')
            ops += o
            feats.append(f)
        return ops, feats


class AdaBinsBinPredictor(nn.Module):
    """Predicts adaptive bin centres from global features (reference :105-149); parameters only."""

    def __init__(self, bottleneck_dim=512, n_bins=128, max_depth=30.0):
        super().__init__()
        self.n_bins = n_bins
        self.max_depth = max_depth
        self.adaptive_pool = nn.AdaptiveAvgPool2d(1)
        self.predictor = nn.Sequential(
            nn.Linear(bottleneck_dim, 256),
            nn.ReLU(inplace=True),
            nn.Dropout(0.1),
            nn.Linear(256, n_bins),
            nn.Softmax(dim=1)
        )

    def forward(self, features):
        _inner('AdaBinsBinPredictor')


class AdaBinsDecoder(nn.Module):
    """Decoder that predicts the per-pixel bin classification (reference :152-207)."""

    def __init__(self, base_channels=64, n_bins=128, output_size=256):
        super().__init__()
        self.n_bins = n_bins
        self.output_size = output_size
        self.up1 = Up(1024, base_channels * 8, bilinear=True)
        self.up2 = Up(768, base_channels * 4, bilinear=True)
        self.up3 = Up(384, base_channels * 2, bilinear=True)
        self.up4 = Up(192, base_channels, bilinear=True)
        self.class_head = nn.Conv2d(base_channels, n_bins, kernel_size=1)

    def forward(self, features, bin_centers):
        _inner('AdaBinsDecoder')

    def adn_ops(self, feats, tag):
        ops, d = [], feats[4]
        for i, up in enumerate((self.up1, self.up2, self.up3, self.up4)):
            o, d = up.adn_ops(d, feats[3 - i], f'{tag}.d{4 - i}')
            ops += o
        return ops, d


class AdaBinsDistillationModel(nn.Module):
    """AdaBins with knowledge distillation from RGB to audio (reference :210-459)."""

    def __init__(self, n_bins=128, base_channels=64, output_size=256, max_depth=30.0, use_pretrained_rgb=False):
        super().__init__()
        self.n_bins = n_bins
        self.max_depth = max_depth
        self.output_size = output_size
        self.rgb_encoder = AdaBinsEncoder(input_channels=3, base_channels=base_channels)
        self.rgb_bin_predictor = AdaBinsBinPredictor(bottleneck_dim=base_channels * 8, n_bins=n_bins, max_depth=max_depth)
        self.rgb_decoder = AdaBinsDecoder(base_channels=base_channels, n_bins=n_bins, output_size=output_size)
        if use_pretrained_rgb:
            self._load_pretrained_rgb()
        self.audio_encoder = AdaBinsEncoder(input_channels=2, base_channels=base_channels)
        self.audio_bin_predictor = AdaBinsBinPredictor(bottleneck_dim=base_channels * 8, n_bins=n_bins,
                                                       max_depth=max_depth)
        self.audio_decoder = AdaBinsDecoder(base_channels=base_channels, n_bins=n_bins, output_size=output_size)
        self.residual_head = nn.Conv2d(base_channels, 1, kernel_size=1)
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

    def _load_pretrained_rgb(self):
        print("INFO: Placeholder for loading pre-trained RGB encoder")

    def engine(self):
        from ..adabins_engine import AdaBinsEngine
        if self._engine is None or self._engine.requested_dtype != self.compute_dtype:
            object.__setattr__(self, '_engine', AdaBinsEngine(self, self.compute_dtype))
        return self._engine

    def forward_audio(self, audio):
        """Student branch (reference :353-399): dict of NCHW f32 tensors.  In training mode with autograd enabled the
        tensors hang off one autograd node (adabins_engine._StudentFunction), so a loss built from them back-propagates
        into the student's parameters; otherwise they are plain values."""
        from ..adabins_engine import run_student
        return run_student(self.engine(), audio, self.training)

    def forward_rgb(self, rgb):
        """Teacher branch (reference :301-351)."""
        return self.engine().run_branch('rgb', rgb, self.training)

    def forward(self, audio, rgb=None, mode='train'):
        """audio [B,2,H,W] (+ rgb [B,3,H,W] in training) -> {'audio': {...}, 'rgb': {...} | None} (reference :401-426).
        The reference's loop (criterion(outputs, gt, mask)[0].backward(); clip; optimizer.step()) works on the returned
        tensors; ``adabins_engine.AdaBinsTrainer`` (fused DistillationLoss + backward + clip + AdamW) is the fast
        replacement of train_adabins_distillation.py:445-456."""
        audio_output = self.forward_audio(audio)
        rgb_output = self.forward_rgb(rgb) if (mode == 'train' and rgb is not None) else None
        return {'audio': audio_output, 'rgb': rgb_output}

    def freeze_rgb(self):
        for part in (self.rgb_encoder, self.rgb_bin_predictor, self.rgb_decoder):
            for param in part.parameters():
                param.requires_grad = False
        print("RGB teacher frozen")

    def get_parameters_count(self):
        cnt = lambda *mods: sum(p.numel() for m in mods for p in m.parameters())
        rgb_params = cnt(self.rgb_encoder, self.rgb_bin_predictor, self.rgb_decoder)
        audio_params = cnt(self.audio_encoder, self.audio_bin_predictor, self.audio_decoder)
        residual_params = cnt(self.residual_head)
        return {'rgb_teacher': rgb_params, 'audio_student': audio_params, 'residual': residual_params,
                'total': rgb_params + audio_params + residual_params}


def create_adabins_distillation_model(n_bins=128, base_channels=64, output_size=256, max_depth=30.0,
                                      use_pretrained_rgb=False, gpu_ids=[]):
    """Factory with the reference's signature (reference :462-494).  More than one GPU means one process per GPU
    with the RCCL gradient reducer (ddp.py), not nn.DataParallel: with several ids the model goes to the first."""
    model = AdaBinsDistillationModel(n_bins=n_bins, base_channels=base_channels, output_size=output_size,
                                     max_depth=max_depth, use_pretrained_rgb=use_pretrained_rgb)
    if len(gpu_ids) > 0 and torch.cuda.is_available():
        model = model.to(f'cuda:{gpu_ids[0]}')
    return model
