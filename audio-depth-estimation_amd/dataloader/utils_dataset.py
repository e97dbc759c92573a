"""Transforms of the data pipeline (mirror of the reference's dataloader/utils_dataset.py).

``get_transform(cfg, convert=False, depth_norm=False)`` keeps the reference signature (:10-28): a composition
of (ToTensor) -> Resize((S,S)) -> (MinMaxNorm).  Resize runs in libadn (adn_resize_bilinear: bilinear,
align_corners=False, ``antialias`` explicit because the torchvision default changed across releases and the
reference pins no version -- SURVEY.md Appendix B item 4; default True).
"""
import numpy as np
import torch

from .. import kernels as K


class Resize:
    def __init__(self, size, antialias=True):
        self.size = size if isinstance(size, int) else size[0]
        self.antialias = antialias

    def __call__(self, tensor):
        dev = tensor.device if tensor.is_cuda else torch.device('cuda', torch.cuda.current_device())
        src = tensor.detach().to(dev, torch.float32).contiguous()
        out = torch.empty(src.shape[0], self.size, self.size, dtype=torch.float32, device=dev)
        K.resize_bilinear(src, self.size, self.antialias, out)
        return out if tensor.is_cuda else out.cpu()


class ToTensor:
    def __call__(self, x):
        return x if torch.is_tensor(x) else torch.from_numpy(np.ascontiguousarray(x))


class MinMaxNorm(torch.nn.Module):
    def __init__(self, min, max):
        super().__init__()
        assert isinstance(min, (float, tuple)) and isinstance(max, (float, tuple))
        self.min = torch.tensor(min)
        self.max = torch.tensor(max)

    def forward(self, tensor):
        return (tensor - self.min.to(tensor.device)) / (self.max.to(tensor.device) - self.min.to(tensor.device))


class Compose:
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


def get_transform(cfg, convert=False, depth_norm=False, antialias=True):
    transform_list = []
    if convert:
        transform_list.append(ToTensor())
    if 'resize' in cfg.dataset.preprocess:
        transform_list.append(Resize((cfg.dataset.images_size, cfg.dataset.images_size), antialias=antialias))
    if depth_norm:
        transform_list.append(MinMaxNorm(min=0.0, max=float(cfg.dataset.max_depth)))
    return Compose(transform_list)


def resize_nearest_cv2(depth, size):
    """cv2.resize(depth, (S,S), interpolation=INTER_NEAREST): src index = floor(dst * in / out)."""
    H, W = depth.shape
    ys = np.minimum((np.arange(size) * (H / size)).astype(np.int64), H - 1)
    xs = np.minimum((np.arange(size) * (W / size)).astype(np.int64), W - 1)
    return depth[ys][:, xs]


class GpuAudioFrontend:
    """Batched audio front-end on the device: raw waveforms [B,2,T] -> network input [B,2,S,S].

    The MI355X-first data path: DataLoader workers only read files (``frontend='raw'`` datasets), the
    STFT / mel / log / min-max / resize of the whole batch is ONE libadn call (adn_frontend).
    mode: 'mel_spectrogram' | 'spectrogram' (BV2: log + min-max) | 'bv1' (raw magnitude); the '_uncut' variants are the
    BV2 configuration without the max_depth cut (win 200 / n_fft 400 / hop 100, BatvisionV2_Dataset.py:96-99).
    """
    MODES = {'mel_spectrogram': 0, 'spectrogram': 1, 'bv1': 2, 'mel_spectrogram_uncut': 3, 'spectrogram_uncut': 4}

    @classmethod
    def bv2_mode(cls, audio_format, max_depth):
        """Mode name for a BV2 audio format and the dataset's max_depth (falsy = no cut)."""
        base = 'mel_spectrogram' if 'mel' in audio_format else 'spectrogram'
        return base if max_depth else base + '_uncut'

    def __init__(self, mode, size, antialias=True):
        self.mode = self.MODES[mode]
        self.size = size
        self.antialias = antialias
        self._ws = None

    def __call__(self, wave):
        wave = wave.contiguous().float()
        B, _, T = wave.shape
        need = K.frontend_workspace_bytes(B, T, self.mode) // 4
        if self._ws is None or self._ws.numel() < need or self._ws.device != wave.device:
            self._ws = torch.empty(need, dtype=torch.float32, device=wave.device)
        out = torch.empty(B, 2, self.size, self.size, dtype=torch.float32, device=wave.device)
        K.frontend(wave, self.mode, self.size, self.antialias, out, self._ws)
        return out


class GpuDepthTarget:
    """Batched depth-target preparation on the device: raw depth maps in millimetres [B,H,W]
    (float32 / uint16 / int32 device tensors) -> [B,1,S,S] f32 metres, exactly the arithmetic of
    BatvisionV2_Dataset.__getitem__ (:65-78) / BatvisionV1_Dataset (:45-64): NaN, inf -> 0; / 1000; clip to max_depth;
    negatives -> 0; cv2.INTER_NEAREST resize; / max_depth when ``depth_norm`` (BV1 :63-64).  DataLoader workers then
    only read files."""

    def __init__(self, size, max_depth, depth_norm=False):
        self.size, self.max_depth, self.depth_norm = size, max_depth, depth_norm

    def __call__(self, raw):
        from .. import kernels as K
        if not raw.is_cuda:
            raise RuntimeError('GpuDepthTarget runs on libadn HIP kernels only (no CPU path)')
        raw = raw.contiguous()
        out = torch.empty(raw.shape[0], 1, self.size, self.size, dtype=torch.float32, device=raw.device)
        K.depth_prepare(raw, self.size, self.max_depth, self.max_depth if self.depth_norm else 0.0, out)
        return out


class GpuImageTransform:
    """Batched camera-image preparation on the device (second half): decoded frames uint8
    [B,H,W,3] in OpenCV's BGR order -> [B,3,S,S] f32 RGB in [0,1], the arithmetic of BatvisionV2_Dataset._load_image
    (:199-210) after ``cv2.imread``: BGR2RGB, ``cv2.resize`` (8-bit INTER_LINEAR), / 255, HWC -> CHW.  DataLoader workers
    then only decode files."""

    def __init__(self, size):
        self.size = size

    def __call__(self, frames):
        from .. import kernels as K
        if not frames.is_cuda:
            raise RuntimeError('GpuImageTransform runs on libadn HIP kernels only (no CPU path)')
        frames = frames.contiguous()
        out = torch.empty(frames.shape[0], 3, self.size, self.size, dtype=torch.float32, device=frames.device)
        K.image_prepare(frames, self.size, out)
        return out
