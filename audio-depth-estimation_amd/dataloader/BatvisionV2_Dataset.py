"""BatVision V2 dataset (mirror of the reference's dataloader/BatvisionV2_Dataset.py).

Same constructor ``BatvisionV2Dataset(cfg, annotation_file, location_blacklist=None, use_image=False)``, CSV
columns and ``__getitem__ -> (input f32 [2|3,S,S], gt f32 [1,S,S])`` as the reference (:12-139).  The audio
branch (:94-135: cut to 2*max_depth/340 s, (mel-)spectrogram 512/64, log, per-channel min-max, resize) runs
in libadn.  ``frontend='device'`` (default) transforms each item on the HIP device like the reference does
on the CPU; ``frontend='raw'`` returns the cut waveform so that a whole batch is transformed at once by
``GpuAudioFrontend`` (what train.py uses: workers do file IO only).  File IO stays host-side Python.
"""
import os

import numpy as np
import pandas as pd
import torch
from torch.utils.data import Dataset

from .utils_dataset import GpuAudioFrontend, get_transform, resize_nearest_cv2


def load_wav(path):
    """float32 [channels, samples] in [-1,1) and the sample rate (scipy, then soundfile)."""
    try:
        from scipy.io import wavfile
        sr, data = wavfile.read(path)
        scale = {np.dtype('int16'): 32768.0, np.dtype('int32'): 2147483648.0}.get(data.dtype, 1.0)
        data = data.astype(np.float32) / scale
    except Exception as e1:
        try:
            import soundfile as sf
            data, sr = sf.read(path, dtype='float32')
        except Exception as e2:
            raise RuntimeError(f'Could not load audio file {path} (scipy: {e1}; soundfile: {e2})')
    data = data[None, :] if data.ndim == 1 else data.T
    return torch.from_numpy(np.ascontiguousarray(data)), sr


class BatvisionV2Dataset(Dataset):
    def __init__(self, cfg, annotation_file, location_blacklist=None, use_image=False, frontend='device',
                 antialias=True):
        self.cfg = cfg
        self.root_dir = cfg.dataset.dataset_dir
        self.audio_format = cfg.dataset.audio_format
        self.use_image = use_image
        self.frontend = frontend
        self.antialias = antialias
        locations = [d for d in os.listdir(self.root_dir)
                     if os.path.isdir(os.path.join(self.root_dir, d)) and not d.startswith('.')
                     and not d.startswith('__') and not d.endswith('_unzipped')]
        if location_blacklist:
            locations = [d for d in locations if d not in location_blacklist]
        frames = []
        for loc in locations:
            csv_path = os.path.join(self.root_dir, loc, annotation_file)
            if os.path.exists(csv_path):
                frames.append(pd.read_csv(csv_path))
            else:
                print(f'Warning: {csv_path} not found, skipping location {loc}')
        if not frames:
            raise ValueError(f'No valid locations found with {annotation_file} in {self.root_dir}. '
                             f'Checked {len(locations)} directories: {locations[:5]}...')
        self.instances = pd.concat(frames)
        self._fe = None

    def __len__(self):
        return len(self.instances)

    def _depth(self, instance):
        depth = np.load(os.path.join(self.root_dir, instance['depth path'], instance['depth file name']))
        depth = depth.astype(np.float32) / 1000.0                      # mm -> m
        if self.cfg.dataset.max_depth:
            depth[depth > self.cfg.dataset.max_depth] = self.cfg.dataset.max_depth
        depth[depth < 0] = 0
        S = self.cfg.dataset.images_size
        return torch.from_numpy(np.ascontiguousarray(resize_nearest_cv2(depth, S))).unsqueeze(0)

    def __getitem__(self, idx):
        instance = self.instances.iloc[idx]
        gt_depth = self._depth(instance)
        if self.use_image:
            return self._load_image(os.path.join(self.root_dir, instance['camera path'],
                                                 instance['camera file name'])), gt_depth
        waveform, sr = load_wav(os.path.join(self.root_dir, instance['audio path'], instance['audio file name']))
        if self.cfg.dataset.max_depth:
            waveform = waveform[:, :int((2 * self.cfg.dataset.max_depth / 340) * sr)]
        if 'waveform' in self.audio_format or self.frontend == 'raw':
            return waveform, gt_depth
        if self._fe is None:
            # (no max_depth: the un-cut STFT configuration of :96-99)
            mode = GpuAudioFrontend.bv2_mode(self.audio_format, self.cfg.dataset.max_depth)
            self._fe = GpuAudioFrontend(mode, self.cfg.dataset.images_size, self.antialias)
        if torch.utils.data.get_worker_info() is not None:
            raise RuntimeError("frontend='device' transforms on the HIP device and cannot run inside a DataLoader worker "
                               "process (a forked worker must not touch the device): use num_workers=0, or build the "
                               "dataset with frontend='raw' and apply GpuAudioFrontend to the batch in the parent")
        dev = torch.device('cuda', torch.cuda.current_device())
        return self._fe(waveform.unsqueeze(0).to(dev))[0].cpu(), gt_depth

    def _load_image(self, image_path):
        try:
            import cv2
        except ImportError:
            raise RuntimeError('camera images need OpenCV (cv2), which is not installed in this image')
        image = cv2.imread(image_path)
        if image is None:
            raise RuntimeError(f'Could not load image file {image_path}')
        if self.frontend == 'raw':
            # decoded frame as is (uint8 [H,W,3], BGR): colour order, resize, scaling and layout are done for the whole
            # batch on the device by utils_dataset.GpuImageTransform (frames of one dataset share their size)
            return torch.from_numpy(image)
        S = self.cfg.dataset.images_size
        image = cv2.resize(cv2.cvtColor(image, cv2.COLOR_BGR2RGB), (S, S)).astype(np.float32) / 255.0
        return torch.from_numpy(image).permute(2, 0, 1)
