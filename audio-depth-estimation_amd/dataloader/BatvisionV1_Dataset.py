"""BatVision V1 dataset (mirror of the reference's dataloader/BatvisionV1_Dataset.py :13-95).

Left/right waveform .npy -> spectrogram(512/64/16) magnitude -> resize (NO log, NO min-max, :76-78); depth
.npy mm -> m, clip, nearest resize, divided by max_depth when cfg.dataset.depth_norm (:63-64).
"""
import os

import numpy as np
import pandas as pd
import torch
from torch.utils.data import Dataset

from .utils_dataset import GpuAudioFrontend, resize_nearest_cv2


class BatvisionV1Dataset(Dataset):
    def __init__(self, cfg, annotation_file, location_blacklist=None, frontend='device', antialias=True):
        self.cfg = cfg
        self.root_dir = cfg.dataset.dataset_dir
        self.audio_format = cfg.dataset.audio_format
        self.frontend = frontend
        self.antialias = antialias
        self.instances = pd.read_csv(os.path.join(self.root_dir, annotation_file))
        if location_blacklist:
            before = len(self.instances)
            for location in location_blacklist:
                self.instances = self.instances[~self.instances['audio path left'].str.contains(location)]
            print(f'BatvisionV1: Filtered {before - len(self.instances)} instances from blacklisted locations: '
                  f'{location_blacklist}')
        self._fe = None

    def __len__(self):
        return len(self.instances)

    def __getitem__(self, idx):
        instance = self.instances.iloc[idx]
        depth = np.load(os.path.join(self.root_dir, instance['depth path'])).astype(np.float32)
        depth = np.nan_to_num(depth)
        depth[np.isinf(depth)] = 0
        depth = depth / 1000
        depth[depth > self.cfg.dataset.max_depth] = self.cfg.dataset.max_depth
        depth[depth < 0.0] = 0.0
        S = self.cfg.dataset.images_size
        depth = resize_nearest_cv2(depth, S)
        if self.cfg.dataset.depth_norm:
            depth = depth / self.cfg.dataset.max_depth
        gt_depth = torch.from_numpy(np.ascontiguousarray(depth)).unsqueeze(0)
        left = np.load(os.path.join(self.root_dir, instance['audio path left'])).astype(np.float32)
        right = np.load(os.path.join(self.root_dir, instance['audio path right'])).astype(np.float32)
        waveform = torch.from_numpy(np.stack((left, right)))
        if 'waveform' in self.audio_format or self.frontend == 'raw':
            return waveform, gt_depth
        if self._fe is None:
            self._fe = GpuAudioFrontend('bv1', S, self.antialias)
        if torch.utils.data.get_worker_info() is not None:
            raise RuntimeError("frontend='device' transforms on the HIP device and cannot run inside a DataLoader worker "
                               "process (a forked worker must not touch the device): use num_workers=0, or build the "
                               "dataset with frontend='raw' and apply GpuAudioFrontend to the batch in the parent")
        dev = torch.device('cuda', torch.cuda.current_device())
        return self._fe(waveform.unsqueeze(0).to(dev))[0].cpu(), gt_depth
