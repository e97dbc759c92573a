"""train.py / test.py / dataloader mirrors: host logic on CPU, end-to-end smoke on the GPU."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch


def _fake_bv2(root, n=3, T=9000, size=(20, 24)):
    import pandas as pd
    from scipy.io import wavfile
    loc = os.path.join(root, 'loc_a')
    os.makedirs(os.path.join(loc, 'audio'))
    os.makedirs(os.path.join(loc, 'depth'))
    os.makedirs(os.path.join(root, '__pycache__'))
    rng = np.random.default_rng(0)
    rows = []
    for i in range(n):
        wav = (rng.normal(size=(T, 2)) * 3000).astype(np.int16)
        wavfile.write(os.path.join(loc, 'audio', f'a{i}.wav'), 44100, wav)
        depth = (rng.random(size) * 40000 - 2000).astype(np.float32)       # mm, some < 0 and some > 30 m
        np.save(os.path.join(loc, 'depth', f'd{i}.npy'), depth)
        rows.append({'audio path': 'loc_a/audio', 'audio file name': f'a{i}.wav', 'depth path': 'loc_a/depth',
                     'depth file name': f'd{i}.npy', 'camera path': 'loc_a/cam', 'camera file name': f'c{i}.png'})
    pd.DataFrame(rows).to_csv(os.path.join(loc, 'train.csv'), index=False)
    return SimpleNamespace(dataset=SimpleNamespace(dataset_dir=root, audio_format='mel_spectrogram', preprocess='resize',
                                                   depth_norm=False, images_size=64, max_depth=30.0))


def test_train_cli_surface_and_loss_resolution():
    from audio_depth_estimation_amd import train
    from audio_depth_estimation_amd.config_loader import load_config
    p = train.build_parser()
    a = p.parse_args([])
    assert (a.dataset, a.batch_size, a.learning_rate, a.criterion, a.best_metric, a.experiment_name) == \
        ('batvisionv2', None, None, None, 'rmse', 'default')
    assert p.parse_args(['--lr', '0.01']).learning_rate == 0.01
    cfg = load_config('batvisionv2', 'train', 'default')
    assert train.resolve_loss(cfg, a) == ('Combined', 0.237, 0.637, 0.869)
    assert train.experiment_name(cfg, a) == 'unet_256_batvisionv2_BS256_Lr0.002_AdamW_default'
    cfg = load_config('batvisionv2', 'train', 'x')
    assert train.resolve_loss(cfg, p.parse_args(['--use_silog', 'false']))[:3] == ('L1', 1.0, 0.0)
    cfg = load_config('batvisionv2', 'train', 'x')
    assert train.resolve_loss(cfg, p.parse_args(['--criterion', 'SIlog', '--silog_lambda', '0.5']))[0::3] == ('SIlog', 0.5)
    cfg = load_config('batvisionv2', 'train', 'x')
    a2 = p.parse_args(['--max_depth', '80', '--eval_img'])
    assert train.experiment_name(cfg, a2).endswith('_AdamW_IMG_MD80_x')
    with pytest.raises(SystemExit):
        p.parse_args(['--optimizer', 'LAMB'])


def test_nearest_resize_is_opencv_floor_rule():
    from audio_depth_estimation_amd.dataloader.utils_dataset import resize_nearest_cv2
    d = np.arange(6 * 10, dtype=np.float32).reshape(6, 10)
    out = resize_nearest_cv2(d, 4)
    ys = [int(np.floor(i * 6 / 4)) for i in range(4)]
    xs = [int(np.floor(i * 10 / 4)) for i in range(4)]
    np.testing.assert_array_equal(out, d[np.ix_(ys, xs)])
    up = resize_nearest_cv2(d, 12)
    assert up.shape == (12, 12) and up[11, 11] == d[5, 9]


def test_bv2_dataset_raw_items(tmp_path):
    from audio_depth_estimation_amd.dataloader.BatvisionV2_Dataset import BatvisionV2Dataset
    cfg = _fake_bv2(str(tmp_path))
    ds = BatvisionV2Dataset(cfg, 'train.csv', frontend='raw')
    assert len(ds) == 3
    wave, gt = ds[1]
    assert wave.shape == (2, 7782) and wave.dtype == torch.float32 and wave.abs().max() <= 1.0   # cut to 2*30/340 s
    assert gt.shape == (1, 64, 64) and gt.dtype == torch.float32
    assert float(gt.min()) >= 0.0 and float(gt.max()) <= 30.0                                    # mm -> m, clipped
    with pytest.raises(ValueError):
        BatvisionV2Dataset(cfg, 'missing.csv')
    assert len(BatvisionV2Dataset(cfg, 'train.csv', location_blacklist=['nope'], frontend='raw')) == 3


@pytest.mark.gpu
def test_bv2_dataset_device_item_matches_oracle(tmp_path):
    from audio_depth_estimation_amd.dataloader.BatvisionV2_Dataset import BatvisionV2Dataset
    from oracle import frontend_oracle as fo
    cfg = _fake_bv2(str(tmp_path))
    raw = BatvisionV2Dataset(cfg, 'train.csv', frontend='raw')
    dev = BatvisionV2Dataset(cfg, 'train.csv', frontend='device')
    wave, gt0 = raw[0]
    item, gt1 = dev[0]
    assert item.shape == (2, 64, 64) and torch.equal(gt0, gt1)
    ref = fo.bv2_audio_to_input(wave.numpy(), 30.0, 64, 'mel_spectrogram', True)
    assert np.abs(item.numpy() - ref).max() <= 2e-3


@pytest.mark.gpu
def test_train_then_test_entrypoints_synthetic(tmp_path, monkeypatch):
    from audio_depth_estimation_amd import test as adn_test
    from audio_depth_estimation_amd import train as adn_train
    monkeypatch.chdir(tmp_path)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    model = adn_train.main(['--synthetic', '8', '--batch_size', '4', '--epochs', '10', '--validation_iter', '10',
                            '--precision', 'bf16', '--experiment_name', 'smoke'])
    exp = 'unet_256_batvisionv2_BS4_Lr0.002_AdamW_smoke'
    ck = torch.load(tmp_path / 'checkpoints' / exp / 'checkpoint_10.pth', map_location='cpu')
    assert set(ck) == {'epoch', 'state_dict', 'optimizer'} and ck['epoch'] == 10 and len(ck['state_dict']) == 82
    opt_sd = ck['optimizer']                                # torch.optim format, what the reference's optimizer writes
    assert float(opt_sd['state'][0]['step']) == 20 and len(opt_sd['param_groups'][0]['params']) == 43
    probe = torch.optim.AdamW([torch.nn.Parameter(torch.zeros_like(p, device='cpu')) for p in model.parameters()], lr=0.002)
    probe.load_state_dict(opt_sd)                           # a real torch optimizer accepts it
    assert (tmp_path / 'checkpoints' / exp / 'best_model.pth').exists()
    mean = adn_test.main(['--synthetic', '4', '--experiment_name', exp, '--checkpoints', '10', '--batch_size', '2'])
    assert len(mean) == 7 and all(np.isfinite(mean))
    stats = torch.load(tmp_path / 'eval' / 'batvisionv2' / 'test' /
                       f'stats_on_batvisionv2_test_set_{exp}_epoch_10.pt')
    assert stats['rmse'].shape == (4,) and stats['pred_imgs'].shape == (4, 256, 256)
