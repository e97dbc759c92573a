"""GPU parity of the kernels either side of the network: audio front-end, masked loss modules, optimizer
kinds, evaluation metrics.  All through the C ABI; references = oracle/ and the reference's golden vectors."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'
GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def test_metrics_match_reference_golden():
    from audio_depth_estimation_amd.utils_criterion import compute_errors, compute_errors_batch
    z = np.load(os.path.join(GOLDEN, 'metrics_cases.npz'))
    for n in sorted({k.split('/')[0] for k in z.files}):
        got = compute_errors(z[n + '/gt'], z[n + '/pred'])
        assert len(got) == 7 and all(isinstance(v, float) for v in got)
        np.testing.assert_allclose(np.array(got), z[n + '/ref'], rtol=2e-5, atol=1e-6, err_msg=n)
    gt, pr = z['batched/gt'], z['batched/pred']
    per = compute_errors_batch(torch.from_numpy(gt).to(DEV), torch.from_numpy(pr).to(DEV)).cpu().numpy()
    from oracle.metrics_oracle import compute_errors as oracle_ce
    for b in range(gt.shape[0]):
        np.testing.assert_allclose(per[b], np.array(oracle_ce(gt[b], pr[b])), rtol=2e-5, atol=1e-6)


def test_loss_modules_match_reference_golden():
    from audio_depth_estimation_amd.utils_loss import MaskedDepthLoss, SIlogLoss
    z = np.load(os.path.join(GOLDEN, 'loss_cases.npz'))
    pred = torch.from_numpy(z['pred']).to(DEV).requires_grad_(True)
    gt = torch.from_numpy(z['gt']).to(DEV)
    m = gt != 0
    for lam in (0.5, 0.869, 1.0):
        crit = SIlogLoss(lambda_scale=lam)
        v = crit(pred[m], gt[m])                      # reference call form: gathered valid pixels
        assert abs(v.item() - float(z[f'silog_{lam}'])) <= 1e-5
        g, = torch.autograd.grad(v, pred)
        np.testing.assert_allclose(g.cpu().numpy(), z[f'silog_grad_{lam}'], rtol=2e-3, atol=1e-7)
    comb = MaskedDepthLoss('Combined', 0.237, 0.637, 0.869)
    v = comb(pred, gt)
    assert abs(v.item() - float(z['combined'])) <= 1e-5
    g, = torch.autograd.grad(v, pred)
    np.testing.assert_allclose(g.cpu().numpy(), z['combined_grad'], rtol=2e-3, atol=1e-7)
    with pytest.raises(RuntimeError):
        SIlogLoss()(torch.rand(4), torch.rand(4))      # CPU tensors: no fallback


@pytest.mark.parametrize('opt,kind,lr,wd', [('AdamW', 0, 0.002, 0.01), ('Adam', 1, 0.002, 0.0), ('SGD', 2, 0.002, 0.0),
                                            ('Adam_wd', 1, 0.001, 0.01)])
def test_optimizer_kinds_match_torch_golden(opt, kind, lr, wd):
    from audio_depth_estimation_amd import kernels as K
    z = np.load(os.path.join(GOLDEN, 'optim_cases.npz'))
    sizes = [z[f'{opt}/p0/{i}'].size for i in range(3)]
    offs = np.cumsum([0] + [(s + 3) // 4 * 4 for s in sizes])
    n = int(offs[-1])
    p = torch.zeros(n, device=DEV)
    for i in range(3):
        p[offs[i]:offs[i] + sizes[i]] = torch.from_numpy(z[f'{opt}/p0/{i}']).to(DEV)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    state = torch.zeros(8, dtype=torch.float64, device=DEV)
    ws = torch.empty(2048, dtype=torch.float64, device=DEV)
    for step in range(3):
        g = torch.zeros(n, device=DEV)
        for i in range(3):
            g[offs[i]:offs[i] + sizes[i]] = torch.from_numpy(z[f'{opt}/g{step}/{i}']).to(DEV)
        K.grad_norm(g, 1.0, state, ws)
        assert abs(state[3].item() - float(z[f'{opt}/norm{step}'])) <= 1e-5 * float(z[f'{opt}/norm{step}'])
        K.optimizer_step(p, g, m, v, kind, lr, 0.9, 0.999, 1e-8, wd, True, state)
        assert int(state[0].item()) == step + 1
        for i in range(3):
            np.testing.assert_allclose(p[offs[i]:offs[i] + sizes[i]].cpu().numpy(), z[f'{opt}/p{step + 1}/{i}'],
                                       rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize('mode,name', [(0, 'mel_spectrogram'), (1, 'spectrogram')])
@pytest.mark.parametrize('antialias', [True, False])
def test_frontend_bv2(mode, name, antialias):
    from audio_depth_estimation_amd import kernels as K
    from audio_depth_estimation_amd._lib import load
    from oracle import frontend_oracle as fo
    rng = np.random.default_rng(5)
    B, T, S = 2, 7782, 256
    wave = (0.1 * rng.normal(size=(B, 2, T))).astype(np.float32)
    out = torch.empty(B, 2, S, S, device=DEV)
    nbytes = load().adn_frontend_workspace_bytes(B, T, mode)
    ws = torch.empty(nbytes // 4, device=DEV)
    K.frontend(torch.from_numpy(wave).to(DEV), mode, S, antialias, out, ws)
    for b in range(B):
        ref = fo.bv2_audio_to_input(wave[b], 30.0, S, name, antialias)
        got = out[b].cpu().numpy()
        assert got.min() >= -1e-5 and got.max() <= 1 + 1e-5
        # log() of near-zero magnitudes amplifies f32 rounding of the DFT: compare in the normalised [0,1] range
        assert np.abs(got - ref).max() <= 2e-3, (b, np.abs(got - ref).max())
        assert np.abs(got - ref).mean() <= 1e-4


@pytest.mark.parametrize('mode,name', [(3, 'mel_spectrogram'), (4, 'spectrogram')])
def test_frontend_bv2_uncut_configuration(mode, name):
    """dataset.max_depth unset: n_fft 400 / win 200 / hop 100 (BatvisionV2_Dataset.py:96-99), modes 3 / 4."""
    from audio_depth_estimation_amd import kernels as K
    from audio_depth_estimation_amd._lib import load
    from audio_depth_estimation_amd.dataloader.utils_dataset import GpuAudioFrontend
    from oracle import frontend_oracle as fo
    rng = np.random.default_rng(15)
    B, T, S = 2, 9000, 128
    wave = (0.1 * rng.normal(size=(B, 2, T))).astype(np.float32)
    out = torch.empty(B, 2, S, S, device=DEV)
    ws = torch.empty(load().adn_frontend_workspace_bytes(B, T, mode) // 4, device=DEV)
    K.frontend(torch.from_numpy(wave).to(DEV), mode, S, True, out, ws)
    for b in range(B):
        ref = fo.bv2_audio_to_input(wave[b], None, S, name, True)
        got = out[b].cpu().numpy()
        assert np.abs(got - ref).max() <= 2e-3 and np.abs(got - ref).mean() <= 1e-4
    fe = GpuAudioFrontend(name + '_uncut', S)
    assert torch.equal(fe(torch.from_numpy(wave).to(DEV)), out)


def test_frontend_bv1_raw_magnitude():
    from audio_depth_estimation_amd import kernels as K
    from audio_depth_estimation_amd._lib import load
    from oracle import frontend_oracle as fo
    rng = np.random.default_rng(6)
    B, T, S = 1, 3200, 256
    wave = rng.normal(size=(B, 2, T)).astype(np.float32)
    out = torch.empty(B, 2, S, S, device=DEV)
    ws = torch.empty(load().adn_frontend_workspace_bytes(B, T, 2) // 4, device=DEV)
    K.frontend(torch.from_numpy(wave).to(DEV), 2, S, True, out, ws)
    ref = fo.bv1_audio_to_input(wave[0], S, True)
    got = out[0].cpu().numpy()
    assert np.abs(got - ref).max() <= 2e-4 * np.abs(ref).max()


def test_frontend_impulse_bin_indexing_exact():
    """Bit-exact bin/frame indexing: an impulse lights exactly the frames whose 64-tap window covers it."""
    from audio_depth_estimation_amd import kernels as K
    from audio_depth_estimation_amd._lib import load
    T, n0 = 3200, 1000
    wave = torch.zeros(1, 2, T)
    wave[0, :, n0] = 1.0
    # BV1 mode (raw magnitude); S chosen so that resize is the identity in time: nT = 201 frames -> use S=201? not square;
    # read the spectrogram from the workspace instead (layout documented in frontend.hip).
    nb = load().adn_frontend_workspace_bytes(1, T, 2)
    ws = torch.zeros(nb // 4, device=DEV)
    out = torch.empty(1, 2, 64, 64, device=DEV)
    K.frontend(wave.to(DEV), 2, 64, False, out, ws)
    nT = 1 + T // 16
    off_spec = 257 * 64 * 2 + 257 * 32
    spec = ws[off_spec:off_spec + 2 * 257 * nT].view(2, 257, nT).cpu().numpy()
    lit = np.nonzero(spec[0, 5] > 1e-7)[0]
    expect = [t for t in range(nT) if 0 < n0 + 256 - (t * 16 + 224) < 64]       # Hann tap j=0 is exactly 0
    assert list(lit) == expect


@pytest.mark.parametrize('dt', [torch.float32, torch.uint16, torch.int32])
@pytest.mark.parametrize('norm', [False, True])
def test_depth_target_preparation(dt, norm):
    """GpuDepthTarget == the depth branch of BatvisionV2_Dataset.__getitem__ (:65-78) / BatvisionV1 (:45-64), restated on
    the host with numpy (mm -> m, NaN/inf -> 0, clip, negatives -> 0, cv2.INTER_NEAREST index rule): bit exact."""
    from audio_depth_estimation_amd.dataloader.utils_dataset import GpuDepthTarget, resize_nearest_cv2
    rng = np.random.default_rng(0)
    B, H, W, S, maxd = 3, 720, 1280, 256, 30.0
    raw = rng.integers(0, 60000, size=(B, H, W)).astype(np.float64)
    if dt == torch.float32:
        raw[0, :5, :5] = np.nan
        raw[1, 7, 9] = np.inf
        raw[2, 100, 100] = -np.inf
        raw[0, 300:310, :] = -50.0
        src = raw.astype(np.float32)
    elif dt == torch.uint16:
        src = raw.astype(np.uint16)
    else:
        src = raw.astype(np.int32)
        src[0, 300:310, :] = -50
    want = []
    for b in range(B):
        d = np.nan_to_num(src[b].astype(np.float32))
        d[np.isinf(src[b].astype(np.float32))] = 0
        d = d / np.float32(1000.0)
        d[d > maxd] = maxd
        d[d < 0] = 0
        d = resize_nearest_cv2(d, S)
        want.append(d / np.float32(maxd) if norm else d)
    got = GpuDepthTarget(S, maxd, depth_norm=norm)(torch.from_numpy(src).to('cuda'))
    assert got.shape == (B, 1, S, S)
    np.testing.assert_array_equal(got[:, 0].cpu().numpy(), np.stack(want))


@pytest.mark.parametrize('which', ['rgb', 'binaural', 'adabins', 'baseres'])
def test_dc_trainer_entry_points_synthetic(which, tmp_path, monkeypatch):
    """train_rgb_depth / train_binaural_attention / train_adabins_distillation counterparts: two epochs on synthetic
    items, checkpoint layout of the reference scripts, resume from a checkpoint."""
    import glob
    from audio_depth_estimation_amd import train_dc
    monkeypatch.chdir(tmp_path)
    common = ['--synthetic', '8', '--batch_size', '4', '--nb_epochs', '2', '--experiment_name', 'smoke']
    if which == 'rgb':
        model = train_dc.main_rgb(common + ['--save_frequency', '1'])
        root, pat = 'checkpoints', 'epoch_0002.pth'
    elif which == 'binaural':
        model = train_dc.main_binaural(common + ['--save_frequency', '1', '--criterion', 'Combined'])
        root, pat = 'checkpoints', 'epoch_0002.pth'
    elif which == 'baseres':
        model = train_dc.main_base_residual(['--synthetic', '8', '--batch_size', '4', '--epochs', '2', '--experiment_name',
                                             'smoke', '--use_adaptive_loss'])
        root, pat = 'checkpoints', 'best_model.pth'
    else:
        model = train_dc.main_adabins(common + ['--use_adaptive_loss'])
        root, pat = 'results', 'best_model.pth'
    files = sorted(os.path.basename(f) for f in glob.glob(os.path.join(root, 'smoke', '*.pth')))
    assert pat in files and 'best_model.pth' in files, files
    ck = torch.load(os.path.join(root, 'smoke', pat), map_location='cpu')
    assert {'epoch', 'model_state_dict', 'optimizer_state_dict'} <= set(ck)
    assert set(ck['model_state_dict']) == set(model.state_dict())
    assert all(torch.isfinite(v).all() for v in ck['model_state_dict'].values() if v.is_floating_point())
    osd = ck['optimizer_state_dict']                                          # torch.optim layout, as the reference saves
    assert {'state', 'param_groups'} <= set(osd)
    # (AdaBins: the never-differentiated teacher parameters carry no state, as in torch.optim -- take any entry)
    assert float(next(iter(osd['state'].values()))['step']) == 2 * ck['epoch']     # 8 items / batch 4 = 2 steps per epoch
    if which == 'rgb':                                                         # resume: continues at epoch 3
        train_dc.main_rgb(common[:4] + ['--nb_epochs', '3', '--experiment_name', 'smoke', '--save_frequency', '1',
                                        '--checkpoints', '2'])
        assert os.path.exists(os.path.join('checkpoints', 'smoke', 'epoch_0003.pth'))


def test_edge_aware_loss_matches_reference_golden():
    """utils_binaural_attention_loss mirror (adn_edge_loss): the three terms, the total and d total / d pred against the
    fixture generated from the reference's BinauralAttentionLoss; the epoch curriculum of the adaptive variant."""
    from audio_depth_estimation_amd.utils_binaural_attention_loss import AdaptiveBinauralAttentionLoss, BinauralAttentionLoss
    z = np.load(os.path.join(GOLDEN, 'edge_loss_cases.npz'))
    for name in ('random', 'box', 'all_invalid'):
        for tag in ('default', 'heavy'):
            k = f'{name}/{tag}/'
            crit = BinauralAttentionLoss(*[float(v) for v in z[k + 'lambdas']])
            assert sorted(crit.state_dict().keys()) == ['sobel_x', 'sobel_y']
            pred = torch.from_numpy(z[name + '/pred']).to(DEV).requires_grad_(True)
            gt = torch.from_numpy(z[name + '/gt']).to(DEV)
            total, d = crit(pred, gt)
            got = np.array([d['loss_recon'], d['loss_edge'], d['loss_smooth'], d['loss_total']])
            np.testing.assert_allclose(got, z[k + 'terms'], rtol=5e-5, atol=1e-6)
            assert abs(float(total) - float(z[k + 'terms'][3])) <= 5e-5 * abs(float(z[k + 'terms'][3])) + 1e-6
            (2.0 * total).backward()                                   # upstream factor reaches the input gradient
            ref = 2.0 * z[k + 'grad']
            assert float(np.abs(pred.grad.cpu().numpy() - ref).max()) <= 1e-4 * float(np.abs(ref).max()) + 1e-8
    ad = AdaptiveBinauralAttentionLoss(warmup_epochs=20, total_epochs=200)
    pred, gt = torch.from_numpy(z['random/pred']).to(DEV), torch.from_numpy(z['random/gt']).to(DEV)
    for epoch, lr, le, ls, tot in z['adaptive']:
        _, d = ad(pred, gt, int(epoch))
        np.testing.assert_allclose([d['lambda_recon'], d['lambda_edge'], d['lambda_smooth']], [lr, le, ls], rtol=1e-6, atol=1e-9)
        assert abs(d['loss_total'] - tot) <= 5e-5 * abs(tot) + 1e-6
    with pytest.raises(RuntimeError):
        BinauralAttentionLoss()(torch.zeros(1, 1, 8, 8), torch.zeros(1, 1, 8, 8))


@pytest.mark.parametrize('H,W,S', [(480, 640, 256), (100, 60, 128), (256, 256, 256), (123, 77, 64)])
def test_camera_image_preparation_bit_exact(H, W, S):
    """adn_image_prepare / GpuImageTransform (BatvisionV2_Dataset._load_image :199-210 after cv2.imread) against the host
    restatement of OpenCV's 8-bit INTER_LINEAR arithmetic: bit-exact (integer work; the final / 255 is one f32 division)."""
    from audio_depth_estimation_amd.dataloader.utils_dataset import GpuImageTransform
    from oracle import frontend_oracle as fo
    rng = np.random.default_rng(H * 1000 + W)
    frames = rng.integers(0, 256, (3, H, W, 3), dtype=np.uint8)
    got = GpuImageTransform(S)(torch.from_numpy(frames).to('cuda'))
    assert got.shape == (3, 3, S, S) and got.dtype == torch.float32
    want = np.stack([fo.load_image_transform(f, S) for f in frames])
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    with pytest.raises(RuntimeError):
        GpuImageTransform(S)(torch.from_numpy(frames))


def test_binaural_entry_point_on_disk_dataset_with_worker_processes(tmp_path, monkeypatch):
    """on a REAL dataset the DataLoader workers are forked after the parent has initialised HIP and must
    never touch the device.  A tiny on-disk BatVision-V2-shaped dataset (wav + npy + csv), ``--num_workers 2``: the
    workers only read files (``frontend='raw'``), the STFT / mel / resize of the whole batch runs in the parent
    (GpuAudioFrontend inside the step and the validation forward)."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from test_entrypoints import _fake_bv2
    from audio_depth_estimation_amd import train_dc
    cfg = _fake_bv2(str(tmp_path / 'data'), n=6)
    cfg.dataset.name = 'batvisionv2'
    cfg.dataset.annotation_file_train = cfg.dataset.annotation_file_val = 'train.csv'
    cfg.dataset.location_blacklist = None
    monkeypatch.setattr(train_dc, 'load_config', lambda **kw: cfg)
    monkeypatch.chdir(tmp_path)
    model = train_dc.main_binaural(['--batch_size', '2', '--nb_epochs', '2', '--num_workers', '2', '--base_channels', '8',
                                    '--experiment_name', 'disk', '--save_frequency', '1'])
    ck = torch.load(os.path.join('checkpoints', 'disk', 'epoch_0002.pth'), map_location='cpu')
    assert ck['epoch'] == 2 and float(next(iter(ck['optimizer_state_dict']['state'].values()))['step']) == 6   # 6 items / 2
    assert all(torch.isfinite(v).all() for v in ck['model_state_dict'].values() if v.is_floating_point())
    assert set(ck['model_state_dict']) == set(model.state_dict())


def test_debug_entry_points():
    """adn_debug_stream_rmw (the overlap experiment's stand-in for the all-reduce's local traffic: dst += src as uint32, any
    workgroup count, `passes` times) and adn_debug_poison_lds (must leave results of the next launch untouched)."""
    import ctypes as C
    from audio_depth_estimation_amd import _lib
    lib = _lib.load()
    n = 4 * 1000 + 4
    src = torch.arange(n, dtype=torch.int32, device='cuda')
    dst = torch.full((n,), 7, dtype=torch.int32, device='cuda')
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for wgs, passes in ((1, 1), (16, 3), (300, 2)):
        before = dst.clone()
        _lib.check(lib.adn_debug_stream_rmw(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), n * 4, wgs, passes, st),
                   'adn_debug_stream_rmw')
        assert torch.equal(dst, before + passes * src)
    assert lib.adn_debug_stream_rmw(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), 10, 1, 1, st) != 0     # not 16-byte sized
    _lib.check(lib.adn_debug_poison_lds(st), 'adn_debug_poison_lds')
    a = torch.rand(3, 5, device='cuda')
    assert torch.equal(a + 0, a)
    torch.cuda.synchronize()
