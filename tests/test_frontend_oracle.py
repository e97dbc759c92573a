"""CPU: pin oracle/frontend_oracle.py (PARITY UNPINNED w.r.t. torchaudio/torchvision, see its header) against
torch.stft / F.interpolate -- the functions those libraries delegate to -- and known-answer tests
(SURVEY.md Appendix B): frame counts, reflect-pad index table, impulse, bin-centred sinusoid, DC."""
import numpy as np
import pytest
import torch

from oracle import frontend_oracle as fo


@pytest.mark.parametrize('T,hop', [(7782, 16), (7782, 32), (3200, 16), (63 + 512, 16), (64 + 512, 16), (65 + 512, 32)])
def test_frame_count_and_torch_stft(T, hop):
    rng = np.random.default_rng(T + hop)
    x = rng.normal(size=(2, T)).astype(np.float32)
    got = fo.stft_mag(x, hop)
    assert got.shape == (2, 257, 1 + T // hop)
    ref = torch.stft(torch.from_numpy(x), n_fft=512, hop_length=hop, win_length=64,
                     window=torch.hann_window(64), center=True, pad_mode='reflect', normalized=False,
                     onesided=True, return_complex=True).abs().numpy()
    np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-4)


def test_bv2_shapes():
    assert fo.cut_samples(30.0) == 7782
    assert fo.frame_count(7782, 16) == 487 and fo.frame_count(7782, 32) == 244


def test_reflect_index_table():
    T = 100
    assert [fo.reflect_index(i, T) for i in (-3, -1, 0, 99, 100, 102)] == [3, 1, 0, 99, 98, 96]


def test_impulse_known_answer():
    T, n0, hop = 2048, 1000, 16
    x = np.zeros((1, T), dtype=np.float32)
    x[0, n0] = 1.0
    mag = fo.stft_mag(x, hop)
    w = fo.hann_periodic(64)
    for t in range(mag.shape[2]):
        j = n0 + 256 - (t * hop + 224)                 # position of the impulse inside frame t's window
        expect = w[j] if 0 <= j < 64 else 0.0
        np.testing.assert_allclose(mag[0, :, t], expect, atol=1e-9)   # |X[k]| = w[j] for every bin


def test_bin_centred_sinusoid_peaks_at_its_bin():
    T, k0 = 4096, 40
    n = np.arange(T)
    x = np.sin(2 * np.pi * k0 * n / 512)[None].astype(np.float32)
    mag = fo.stft_mag(x, 16)
    interior = mag[0, :, 20:-20]
    assert (interior.argmax(axis=0) == k0).all()


def test_dc_energy_near_bin_zero():
    mag = fo.stft_mag(np.ones((1, 2048), dtype=np.float32), 16)
    interior = mag[0, :, 20:-20]
    assert (interior.argmax(axis=0) == 0).all()
    np.testing.assert_allclose(interior[0], fo.hann_periodic(64).sum(), rtol=1e-9)
    # Hann(64) in a 512-point DFT: main lobe 2*512/64 = 16 bins wide each side, side lobes <= -31 dB and falling
    assert (np.diff(interior[:16, 0]) < 0).all()
    assert interior[17:].max() < 0.03 * interior[0].max()
    assert interior[64:].max() < 1e-3 * interior[0].max()


def test_mel_filterbank_properties():
    fb = fo.mel_fbanks()
    assert fb.shape == (257, 32) and fb.min() >= 0.0 and fb.max() <= 1.0
    assert fb[0].sum() == 0.0                       # 0 Hz is below f_min = 20 Hz
    peaks = fb.argmax(axis=0)
    assert (np.diff(peaks) >= 0).all()


@pytest.mark.parametrize('shape,antialias', [((2, 32, 244), True), ((2, 32, 244), False), ((2, 257, 487), True),
                                             ((2, 257, 487), False)])
def test_resize_matches_torch_interpolate(shape, antialias):
    rng = np.random.default_rng(1)
    x = rng.random(shape).astype(np.float32)
    got = fo.resize_bilinear(x, 256, antialias)
    ref = torch.nn.functional.interpolate(torch.from_numpy(x)[None], size=(256, 256), mode='bilinear',
                                          align_corners=False, antialias=antialias)[0].numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-5)


def test_log_minmax_branches():
    s = np.stack([np.linspace(1e-3, 5.0, 50).reshape(5, 10), np.full((5, 10), 2.0)])
    out = fo.log_minmax(s)
    assert out[0].min() == 0.0 and out[0].max() == 1.0
    assert (out[1] == 0.0).all()                    # flat channel -> zeros (BatvisionV2_Dataset.py:130-131)


def test_uncut_stft_configuration_matches_torch_stft():
    """The un-cut configuration of BatvisionV2_Dataset.py:96-99 (n_fft 400, win 200, hop 100) against torch.stft, the
    function torchaudio's Spectrogram delegates to."""
    rng = np.random.default_rng(9)
    x = rng.normal(size=(2, 5000)).astype(np.float32)
    got = fo.stft_mag(x, 100, n_fft=400, win=200)
    ref = torch.stft(torch.from_numpy(x), n_fft=400, hop_length=100, win_length=200, window=torch.hann_window(200),
                     center=True, pad_mode='reflect', normalized=False, onesided=True, return_complex=True).abs().numpy()
    assert got.shape == ref.shape == (2, 201, 51)
    np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-4)
    out = fo.bv2_audio_to_input(x, None, 64, 'mel_spectrogram', True)
    assert out.shape == (2, 64, 64) and out.min() >= 0.0 and out.max() <= 1.0
    assert fo.mel_fbanks(n_freqs=201).shape == (201, 32)


def test_cv2_linear_resize_restatement_known_answers():
    """oracle/frontend_oracle.resize_linear_cv2_u8 (OpenCV's 8-bit INTER_LINEAR integer arithmetic, parity unpinned: cv2
    is absent): identity at equal size, constants stay constant, an exact 2x reduction averages pixel pairs (round half
    up of the fixed-point sum), borders clamp."""
    from oracle import frontend_oracle as fo
    rng = np.random.default_rng(0)
    sq = rng.integers(0, 256, (16, 16, 3), dtype=np.uint8)
    np.testing.assert_array_equal(fo.resize_linear_cv2_u8(sq, 16), sq)
    np.testing.assert_array_equal(fo.resize_linear_cv2_u8(np.full((7, 9, 3), 201, np.uint8), 5), np.full((5, 5, 3), 201, np.uint8))
    ramp = np.tile((np.arange(16, dtype=np.uint8) * 10)[None, :, None], (16, 1, 1))
    half = fo.resize_linear_cv2_u8(ramp, 8)                # source position 2 dx + 0.5: mean of pixels 2dx and 2dx + 1
    np.testing.assert_array_equal(half[0, :, 0], (np.arange(8) * 20 + 5))
    up = fo.resize_linear_cv2_u8(ramp, 32)                 # 2x enlargement: first / last output columns clamp to the border
    assert up[0, 0, 0] == 0 and up[0, -1, 0] == 150 and up[0, 2, 0] == 8      # 0.75 * 10 = 7.5 -> 8
    out = fo.load_image_transform(sq, 16)
    assert out.shape == (3, 16, 16) and out.dtype == np.float32
    np.testing.assert_array_equal(out[0], sq[..., 2].astype(np.float32) / np.float32(255))      # R plane = BGR channel 2
