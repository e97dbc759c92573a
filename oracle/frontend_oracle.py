"""CPU restatement of the audio front-end (TEST INFRASTRUCTURE ONLY).  PARITY UNPINNED.

Restates the audio branch of
  * BatvisionV2Dataset.__getitem__       /root/reference/dataloader/BatvisionV2_Dataset.py:94-135
  * _get_spectrogram / _get_melspectrogram                                       :177-197
  * BatvisionV1Dataset audio branch      /root/reference/dataloader/BatvisionV1_Dataset.py:68-83,86-95
  * get_transform -> transforms.Resize   /root/reference/dataloader/utils_dataset.py:10-28
The arithmetic lives in third-party dependencies that are absent from /root/reference AND
from this image (torchaudio.transforms.Spectrogram / MelSpectrogram, torchvision
transforms.Resize; no version pinned anywhere in the reference).  This file restates their
published algorithms following SURVEY.md Appendix B, as a direct windowed DFT in float64
numpy.  It is pinned only against torch.stft / F.interpolate (the functions those libraries
delegate to) and against known-answer tests (impulse, bin-centred sinusoid, DC, frame
counts, reflect-pad index table) in tests/test_frontend_oracle.py -- hence "parity unpinned"
with respect to torchaudio/torchvision themselves.
"""
from __future__ import annotations

import numpy as np

N_FFT = 512
WIN = 64
SR = 44100


def cut_samples(max_depth: float, sr: int = SR) -> int:
    """BatvisionV2_Dataset.py:102-104: int((2*max_depth/340)*sr)."""
    return int((2 * max_depth / 340) * sr)


def hann_periodic(n: int = WIN) -> np.ndarray:
    """torch.hann_window(n) default periodic=True."""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def reflect_index(i: int, T: int) -> int:
    """Index into x for padded position i in [-pad, T+pad): reflect without edge repeat."""
    if i < 0:
        return -i
    if i >= T:
        return 2 * (T - 1) - i
    return i


def frame_count(T: int, hop: int) -> int:
    return 1 + T // hop


def stft_mag(x: np.ndarray, hop: int, n_fft: int = N_FFT, win: int = WIN) -> np.ndarray:
    """|STFT| with centre=True, reflect pad, periodic Hann(win) centred in n_fft, onesided.

    x [C, T] -> [C, n_fft//2+1, 1+T//hop].  Only ``win`` taps of each frame are non-zero:
    X[k,t] = sum_j w[j] x_pad[t*hop + off + j] exp(-2 pi i k (off+j)/n_fft), off=(n_fft-win)//2.
    """
    x = np.asarray(x, dtype=np.float64)
    C, T = x.shape
    pad = n_fft // 2
    off = (n_fft - win) // 2
    nT = frame_count(T, hop)
    w = hann_periodic(win)
    idx = np.empty((nT, win), dtype=np.int64)
    for t in range(nT):
        for j in range(win):
            idx[t, j] = reflect_index(t * hop + off + j - pad, T)
    frames = x[:, idx] * w[None, None, :]                       # [C, nT, win]
    k = np.arange(n_fft // 2 + 1)[:, None]
    n = (off + np.arange(win))[None, :]
    ang = -2.0 * np.pi * ((k * n) % n_fft) / n_fft              # exact integer phase reduction
    basis = np.cos(ang) + 1j * np.sin(ang)                      # [F, win]
    X = np.einsum('ctj,fj->cft', frames, basis)
    return np.abs(X)


def _hz_to_mel_htk(f):
    return 2595.0 * np.log10(1.0 + f / 700.0)


def _mel_to_hz_htk(m):
    return 700.0 * (10.0 ** (m / 2595.0) - 1.0)


def mel_fbanks(n_freqs=N_FFT // 2 + 1, f_min=20.0, f_max=20000.0, n_mels=32, sr=SR) -> np.ndarray:
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale='htk') -> [n_freqs, n_mels]."""
    all_freqs = np.linspace(0.0, sr // 2, n_freqs)
    m_pts = np.linspace(_hz_to_mel_htk(f_min), _hz_to_mel_htk(f_max), n_mels + 2)
    f_pts = _mel_to_hz_htk(m_pts)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return np.maximum(0.0, np.minimum(down, up))


def log_minmax(spec: np.ndarray) -> np.ndarray:
    """BatvisionV2_Dataset.py:122-132: log(x+1e-8), per-channel min-max (zeros if flat)."""
    s = np.log(spec + 1e-8)
    out = np.empty_like(s)
    for c in range(s.shape[0]):
        lo, hi = s[c].min(), s[c].max()
        out[c] = (s[c] - lo) / (hi - lo) if hi > lo else 0.0
    return out


def _aa_weights(in_size, out_size, antialias):
    """Separable bilinear (triangle) resampling weights, align_corners=False.

    antialias=False: 2-tap lerp at src=(dst+0.5)*scale-0.5 clamped to >=0 (ATen upsample_bilinear2d).
    antialias=True : ATen _upsample_bilinear2d_aa: support = max(scale,1), triangle filter
    evaluated at (j + xmin - center + 0.5)/max(scale,1), normalised per output sample.
    """
    scale = in_size / out_size
    W = np.zeros((out_size, in_size), dtype=np.float64)
    if not antialias or scale <= 1.0:
        for d in range(out_size):
            src = max((d + 0.5) * scale - 0.5, 0.0)
            i0 = min(int(np.floor(src)), in_size - 1)
            i1 = min(i0 + 1, in_size - 1)
            l1 = src - i0
            W[d, i0] += 1.0 - l1
            W[d, i1] += l1
        return W
    support = scale
    for d in range(out_size):
        center = scale * (d + 0.5)
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size)
        ws = []
        for j in range(xmax - xmin):
            a = abs((j + xmin - center + 0.5) / scale)
            ws.append(max(0.0, 1.0 - a))
        tot = sum(ws)
        for j, wv in enumerate(ws):
            W[d, xmin + j] = wv / tot
    return W


def resize_bilinear(x: np.ndarray, size: int, antialias: bool = True) -> np.ndarray:
    """transforms.Resize((S,S)) on a [C,H,W] tensor (utils_dataset.py:18-20)."""
    C, H, W = x.shape
    Wy = _aa_weights(H, size, antialias)
    Wx = _aa_weights(W, size, antialias)
    return np.einsum('yh,chw,xw->cyx', Wy, x.astype(np.float64), Wx)


def bv2_audio_to_input(wave: np.ndarray, max_depth=30.0, images_size=256,
                       audio_format='mel_spectrogram', antialias=True) -> np.ndarray:
    """Full BV2 audio branch: cut -> (mel)spectrogram -> log -> minmax -> resize.  ``max_depth`` falsy = the un-cut
    configuration of :96-99 (win_length 200, n_fft 400, hop 100; the mel path's default hop win // 2 is 100 too)."""
    if max_depth:
        w = wave[:, :cut_samples(max_depth)]
        n_fft, win, hop_lin = N_FFT, WIN, WIN // 4               # :105-108
    else:
        w = wave
        n_fft, win, hop_lin = 400, 200, 100                      # :96-99
    if 'mel' in audio_format:
        spec = stft_mag(w, hop=win // 2, n_fft=n_fft, win=win)   # hop_length not passed -> win//2 (:187-197)
        spec = np.einsum('cft,fm->cmt', spec, mel_fbanks(n_freqs=n_fft // 2 + 1))
    else:
        spec = stft_mag(w, hop=hop_lin, n_fft=n_fft, win=win)    # (:108,:118)
    return resize_bilinear(log_minmax(spec), images_size, antialias)


def bv1_audio_to_input(wave: np.ndarray, images_size=256, antialias=True) -> np.ndarray:
    """BV1 audio branch: spectrogram(512/64/16) -> resize; no log, no min-max (:76-78)."""
    return resize_bilinear(stft_mag(wave, hop=WIN // 4), images_size, antialias)


def resize_linear_cv2_u8(img, S):
    """cv2.resize(img, (S, S)) for uint8 images, INTER_LINEAR: the integer arithmetic of OpenCV's 8-bit path restated
    from its published source (modules/imgproc/src/resize.cpp: 11-bit coefficients INTER_RESIZE_COEF_BITS,
    HResizeLinear in int, the uchar specialisation of VResizeLinear).  img [H,W,C] uint8 -> [S,S,C] uint8.
    PARITY UNPINNED against OpenCV (cv2 is not installed in this image): pinned by known answers only."""
    img = np.asarray(img, dtype=np.uint8)
    H, W = img.shape[:2]

    def coef(n_out, n_in):
        d = np.arange(n_out)
        f = ((d + 0.5) * (float(n_in) / n_out) - 0.5).astype(np.float32)
        s0 = np.floor(f).astype(np.int64)
        a = (f - s0).astype(np.float32)
        lo, hi = s0 < 0, s0 >= n_in - 1
        s0 = np.where(lo, 0, np.where(hi, n_in - 1, s0))
        a = np.where(lo | hi, np.float32(0), a)
        c1 = np.rint(a * np.float32(2048)).astype(np.int64)            # cvRound: half to even
        return s0, np.minimum(s0 + 1, n_in - 1), 2048 - c1, c1

    y0, y1, b0, b1 = coef(S, H)
    x0, x1, a0, a1 = coef(S, W)
    src = img.astype(np.int64)
    rows = src[:, x0] * a0[None, :, None] + src[:, x1] * a1[None, :, None]          # [H,S,C] horizontal pass
    d0, d1 = rows[y0], rows[y1]
    v = (((b0[:, None, None] * (d0 >> 4)) >> 16) + ((b1[:, None, None] * (d1 >> 4)) >> 16) + 2) >> 2
    return v.astype(np.uint8)


def load_image_transform(frame_bgr, S):
    """BatvisionV2_Dataset._load_image (:199-210) after cv2.imread: BGR2RGB, resize, / 255, HWC -> CHW (float32)."""
    rgb = resize_linear_cv2_u8(np.asarray(frame_bgr)[..., ::-1], S)
    return np.ascontiguousarray((rgb.astype(np.float32) / np.float32(255.0)).transpose(2, 0, 1))
