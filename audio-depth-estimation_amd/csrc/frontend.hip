// Audio front-end on device: centred reflect-padded STFT magnitude as a WIN-tap windowed DFT (periodic Hann), optional
// HTK mel projection (-> 32), log + per-channel min-max, bilinear resize (align_corners=False, optional antialias) to
// S x S.  Two STFT configurations, as in the reference (BatvisionV2_Dataset.py:96-109): the cut one (audio cut to
// 2*max_depth/340 s: n_fft 512, win 64, hop 16, mel hop 32) and the un-cut one (dataset.max_depth unset: n_fft 400,
// win 200, hop 100 for both formats).
// Follows SURVEY.md Appendix B; reference call sites: dataloader/BatvisionV2_Dataset.py:94-135,
// :177-197, dataloader/BatvisionV1_Dataset.py:68-95, dataloader/utils_dataset.py:18-20.
//
// Workspace layout (floats): basis_cos[257*64] | basis_sin[257*64] | fb[257*32] |
//                            spec[B*2*F*nT] | minmax partials [B*2*tiles*2]
#include "adn_common.h"

namespace {

constexpr int NMEL = 32;
constexpr int FT = 8;  // frames per block

template <int NFFT, int WIN>
__global__ __launch_bounds__(256) void fe_tables_kernel(float* bcos, float* bsin, float* fb) {
  constexpr int NBIN = NFFT / 2 + 1, OFF = (NFFT - WIN) / 2;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx < NBIN * WIN) {
    const int k = idx / WIN, j = idx % WIN;
    const int ph = (k * (OFF + j)) % NFFT;                        // exact integer phase reduction
    const double w = 0.5 - 0.5 * cospi(2.0 * (double)j / WIN);    // periodic Hann
    bcos[idx] = (float)(w * cospi(2.0 * ph / (double)NFFT));
    bsin[idx] = (float)(-w * sinpi(2.0 * ph / (double)NFFT));
  }
  if (idx < NBIN * NMEL) {
    const int f = idx / NMEL, m = idx % NMEL;
    auto hz2mel = [](double h) { return 2595.0 * log10(1.0 + h / 700.0); };
    auto mel2hz = [](double x) { return 700.0 * (pow(10.0, x / 2595.0) - 1.0); };
    const double mlo = hz2mel(20.0), mhi = hz2mel(20000.0);
    const double step = (mhi - mlo) / (NMEL + 1);
    const double f0 = mel2hz(mlo + step * m), f1 = mel2hz(mlo + step * (m + 1)), f2 = mel2hz(mlo + step * (m + 2));
    const double freq = (double)f * (44100 / 2) / (NBIN - 1);
    const double down = (freq - f0) / (f1 - f0), up = (f2 - freq) / (f2 - f1);
    const double v = fmin(down, up);
    fb[idx] = (float)(v > 0.0 ? v : 0.0);
  }
}

__device__ __forceinline__ int reflect(int i, int T) { return i < 0 ? -i : (i >= T ? 2 * (T - 1) - i : i); }

// grid (frame tiles, B*2).  mode 0: mel+log, 1: linear+log, 2: linear raw.
template <int NFFT, int WIN>
__global__ __launch_bounds__(256) void fe_stft_kernel(const float* wave, int T, int hop, int nT, int mode,
                                                      const float* bcos, const float* bsin, const float* fb,
                                                      float* spec, float* mm_part, int tiles) {
  constexpr int NBIN = NFFT / 2 + 1, PAD = NFFT / 2, OFF = (NFFT - WIN) / 2;
  __shared__ float fr[FT][WIN];
  __shared__ float mag[FT][NBIN + 3];
  __shared__ float red[8];
  const int bc = blockIdx.y;
  const int t0 = blockIdx.x * FT;
  const float* x = wave + (int64_t)bc * T;
  const int tid = threadIdx.x;
  for (int e = tid; e < FT * WIN; e += 256) {
    const int f = e / WIN, j = e % WIN;
    const int t = t0 + f;
    fr[f][j] = t < nT ? x[reflect(t * hop + OFF + j - PAD, T)] : 0.f;
  }
  __syncthreads();
  for (int k = tid; k < NBIN; k += 256) {
    float re[FT], im[FT];
#pragma unroll
    for (int f = 0; f < FT; ++f) re[f] = im[f] = 0.f;
    for (int j = 0; j < WIN; ++j) {
      const float c = bcos[k * WIN + j], s = bsin[k * WIN + j];
#pragma unroll
      for (int f = 0; f < FT; ++f) {
        re[f] += fr[f][j] * c;
        im[f] += fr[f][j] * s;
      }
    }
#pragma unroll
    for (int f = 0; f < FT; ++f) mag[f][k] = sqrtf(re[f] * re[f] + im[f] * im[f]);
  }
  __syncthreads();
  const int F = mode == 0 ? NMEL : NBIN;
  float lo = INFINITY, hi = -INFINITY;
  for (int e = tid; e < F * FT; e += 256) {
    const int f = e % FT, r = e / FT;
    const int t = t0 + f;
    if (t >= nT) continue;
    float v;
    if (mode == 0) {
      v = 0.f;
      for (int k = 0; k < NBIN; ++k) v += mag[f][k] * fb[k * NMEL + r];
    } else {
      v = mag[f][r];
    }
    if (mode != 2) v = logf(v + 1e-8f);
    lo = fminf(lo, v);
    hi = fmaxf(hi, v);
    spec[((int64_t)bc * F + r) * nT + t] = v;
  }
  lo = wave_min(lo);
  hi = wave_max(hi);
  if ((tid & 63) == 0) {
    red[(tid >> 6) * 2] = lo;
    red[(tid >> 6) * 2 + 1] = hi;
  }
  __syncthreads();
  if (tid == 0) {
    mm_part[((int64_t)bc * tiles + blockIdx.x) * 2 + 0] = fminf(fminf(red[0], red[2]), fminf(red[4], red[6]));
    mm_part[((int64_t)bc * tiles + blockIdx.x) * 2 + 1] = fmaxf(fmaxf(red[1], red[3]), fmaxf(red[5], red[7]));
  }
}

// separable triangle-filter weights of one output coordinate (ATen upsample_bilinear2d /
// _upsample_bilinear2d_aa, align_corners=False)
struct Taps {
  int start, count;
  float scale_inv, center, total;
};
__device__ __forceinline__ float tap_w(const Taps& t, int j, bool aa, float lerp) {
  if (!aa) return j == 0 ? 1.0f - lerp : lerp;
  const float a = fabsf(((float)(j + t.start) - t.center + 0.5f) * t.scale_inv);
  return a < 1.0f ? 1.0f - a : 0.0f;
}

__global__ __launch_bounds__(256) void fe_resize_kernel(const float* spec, const float* mm_part, int tiles, int F,
                                                        int nT, int S, int antialias, int normalise, float* out) {
  const int bc = blockIdx.y;
  __shared__ float s_lo, s_hi;
  if (threadIdx.x == 0) {
    float lo = INFINITY, hi = -INFINITY;
    for (int i = 0; i < tiles; ++i) {
      lo = fminf(lo, mm_part[((int64_t)bc * tiles + i) * 2]);
      hi = fmaxf(hi, mm_part[((int64_t)bc * tiles + i) * 2 + 1]);
    }
    s_lo = lo;
    s_hi = hi;
  }
  __syncthreads();
  const float lo = s_lo, hi = s_hi;
  const bool flat = !(hi > lo);
  const float inv = flat ? 0.f : 1.0f / (hi - lo);
  const float* src = spec + (int64_t)bc * F * nT;
  const float sy = (float)F / S, sx = (float)nT / S;
  const bool aay = antialias && sy > 1.0f, aax = antialias && sx > 1.0f;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < S * S; e += gridDim.x * 256) {
    const int oy = e / S, ox = e % S;
    Taps ty, tx;
    float ly = 0.f, lx = 0.f;
    if (aay) {
      const float c = sy * (oy + 0.5f);
      ty.center = c;
      ty.scale_inv = 1.0f / sy;
      ty.start = max((int)(c - sy + 0.5f), 0);
      ty.count = min((int)(c + sy + 0.5f), F) - ty.start;
    } else {
      const float s = fmaxf((oy + 0.5f) * sy - 0.5f, 0.f);
      const int i0 = min((int)s, F - 1);
      ty.start = i0;
      ty.count = 2;
      ly = s - i0;
    }
    if (aax) {
      const float c = sx * (ox + 0.5f);
      tx.center = c;
      tx.scale_inv = 1.0f / sx;
      tx.start = max((int)(c - sx + 0.5f), 0);
      tx.count = min((int)(c + sx + 0.5f), nT) - tx.start;
    } else {
      const float s = fmaxf((ox + 0.5f) * sx - 0.5f, 0.f);
      const int i0 = min((int)s, nT - 1);
      tx.start = i0;
      tx.count = 2;
      lx = s - i0;
    }
    float wy_tot = 0.f, wx_tot = 0.f;
    for (int j = 0; j < ty.count; ++j) wy_tot += tap_w(ty, j, aay, ly);
    for (int j = 0; j < tx.count; ++j) wx_tot += tap_w(tx, j, aax, lx);
    float acc = 0.f;
    for (int jy = 0; jy < ty.count; ++jy) {
      const int iy = min(ty.start + jy, F - 1);
      const float wy = tap_w(ty, jy, aay, ly) / wy_tot;
      float row = 0.f;
      for (int jx = 0; jx < tx.count; ++jx) {
        const int ix = min(tx.start + jx, nT - 1);
        float v = src[(int64_t)iy * nT + ix];
        if (normalise) v = flat ? 0.f : (v - lo) * inv;
        row += v * (tap_w(tx, jx, aax, lx) / wx_tot);
      }
      acc += row * wy;
    }
    out[(int64_t)bc * S * S + e] = acc;
  }
}

struct FeDims {
  int hop, nT, F, tiles, nfft, win, nbin, kmode;
  int64_t off_cos, off_sin, off_fb, off_spec, off_mm, total;
};
// mode 0 / 1 / 2: cut configuration (mel+log, linear+log, linear raw); 3 / 4: un-cut configuration (mel+log, linear+log)
FeDims fe_dims(int B, int T, int mode) {
  FeDims d;
  const bool uncut = mode >= 3;
  d.nfft = uncut ? 400 : 512;
  d.win = uncut ? 200 : 64;
  d.nbin = d.nfft / 2 + 1;
  d.kmode = uncut ? mode - 3 : mode;            // what the kernels see: 0 mel+log, 1 linear+log, 2 linear raw
  const int NBIN = d.nbin, WIN = d.win;
  // mel path passes no hop_length -> win_length // 2 (= 32 cut, 100 un-cut); linear: 16 cut (:108), 100 un-cut (:99)
  d.hop = uncut ? 100 : (mode == 0 ? WIN / 2 : WIN / 4);
  d.nT = 1 + T / d.hop;
  d.F = d.kmode == 0 ? NMEL : NBIN;
  d.tiles = (int)adn_cdiv(d.nT, FT);
  d.off_cos = 0;
  d.off_sin = d.off_cos + NBIN * WIN;
  d.off_fb = d.off_sin + NBIN * WIN;
  d.off_spec = d.off_fb + NBIN * NMEL;
  d.off_mm = d.off_spec + (int64_t)B * 2 * d.F * d.nT;
  d.total = d.off_mm + (int64_t)B * 2 * d.tiles * 2;
  return d;
}

}  // namespace

extern "C" int64_t adn_frontend_workspace_bytes(int32_t B, int32_t T, int32_t mode) {
  if (B <= 0 || T <= 0 || mode < 0 || mode > 4) return -1;
  return fe_dims(B, T, mode).total * 4;
}

extern "C" int adn_frontend(const float* wave, int32_t B, int32_t T, int32_t mode, int32_t S, int32_t antialias,
                            float* out, void* workspace, int64_t workspace_bytes, void* stream) {
  ADN_CHECK_ARG(wave && out && workspace && B > 0 && S > 0, "adn_frontend: bad arguments");
  ADN_CHECK_ARG(mode >= 0 && mode <= 4, "adn_frontend: bad mode %d", mode);
  const FeDims d = fe_dims(B, T, mode);
  ADN_CHECK_ARG(T > d.nfft / 2, "adn_frontend: reflect padding needs T > %d samples (got %d)", d.nfft / 2, T);
  ADN_CHECK_ARG(workspace_bytes >= d.total * 4, "adn_frontend: workspace too small");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  float* ws = reinterpret_cast<float*>(workspace);
  const dim3 tgrid((unsigned)adn_cdiv(d.nbin * (d.win > NMEL ? d.win : NMEL), 256)), sgrid(d.tiles, B * 2);
  if (d.nfft == 512) {
    hipLaunchKernelGGL((fe_tables_kernel<512, 64>), tgrid, dim3(256), 0, st, ws + d.off_cos, ws + d.off_sin, ws + d.off_fb);
    ADN_CHECK_LAUNCH();
    hipLaunchKernelGGL((fe_stft_kernel<512, 64>), sgrid, dim3(256), 0, st, wave, T, d.hop, d.nT, d.kmode, ws + d.off_cos,
                       ws + d.off_sin, ws + d.off_fb, ws + d.off_spec, ws + d.off_mm, d.tiles);
  } else {
    hipLaunchKernelGGL((fe_tables_kernel<400, 200>), tgrid, dim3(256), 0, st, ws + d.off_cos, ws + d.off_sin, ws + d.off_fb);
    ADN_CHECK_LAUNCH();
    hipLaunchKernelGGL((fe_stft_kernel<400, 200>), sgrid, dim3(256), 0, st, wave, T, d.hop, d.nT, d.kmode, ws + d.off_cos,
                       ws + d.off_sin, ws + d.off_fb, ws + d.off_spec, ws + d.off_mm, d.tiles);
  }
  ADN_CHECK_LAUNCH();
  hipLaunchKernelGGL(fe_resize_kernel, dim3((unsigned)adn_cdiv((int64_t)S * S, 256), B * 2), dim3(256), 0, st,
                     ws + d.off_spec, ws + d.off_mm, d.tiles, d.F, d.nT, S, antialias, d.kmode != 2 ? 1 : 0, out);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_resize_bilinear(const float* src, int32_t planes, int32_t H, int32_t W, int32_t S, int32_t antialias,
                                   float* out, void* stream) {
  ADN_CHECK_ARG(src && out && planes > 0 && H > 0 && W > 0 && S > 0, "adn_resize_bilinear: bad arguments");
  hipLaunchKernelGGL(fe_resize_kernel, dim3((unsigned)adn_cdiv((int64_t)S * S, 256), planes), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), src, static_cast<const float*>(nullptr), 0, H, W, S, antialias,
                     0, out);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

// Adjoint of the non-antialiased bilinear resize above ([planes][H][W] -> [planes][S][S], align_corners=False): one thread
// per SOURCE pixel gathers the output gradients that interpolate from it, with the forward's own index / weight
// arithmetic (no atomics: the candidate output rows / columns of a source row are a short contiguous range).
__device__ __forceinline__ void resize_src(int o, float scale, int n, int& i0, int& i1, float& l) {
  const float s = fmaxf((o + 0.5f) * scale - 0.5f, 0.f);
  i0 = min((int)s, n - 1);
  i1 = min(i0 + 1, n - 1);
  l = s - i0;
}
__global__ __launch_bounds__(256) void resize_bilinear_bwd_kernel(const float* gout, int planes, int H, int W, int S,
                                                                  float* gin) {
  const float sy = (float)H / S, sx = (float)W / S;
  const int64_t n = (int64_t)planes * H * W;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int ix = (int)(e % W), iy = (int)((e / W) % H);
    const int64_t pl = e / ((int64_t)W * H);
    // outputs whose i0 is iy - 1 or iy: (o + 0.5) * scale - 0.5 in [iy - 1, iy + 1)
    const int oy_lo = max((int)floorf(((float)iy - 0.5f) / sy - 0.5f) - 1, 0);
    const int oy_hi = min((int)ceilf(((float)iy + 1.5f) / sy - 0.5f) + 1, S - 1);
    const int ox_lo = max((int)floorf(((float)ix - 0.5f) / sx - 0.5f) - 1, 0);
    const int ox_hi = min((int)ceilf(((float)ix + 1.5f) / sx - 0.5f) + 1, S - 1);
    const float* g = gout + pl * S * S;
    float acc = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      int y0, y1;
      float ly;
      resize_src(oy, sy, H, y0, y1, ly);
      const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
      if (wy == 0.f) continue;
      float row = 0.f;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        int x0, x1;
        float lx;
        resize_src(ox, sx, W, x0, x1, lx);
        const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
        if (wx != 0.f) row += wx * g[oy * S + ox];
      }
      acc += wy * row;
    }
    gin[e] = acc;
  }
}

extern "C" int adn_resize_bilinear_bwd(const float* gout, int32_t planes, int32_t H, int32_t W, int32_t S, float* gin,
                                       void* stream) {
  ADN_CHECK_ARG(gout && gin && planes > 0 && H > 0 && W > 0 && S > 0, "adn_resize_bilinear_bwd: bad arguments");
  int64_t blocks = adn_cdiv((int64_t)planes * H * W, 256);
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(resize_bilinear_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), gout, planes, H, W, S, gin);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

// ---- depth-target preparation (BatvisionV2_Dataset.py:65-78, BatvisionV1_Dataset.py:45-64) -------------------------
// raw depth in millimetres (f32 / u16 / i32) -> metres, NaN / +-inf -> 0, clip to max_depth (when > 0), negatives -> 0,
// cv2.INTER_NEAREST resize (source index = min(floor(dst * in / out), in - 1)), optional / norm.
template <typename S>
__global__ __launch_bounds__(256) void depth_prepare_kernel(const S* src, int planes, int H, int W, int So, float maxd,
                                                            float norm, float* out) {
  const int64_t n = (int64_t)planes * So * So;
  const double sy = (double)H / So, sx = (double)W / So;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int x = (int)(e % So), y = (int)((e / So) % So);
    const int64_t pl = e / ((int64_t)So * So);
    int iy = (int)floor(y * sy), ix = (int)floor(x * sx);
    iy = iy < H - 1 ? iy : H - 1;
    ix = ix < W - 1 ? ix : W - 1;
    float d = (float)src[(pl * H + iy) * W + ix];
    if (!(d == d) || d == INFINITY || d == -INFINITY) d = 0.f;
    d = d / 1000.0f;
    if (maxd > 0.f && d > maxd) d = maxd;
    if (d < 0.f) d = 0.f;
    out[e] = norm > 0.f ? d / norm : d;
  }
}

extern "C" int adn_depth_prepare(const void* src, int32_t src_type, int32_t planes, int32_t H, int32_t W, int32_t S,
                                 float max_depth, float norm, float* out, void* stream) {
  ADN_CHECK_ARG(src && out && planes > 0 && H > 0 && W > 0 && S > 0, "adn_depth_prepare: bad arguments");
  ADN_CHECK_ARG(src_type >= 0 && src_type <= 2, "adn_depth_prepare: src_type %d (0 f32, 1 u16, 2 i32)", src_type);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int64_t blocks = adn_cdiv((int64_t)planes * S * S, 256);
  if (blocks > 4096) blocks = 4096;
  const dim3 grid((unsigned)blocks);
  if (src_type == 0)
    hipLaunchKernelGGL((depth_prepare_kernel<float>), grid, dim3(256), 0, st, reinterpret_cast<const float*>(src), planes, H, W, S, max_depth, norm, out);
  else if (src_type == 1)
    hipLaunchKernelGGL((depth_prepare_kernel<uint16_t>), grid, dim3(256), 0, st, reinterpret_cast<const uint16_t*>(src), planes, H, W, S, max_depth, norm, out);
  else
    hipLaunchKernelGGL((depth_prepare_kernel<int32_t>), grid, dim3(256), 0, st, reinterpret_cast<const int32_t*>(src), planes, H, W, S, max_depth, norm, out);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

// ---- camera-image preparation (BatvisionV2_Dataset.py:199-210 _load_image after cv2.imread) -------------------------------
// src u8 [B][H][W][3] BGR (the decoded file) -> out f32 [B][3][S][S] RGB in [0,1]:
//   cv2.cvtColor(BGR2RGB) -> cv2.resize((S,S)) (INTER_LINEAR, 8-bit path) -> / 255 -> HWC to CHW.
// The 8-bit INTER_LINEAR path of OpenCV is integer arithmetic, restated here so the result is bit-identical to the host
// restatement (oracle/frontend_oracle.resize_linear_cv2_u8; "parity unpinned" against OpenCV itself: cv2 is not installed):
//   source position fx = (dx + 0.5) * (W / S) - 0.5, sx = floor(fx), a = fx - sx; sx < 0 -> (0, a = 0); sx >= W - 1 -> (W - 1, a = 0)
//   coefficients in 11 bits: c1 = saturate_short(rint(a * 2048)), c0 = 2048 - c1 (cvRound = round half to even)
//   horizontal pass in int: D = S[sx] * c0 + S[sx + 1] * c1;  vertical pass of the 8-bit specialisation:
//   dst = (((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2.
__device__ __forceinline__ void cv_lin_coef(int d, double scale, int n, int& s0, int& s1, int& c0, int& c1) {
  const float fx = (float)((d + 0.5) * scale - 0.5);
  int sx = (int)floorf(fx);
  float a = fx - sx;
  if (sx < 0) {
    sx = 0;
    a = 0.f;
  }
  if (sx >= n - 1) {
    sx = n - 1;
    a = 0.f;
  }
  s0 = sx;
  s1 = sx + 1 < n ? sx + 1 : n - 1;
  c1 = (int)rintf(a * 2048.f);
  c0 = 2048 - c1;
}

__global__ __launch_bounds__(256) void image_prepare_kernel(const uint8_t* src, int B, int H, int W, int So, float* out) {
  const int64_t n = (int64_t)B * So * So;
  const double sy = (double)H / So, sx = (double)W / So;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int x = (int)(e % So), y = (int)((e / So) % So);
    const int64_t b = e / ((int64_t)So * So);
    int y0, y1, b0, b1, x0, x1, a0, a1;
    cv_lin_coef(y, sy, H, y0, y1, b0, b1);
    cv_lin_coef(x, sx, W, x0, x1, a0, a1);
    const uint8_t* r0 = src + ((b * H + y0) * W) * 3;
    const uint8_t* r1 = src + ((b * H + y1) * W) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {                     // c = BGR channel of the source; RGB plane 2 - c of the output
      const int d0 = r0[x0 * 3 + c] * a0 + r0[x1 * 3 + c] * a1;
      const int d1 = r1[x0 * 3 + c] * a0 + r1[x1 * 3 + c] * a1;
      const int v = (((b0 * (d0 >> 4)) >> 16) + ((b1 * (d1 >> 4)) >> 16) + 2) >> 2;
      out[((b * 3 + (2 - c)) * So + y) * So + x] = (float)v / 255.0f;
    }
  }
}

extern "C" int adn_image_prepare(const void* src_bgr_u8, int32_t B, int32_t H, int32_t W, int32_t S, float* out, void* stream) {
  ADN_CHECK_ARG(src_bgr_u8 && out && B > 0 && H > 0 && W > 0 && S > 0, "adn_image_prepare: bad arguments");
  int64_t blocks = adn_cdiv((int64_t)B * S * S, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(image_prepare_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const uint8_t*>(src_bgr_u8), B, H, W, S, out);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

// ---- F.interpolate(mode='nearest') to (S, S): src f32 [planes][H][W] -> out [planes][S][S], source index
// min(floor(dst * in / out), in - 1) (torch's legacy 'nearest' = OpenCV's INTER_NEAREST rule).  Used by the AdaBins model
// for its outputs when output_size != input size (adabins_distillation_model.py:196-198, 334-337, 383-386).
__global__ __launch_bounds__(256) void resize_nearest_kernel(const float* src, int64_t planes, int H, int W, int So, float* out) {
  const int64_t n = planes * So * So;
  const float sy = (float)H / So, sx = (float)W / So;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int x = (int)(e % So), y = (int)((e / So) % So);
    const int64_t pl = e / ((int64_t)So * So);
    int iy = (int)floorf(y * sy), ix = (int)floorf(x * sx);
    iy = iy < H - 1 ? iy : H - 1;
    ix = ix < W - 1 ? ix : W - 1;
    out[e] = src[(pl * H + iy) * W + ix];
  }
}

// Backward of the above as a gather: gsrc[pl][iy][ix] = sum of gout over the output pixels whose source index is (iy, ix)
// (the same index rule, evaluated on a candidate window around iy / scale; no atomics).
__global__ __launch_bounds__(256) void resize_nearest_bwd_kernel(const float* gout, int64_t planes, int H, int W, int So,
                                                                 float* gsrc) {
  const int64_t n = planes * H * W;
  const float sy = (float)H / So, sx = (float)W / So;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int ix = (int)(e % W), iy = (int)((e / W) % H);
    const int64_t pl = e / ((int64_t)H * W);
    int y0 = (int)floorf(iy / sy) - 1, y1 = (int)floorf((iy + 1) / sy) + 1;
    int x0 = (int)floorf(ix / sx) - 1, x1 = (int)floorf((ix + 1) / sx) + 1;
    y0 = y0 < 0 ? 0 : y0;
    x0 = x0 < 0 ? 0 : x0;
    y1 = y1 > So - 1 ? So - 1 : y1;
    x1 = x1 > So - 1 ? So - 1 : x1;
    float acc = 0.f;
    for (int y = y0; y <= y1; ++y) {
      int ty = (int)floorf(y * sy);
      ty = ty < H - 1 ? ty : H - 1;
      if (ty != iy) continue;
      for (int x = x0; x <= x1; ++x) {
        int tx = (int)floorf(x * sx);
        tx = tx < W - 1 ? tx : W - 1;
        if (tx == ix) acc += gout[(pl * So + y) * So + x];
      }
    }
    gsrc[e] = acc;
  }
}

extern "C" int adn_resize_nearest(const float* src, int64_t planes, int32_t H, int32_t W, int32_t S, float* out, void* stream) {
  ADN_CHECK_ARG(src && out && planes > 0 && H > 0 && W > 0 && S > 0, "adn_resize_nearest: bad arguments");
  int64_t blocks = adn_cdiv(planes * S * S, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(resize_nearest_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src,
                     planes, H, W, S, out);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_resize_nearest_bwd(const float* gout, int64_t planes, int32_t H, int32_t W, int32_t S, float* gsrc,
                                      void* stream) {
  ADN_CHECK_ARG(gout && gsrc && planes > 0 && H > 0 && W > 0 && S > 0, "adn_resize_nearest_bwd: bad arguments");
  int64_t blocks = adn_cdiv(planes * H * W, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(resize_nearest_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     gout, planes, H, W, S, gsrc);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}
