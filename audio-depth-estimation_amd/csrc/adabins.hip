// AdaBins distillation model: the pieces that are not convolutions (adabins_distillation_model.py:105-207, 301-399)
// and the distillation loss (utils_distillation_loss.py:48-238).  All memory-bound or tiny; f32 arithmetic, f64 final
// reductions, no atomics.
//   pool_mean      AdaptiveAvgPool2d(1) over NHWC (also the spatial mean of the bin logits for the KL term)
//   binpred_*      Linear -> ReLU -> Dropout -> Linear -> Softmax -> cumsum -> bin centres, forward and backward
//   bcast_add      gradient of the average pool
//   bins_*         per-pixel softmax over the bins and the expectation sum_k p_k c_k (base depth), fwd / bwd
//   distill_pix_*  final = clamp(base + residual), masked L1 / MSE-to-teacher / |residual| terms and their gradients
//   featcos_*      1 - mean cosine similarity of spatially normalised features, per encoder level
//   distill_small  KL of the temperature-softened mean logits, bin-centre MSE, assembly of the total loss
#include "adn_common.h"
#include "epilogue.h"

namespace {

template <typename T>
__device__ __forceinline__ float ldg(const void* base, int64_t idx) {
  return ElemTraits<T>::load(reinterpret_cast<const T*>(base) + idx);
}

__device__ __forceinline__ float block_sum(float v, float* sm) {      // 256 threads, sm[4]
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  return sm[0] + sm[1] + sm[2] + sm[3];
}
__device__ __forceinline__ float block_max(float v, float* sm) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}

// ---- per-sample sums over rows: x [B][HW][ld] -> partial [B][P][NQ][C] -------------------------------------------
// NQ = 1: sum x;  NQ = 3 (second tensor y): sum x^2, sum y^2, sum x*y
template <typename T, int NQ>
__global__ __launch_bounds__(256) void rowsum_partial(const void* x, const void* y, int HW, int C, int ld, float* partial) {
  __shared__ float red[256 * 8 * NQ];
  const int b = blockIdx.y, P = gridDim.x;
  const int rpb = (HW + P - 1) / P;
  const int r0 = blockIdx.x * rpb, r1 = r0 + rpb < HW ? r0 + rpb : HW;
  float* po = partial + (((int64_t)b * P + blockIdx.x) * NQ) * C;
  if ((C & 7) == 0 && (ld & 7) == 0 && C <= 2048) {
    // 8 channels per thread, 256 / (C/8) rows per iteration, cross-row-group reduction through LDS
    const int ncg = C >> 3;
    const int rpi = 256 / ncg;
    const int r = threadIdx.x / ncg, cg = threadIdx.x - r * ncg;
    float s[NQ][8];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int k = 0; k < 8; ++k) s[q][k] = 0.f;
    if (r < rpi) {
      for (int row = r0 + r; row < r1; row += rpi) {
        const int64_t e = ((int64_t)b * HW + row) * ld + cg * 8;
        float a[8];
        load8<T>(x, e, a);
        if constexpr (NQ == 1) {
#pragma unroll
          for (int k = 0; k < 8; ++k) s[0][k] += a[k];
        } else {
          float q8[8];
          load8<T>(y, e, q8);
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            s[0][k] += a[k] * a[k];
            s[NQ - 2][k] += q8[k] * q8[k];
            s[NQ - 1][k] += a[k] * q8[k];
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int k = 0; k < 8; ++k) red[(threadIdx.x * NQ + q) * 8 + k] = s[q][k];
    __syncthreads();
    for (int t = threadIdx.x; t < NQ * C; t += 256) {
      const int q = t / C, c = t - q * C;
      const int g = c >> 3, k = c & 7;
      float a = 0.f;
      for (int rr = 0; rr < rpi; ++rr) a += red[((rr * ncg + g) * NQ + q) * 8 + k];
      po[(int64_t)q * C + c] = a;
    }
    return;
  }
  for (int c = threadIdx.x; c < C; c += 256) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int r = r0; r < r1; ++r) {
      const int64_t e = ((int64_t)b * HW + r) * ld + c;
      const float a = ldg<T>(x, e);
      if (NQ == 1) s0 += a;
      else {
        const float q = ldg<T>(y, e);
        s0 += a * a;
        s1 += q * q;
        s2 += a * q;
      }
    }
    po[c] = s0;
    if (NQ == 3) {
      po[C + c] = s1;
      po[2 * C + c] = s2;
    }
  }
}

// out[b][q][c] = scale * sum_p partial[b][p][q][c]
__global__ __launch_bounds__(256) void rowsum_final(const float* partial, int P, int NQC, float scale, float* out) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= NQC) return;
  double s = 0.0;
  for (int p = 0; p < P; ++p) s += (double)partial[((int64_t)b * P + p) * NQC + j];
  out[(int64_t)b * NQC + j] = (float)(s * scale);
}

// ---- bin predictor ------------------------------------------------------------------------------------------------
constexpr int kMaxBins = 256, kMaxHid = 256, kMaxBott = 1024;

__global__ __launch_bounds__(256) void binpred_fwd_kernel(const float* g, const float* W1, const float* b1, const float* W2,
                                                          const float* b2, const uint8_t* mask, float drop_scale,
                                                          float maxd, int Cb, int Hd, int nb, float* h1, float* widths,
                                                          float* centers) {
  __shared__ float gs[kMaxBott], hs[kMaxHid], ls[kMaxBins], sm[4];
  const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int c = threadIdx.x; c < Cb; c += 256) gs[c] = g[(int64_t)b * Cb + c];
  __syncthreads();
  for (int j = wave; j < Hd; j += 4) {
    float s = 0.f;
    for (int c = lane; c < Cb; c += 64) s += W1[(int64_t)j * Cb + c] * gs[c];
    s = wave_sum(s);
    if (lane == 0) {
      float h = fmaxf(s + b1[j], 0.f);
      if (mask) h = mask[(int64_t)b * Hd + j] ? h * drop_scale : 0.f;
      hs[j] = h;
      h1[(int64_t)b * Hd + j] = h;
    }
  }
  __syncthreads();
  for (int k = wave; k < nb; k += 4) {
    float s = 0.f;
    for (int j = lane; j < Hd; j += 64) s += W2[(int64_t)k * Hd + j] * hs[j];
    s = wave_sum(s);
    if (lane == 0) ls[k] = s + b2[k];
  }
  __syncthreads();
  const float lv = threadIdx.x < nb ? ls[threadIdx.x] : -INFINITY;
  const float mx = block_max(lv, sm);
  const float ev = threadIdx.x < nb ? __expf(lv - mx) : 0.f;
  const float tot = block_sum(ev, sm);
  if (threadIdx.x < nb) {
    ls[threadIdx.x] = ev / tot;
    widths[(int64_t)b * nb + threadIdx.x] = ev / tot;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float edge = 0.f;
    for (int k = 0; k < nb; ++k) {          // edges = cumsum(widths) * max_depth; centre = midpoint
      const float lo = edge * maxd;
      edge += ls[k];
      centers[(int64_t)b * nb + k] = (lo + edge * maxd) * 0.5f;
    }
  }
}

__global__ __launch_bounds__(256) void binpred_bwd_kernel(const float* dcent, const float* widths, const float* h1,
                                                          const float* g, const float* W1, const float* W2,
                                                          float drop_scale, float maxd, int Cb, int Hd, int nb,
                                                          float* dW2p, float* db2p, float* dW1p, float* db1p, float* dg) {
  __shared__ float dl[kMaxBins], dz[kMaxHid], gs[kMaxBott], sm[4];
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < Cb; c += 256) gs[c] = g[(int64_t)b * Cb + c];
  if (threadIdx.x == 0) {
    // centre_i = maxd * (sum_{j<i} w_j + w_i / 2)  =>  d w_j = maxd * (sum_{i>j} dC_i + dC_j / 2)
    float suffix = 0.f;
    for (int k = nb - 1; k >= 0; --k) {
      const float d = dcent[(int64_t)b * nb + k];
      dl[k] = maxd * (suffix + 0.5f * d);
      suffix += d;
    }
  }
  __syncthreads();
  const float w = threadIdx.x < nb ? widths[(int64_t)b * nb + threadIdx.x] : 0.f;
  const float dw = threadIdx.x < nb ? dl[threadIdx.x] : 0.f;
  const float dot = block_sum(w * dw, sm);
  __syncthreads();
  if (threadIdx.x < nb) {
    const float d = w * (dw - dot);                   // softmax backward
    dl[threadIdx.x] = d;
    db2p[(int64_t)b * nb + threadIdx.x] = d;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < nb * Hd; e += 256) {
    const int k = e / Hd, j = e - k * Hd;
    dW2p[(int64_t)b * nb * Hd + e] = dl[k] * h1[(int64_t)b * Hd + j];
  }
  for (int j = threadIdx.x; j < Hd; j += 256) {
    float s = 0.f;
    for (int k = 0; k < nb; ++k) s += W2[(int64_t)k * Hd + j] * dl[k];
    const float d = h1[(int64_t)b * Hd + j] > 0.f ? s * drop_scale : 0.f;   // dropout scale and ReLU mask in one
    dz[j] = d;
    db1p[(int64_t)b * Hd + j] = d;
  }
  __syncthreads();
  for (int64_t e = threadIdx.x; e < (int64_t)Hd * Cb; e += 256) {
    const int j = (int)(e / Cb), c = (int)(e - (int64_t)j * Cb);
    dW1p[(int64_t)b * Hd * Cb + e] = dz[j] * gs[c];
  }
  for (int c = threadIdx.x; c < Cb; c += 256) {
    float s = 0.f;
    for (int j = 0; j < Hd; ++j) s += W1[(int64_t)j * Cb + c] * dz[j];
    dg[(int64_t)b * Cb + c] = s;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bcast_add_kernel(void* gx, const float* dg, int HW, int C, float scale,
                                                        int accumulate, int64_t n) {
  T* gp = reinterpret_cast<T*>(gx);
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int c = (int)(e % C);
    const int64_t b = e / ((int64_t)HW * C);
    float v = dg[b * C + c] * scale;
    if (accumulate) v += ElemTraits<T>::load(gp + e);
    ElemTraits<T>::store(gp + e, v);
  }
}

// Bernoulli keep mask from a counter-based hash (one draw per element and step; not torch's RNG stream)
__global__ __launch_bounds__(256) void dropout_mask_kernel(uint8_t* mask, int64_t n, float p, uint64_t seed,
                                                           const double* counter) {
  if (counter) seed += 0xD1B54A32D192ED03ull * (uint64_t)(counter[0] + 1.0);     // device-side step count: graph-replay safe
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(e + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const float u = (float)(z >> 40) * (1.0f / 16777216.0f);
    mask[e] = u >= p ? 1 : 0;
  }
}

// ---- per-pixel softmax over the bins and its expectation ------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bins_fwd_kernel(const void* logits, const float* centers, int64_t pixels, int HW,
                                                       int nb, float* base) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int64_t pix = (int64_t)blockIdx.x * 4 + wave; pix < pixels; pix += (int64_t)gridDim.x * 4) {
    const int64_t b = pix / HW;
    float lv[4], mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = lane + 64 * u;
      lv[u] = k < nb ? ldg<T>(logits, pix * nb + k) : -INFINITY;
      mx = fmaxf(mx, lv[u]);
    }
    mx = wave_max(mx);
    float se = 0.f, sc = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = lane + 64 * u;
      if (k < nb) {
        const float e = __expf(lv[u] - mx);
        se += e;
        sc += e * centers[b * nb + k];
      }
    }
    se = wave_sum(se);
    sc = wave_sum(sc);
    if (lane == 0) base[pix] = sc / se;
  }
}

// bf16, nb % 8 == 0, nb / 8 a power of two <= 64: a pixel's bins are nb / 8 lanes x one 16-byte load each, 64 / (nb / 8)
// pixels per wave-iteration, reductions across those few lanes only (128 bins: 4 xor steps instead of 6, x4 pixels)
template <int LPP>
__global__ __launch_bounds__(256) void bins_fwd_vec_kernel(const uint16_t* logits, const float* centers, int64_t pixels,
                                                           int HW, float* base) {
  constexpr int NB = LPP * 8, PPW = 64 / LPP;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = lane % LPP, pw = lane / LPP;
  for (int64_t p0 = ((int64_t)blockIdx.x * 4 + wave) * PPW; p0 < pixels; p0 += (int64_t)gridDim.x * 4 * PPW) {
    const int64_t pix = p0 + pw;
    const bool live = pix < pixels;
    float lv[8];
    float mx = -INFINITY;
    if (live) {
      const u32x4_t c = *reinterpret_cast<const u32x4_t*>(logits + pix * NB + li * 8);
      Chunk<uint16_t>::unpack(c, lv);
#pragma unroll
      for (int k = 0; k < 8; ++k) mx = fmaxf(mx, lv[k]);
    }
#pragma unroll
    for (int o = 1; o < LPP; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float se = 0.f, sc = 0.f;
    if (live) {
      const float* cr = centers + (pix / HW) * NB + li * 8;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float e = __expf(lv[k] - mx);
        se += e;
        sc += e * cr[k];
      }
    }
#pragma unroll
    for (int o = 1; o < LPP; o <<= 1) {
      se += __shfl_xor(se, o, 64);
      sc += __shfl_xor(sc, o, 64);
    }
    if (live && li == 0) base[pix] = sc / se;
  }
}

// dlogit_k = p_k (c_k - base) dbase + dmean[b][k] / HW;  dcentre partial[b][block][k] = sum_pix p_k dbase
template <typename T>
__global__ __launch_bounds__(256) void bins_bwd_kernel(const void* logits, const float* centers, const float* base,
                                                       const float* dbase, const float* dmean, float inv_hw, int HW, int nb,
                                                       void* dlogits, float* dcpart) {
  __shared__ float red[4][kMaxBins];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.y, P = gridDim.x;
  const int rpb = (HW + P - 1) / P;
  const int r0 = blockIdx.x * rpb, r1 = r0 + rpb < HW ? r0 + rpb : HW;
  float dc[4] = {0.f, 0.f, 0.f, 0.f};
  float cv[4], dm[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int k = lane + 64 * u;
    cv[u] = k < nb ? centers[(int64_t)b * nb + k] : 0.f;
    dm[u] = (k < nb && dmean) ? dmean[(int64_t)b * nb + k] * inv_hw : 0.f;
  }
  for (int r = r0 + wave; r < r1; r += 4) {
    const int64_t pix = (int64_t)b * HW + r;
    float lv[4], mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = lane + 64 * u;
      lv[u] = k < nb ? ldg<T>(logits, pix * nb + k) : -INFINITY;
      mx = fmaxf(mx, lv[u]);
    }
    mx = wave_max(mx);
    float se = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      lv[u] = (lane + 64 * u) < nb ? __expf(lv[u] - mx) : 0.f;
      se += lv[u];
    }
    se = wave_sum(se);
    const float inv = 1.f / se, bs = base[pix], db = dbase[pix];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = lane + 64 * u;
      if (k < nb) {
        const float pk = lv[u] * inv;
        dc[u] += pk * db;
        ElemTraits<T>::store(reinterpret_cast<T*>(dlogits) + pix * nb + k, pk * (cv[u] - bs) * db + dm[u]);
      }
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) red[wave][lane + 64 * u] = dc[u];
  __syncthreads();
  for (int k = threadIdx.x; k < nb; k += 256)
    dcpart[((int64_t)b * P + blockIdx.x) * nb + k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
}

// ---- pixel terms of the distillation loss ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void distill_pix_stats_kernel(const float* base, const float* resid, const float* gt,
                                                                const float* teacher, int64_t n, float maxd, float* final_out,
                                                                double* partial) {
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const float f = fminf(fmaxf(base[e] + resid[e], 0.f), maxd);
    final_out[e] = f;
    if (gt[e] > 0.f) {
      s0 += 1.0;
      s1 += (double)fabsf(f - gt[e]);
      if (teacher) {
        const float d = f - teacher[e];
        s2 += (double)(d * d);
      }
      s3 += (double)fabsf(resid[e]);
    }
  }
  __shared__ double sm[4][4];
  s0 = wave_sum_d(s0); s1 = wave_sum_d(s1); s2 = wave_sum_d(s2); s3 = wave_sum_d(s3);
  if ((threadIdx.x & 63) == 0) {
    const int w = threadIdx.x >> 6;
    sm[0][w] = s0; sm[1][w] = s1; sm[2][w] = s2; sm[3][w] = s3;
  }
  __syncthreads();
  if (threadIdx.x < 4)
    partial[(int64_t)blockIdx.x * 4 + threadIdx.x] = sm[threadIdx.x][0] + sm[threadIdx.x][1] + sm[threadIdx.x][2] + sm[threadIdx.x][3];
}

__global__ __launch_bounds__(64) void sum4_kernel(const double* partial, int nbk, double* stats) {
  for (int q = 0; q < 4; ++q) {
    double s = 0.0;
    for (int r = threadIdx.x; r < nbk; r += 64) s += partial[(int64_t)r * 4 + q];
    s = wave_sum_d(s);
    if (threadIdx.x == 0) stats[q] = s;
  }
}

__device__ __forceinline__ float sgnf(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

__global__ __launch_bounds__(256) void distill_pix_grad_kernel(const float* base, const float* resid, const float* gt,
                                                               const float* teacher, int64_t n, float maxd,
                                                               const double* stats, float lt, float lr, float ls,
                                                               float* dbase, float* dres) {
  const float invn = stats[0] > 0.0 ? (float)(1.0 / stats[0]) : 0.f;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const float s = base[e] + resid[e];
    const float f = fminf(fmaxf(s, 0.f), maxd);
    float db = 0.f, dr = 0.f;
    if (gt[e] > 0.f) {
      float df = lt * sgnf(f - gt[e]) * invn;
      if (teacher) df += lr * 2.f * (f - teacher[e]) * invn;
      if (s >= 0.f && s <= maxd) db = df;                  // torch.clamp passes the gradient on [min, max]
      dr = db + ls * sgnf(resid[e]) * invn;
    }
    dbase[e] = db;
    dres[e] = dr;
  }
}

// ---- feature cosine distance --------------------------------------------------------------------------------------------
// stats [B][3][C] = (sum a^2, sum r^2, sum a r) over the pixels.  cos[b][c] = ar / (max(|a|,eps) max(|r|,eps)).
// ga (+)= coef * d cos / d a,  d cos / d a = r / (|a||r|) - ar * a / (|a|^3 |r|)
template <typename T>
__global__ __launch_bounds__(256) void featcos_grad_kernel(const void* a, const void* r, const float* stats, int HW, int C,
                                                           float coef, void* ga, int64_t n) {
  T* gp = reinterpret_cast<T*>(ga);
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int c = (int)(e % C);
    const int64_t b = e / ((int64_t)HW * C);
    const float* st = stats + b * 3 * C;
    const float na = fmaxf(sqrtf(st[c]), 1e-12f), nr = fmaxf(sqrtf(st[C + c]), 1e-12f), ar = st[2 * C + c];
    const float av = ldg<T>(a, e), rv = ldg<T>(r, e);
    float d = rv / (na * nr);
    if (sqrtf(st[c]) > 1e-12f) d -= ar * av / (na * na * na * nr);
    ElemTraits<T>::store(gp + e, ElemTraits<T>::load(gp + e) + coef * d);
  }
}

// ---- small terms + assembly: one block --------------------------------------------------------------------------------------
struct SmallParams {
  const float* ms; const float* mt;            // mean logits student / teacher [B][nb]
  const float* cs; const float* ct;            // bin centres
  const float* fstats[5]; int fC[5];           // feature statistics [B][3][C] per level
  const double* pstats;                        // pixel statistics
  int B, nb, has_teacher;
  float temperature, lt, lr, lf, lb, ls;
  float* terms;                                // [8]: task, response, feature, bin, bin_centers, sparse, total
  float* dmean; float* dcent;                  // [B][nb] gradients wrt the mean logits / extra gradient wrt the centres
};

__global__ __launch_bounds__(256) void distill_small_kernel(SmallParams p) {
  __shared__ float sm[4];
  const double N = p.pstats[0];
  const float task = N > 0 ? (float)(p.pstats[1] / N) : NAN;
  const float resp = p.has_teacher ? (N > 0 ? (float)(p.pstats[2] / N) : NAN) : 0.f;
  const float sparse = N > 0 ? (float)(p.pstats[3] / N) : NAN;
  float feat = 0.f, kl = 0.f, cm = 0.f;
  if (p.has_teacher) {
    for (int lv = 0; lv < 5; ++lv) {
      const int C = p.fC[lv];
      float s = 0.f;
      for (int e = threadIdx.x; e < p.B * C; e += 256) {
        const int b = e / C, c = e - b * C;
        const float* st = p.fstats[lv] + (int64_t)b * 3 * C;
        s += st[2 * C + c] / (fmaxf(sqrtf(st[c]), 1e-12f) * fmaxf(sqrtf(st[C + c]), 1e-12f));
      }
      s = block_sum(s, sm);
      feat += 1.f - s / (float)(p.B * C);
    }
    feat *= 0.2f;
    // KL(batchmean) of softmax(mt / T) against log_softmax(ms / T), one sample at a time
    const float invT = 1.f / p.temperature;
    for (int b = 0; b < p.B; ++b) {
      const int k = threadIdx.x;
      const float sv = k < p.nb ? p.ms[b * p.nb + k] * invT : -INFINITY;
      const float tv = k < p.nb ? p.mt[b * p.nb + k] * invT : -INFINITY;
      const float smx = block_max(sv, sm);
      const float ssum = block_sum(k < p.nb ? __expf(sv - smx) : 0.f, sm);
      const float tmx = block_max(tv, sm);
      const float tsum = block_sum(k < p.nb ? __expf(tv - tmx) : 0.f, sm);
      float term = 0.f;
      if (k < p.nb) {
        const float ls = sv - smx - __logf(ssum), lt2 = tv - tmx - __logf(tsum);
        const float pt = __expf(lt2), ps = __expf(ls);
        term = pt > 0.f ? pt * (lt2 - ls) : 0.f;
        p.dmean[b * p.nb + k] = p.lb * invT * (ps - pt) / (float)p.B;
      }
      kl += block_sum(term, sm);
    }
    kl /= (float)p.B;
    float s = 0.f;
    for (int e = threadIdx.x; e < p.B * p.nb; e += 256) {
      const float d = p.cs[e] - p.ct[e];
      s += d * d;
      p.dcent[e] = p.lb * 2.f * d / (float)(p.B * p.nb);
    }
    cm = block_sum(s, sm) / (float)(p.B * p.nb);
  } else {
    for (int e = threadIdx.x; e < p.B * p.nb; e += 256) {
      p.dmean[e] = 0.f;
      p.dcent[e] = 0.f;
    }
  }
  if (threadIdx.x == 0) {
    p.terms[0] = task; p.terms[1] = resp; p.terms[2] = feat; p.terms[3] = kl; p.terms[4] = cm; p.terms[5] = sparse;
    p.terms[6] = p.lt * task + p.lr * resp + p.lf * feat + p.lb * (kl + cm) + p.ls * sparse;
    p.terms[7] = (float)N;
  }
}

inline unsigned blocks_for(int64_t n) {
  int64_t b = adn_cdiv(n, 256);
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}
inline int pool_parts(int HW) {
  int P = (int)adn_cdiv(HW, 64);
  if (P > 64) P = 64;
  if (P < 1) P = 1;
  return P;
}

}  // namespace

extern "C" int64_t adn_pool_workspace_bytes(int32_t B, int32_t HW, int32_t C, int32_t nq) {
  if (B <= 0 || HW <= 0 || C <= 0 || (nq != 1 && nq != 3)) return -1;
  return (int64_t)B * pool_parts(HW) * nq * C * 4;
}

// nq = 1: out [B][C] = mean over the HW rows of x (scale = 1 / HW) ; nq = 3: out [B][3][C] = sum x^2, sum y^2, sum x y
extern "C" int adn_pool(const void* x, const void* y, int32_t B, int32_t HW, int32_t C, int32_t ld, int32_t nq,
                        int32_t dtype, float scale, float* out, void* workspace, int64_t workspace_bytes, void* stream) {
  ADN_CHECK_ARG(x && out && B > 0 && HW > 0 && C > 0 && ld >= C && (nq == 1 || (nq == 3 && y)), "adn_pool: bad arguments");
  ADN_CHECK_ARG(dtype == ADN_F32 || dtype == ADN_BF16, "adn_pool: bad dtype %d", dtype);
  ADN_CHECK_ARG(workspace && workspace_bytes >= adn_pool_workspace_bytes(B, HW, C, nq), "adn_pool: workspace too small");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int P = pool_parts(HW);
  float* part = reinterpret_cast<float*>(workspace);
  const dim3 grid(P, B);
  if (dtype == ADN_BF16) {
    if (nq == 1) hipLaunchKernelGGL((rowsum_partial<uint16_t, 1>), grid, dim3(256), 0, st, x, y, HW, C, ld, part);
    else hipLaunchKernelGGL((rowsum_partial<uint16_t, 3>), grid, dim3(256), 0, st, x, y, HW, C, ld, part);
  } else {
    if (nq == 1) hipLaunchKernelGGL((rowsum_partial<float, 1>), grid, dim3(256), 0, st, x, y, HW, C, ld, part);
    else hipLaunchKernelGGL((rowsum_partial<float, 3>), grid, dim3(256), 0, st, x, y, HW, C, ld, part);
  }
  ADN_CHECK_LAUNCH();
  hipLaunchKernelGGL(rowsum_final, dim3((unsigned)adn_cdiv(nq * C, 256), B), dim3(256), 0, st, part, P, nq * C, scale, out);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_binpred_fwd(const float* g, const float* W1, const float* b1, const float* W2, const float* b2,
                               const uint8_t* mask, float drop_p, float max_depth, int32_t B, int32_t Cb, int32_t Hd,
                               int32_t nb, float* h1, float* widths, float* centers, void* stream) {
  ADN_CHECK_ARG(g && W1 && b1 && W2 && b2 && h1 && widths && centers && B > 0, "adn_binpred_fwd: bad arguments");
  ADN_CHECK_ARG(Cb > 0 && Cb <= kMaxBott && Hd > 0 && Hd <= kMaxHid && nb > 0 && nb <= kMaxBins,
                "adn_binpred_fwd: dims %d/%d/%d exceed %d/%d/%d", Cb, Hd, nb, kMaxBott, kMaxHid, kMaxBins);
  hipLaunchKernelGGL(binpred_fwd_kernel, dim3(B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g, W1, b1, W2, b2,
                     mask, mask ? 1.f / (1.f - drop_p) : 1.f, max_depth, Cb, Hd, nb, h1, widths, centers);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_binpred_bwd(const float* dcent, const float* widths, const float* h1, const float* g, const float* W1,
                               const float* W2, int32_t has_mask, float drop_p, float max_depth, int32_t B, int32_t Cb,
                               int32_t Hd, int32_t nb, float* dW2p, float* db2p, float* dW1p, float* db1p, float* dg,
                               void* stream) {
  ADN_CHECK_ARG(dcent && widths && h1 && g && W1 && W2 && dW2p && db2p && dW1p && db1p && dg && B > 0,
                "adn_binpred_bwd: bad arguments");
  ADN_CHECK_ARG(Cb > 0 && Cb <= kMaxBott && Hd > 0 && Hd <= kMaxHid && nb > 0 && nb <= kMaxBins, "adn_binpred_bwd: bad dims");
  hipLaunchKernelGGL(binpred_bwd_kernel, dim3(B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dcent, widths, h1,
                     g, W1, W2, has_mask ? 1.f / (1.f - drop_p) : 1.f, max_depth, Cb, Hd, nb, dW2p, db2p, dW1p, db1p, dg);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_dropout_mask(uint8_t* mask, int64_t n, float p, uint64_t seed, const double* counter, void* stream) {
  ADN_CHECK_ARG(mask && n > 0 && p >= 0.f && p < 1.f, "adn_dropout_mask: bad arguments");
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(blocks_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), mask, n,
                     p, seed, counter);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_bcast_add(void* gx, const float* dg, int32_t B, int32_t HW, int32_t C, float scale, int32_t accumulate,
                             int32_t dtype, void* stream) {
  ADN_CHECK_ARG(gx && dg && B > 0 && HW > 0 && C > 0, "adn_bcast_add: bad arguments");
  ADN_CHECK_ARG(dtype == ADN_F32 || dtype == ADN_BF16, "adn_bcast_add: bad dtype %d", dtype);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t n = (int64_t)B * HW * C;
  if (dtype == ADN_BF16) hipLaunchKernelGGL((bcast_add_kernel<uint16_t>), dim3(blocks_for(n)), dim3(256), 0, st, gx, dg, HW, C, scale, accumulate, n);
  else hipLaunchKernelGGL((bcast_add_kernel<float>), dim3(blocks_for(n)), dim3(256), 0, st, gx, dg, HW, C, scale, accumulate, n);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_bins_fwd(const void* logits, const float* centers, int32_t B, int32_t HW, int32_t nb, int32_t dtype,
                            float* base, void* stream) {
  ADN_CHECK_ARG(logits && centers && base && B > 0 && HW > 0 && nb > 0 && nb <= kMaxBins, "adn_bins_fwd: bad arguments");
  ADN_CHECK_ARG(dtype == ADN_F32 || dtype == ADN_BF16, "adn_bins_fwd: bad dtype %d", dtype);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t pixels = (int64_t)B * HW;
  const dim3 grid(blocks_for(pixels * 64));
  if (dtype == ADN_BF16 && (nb == 128 || nb == 64) && (reinterpret_cast<uintptr_t>(logits) & 15) == 0) {
    const uint16_t* lg = reinterpret_cast<const uint16_t*>(logits);
    const dim3 gv(blocks_for(pixels * (nb / 8)));
    if (nb == 128) hipLaunchKernelGGL((bins_fwd_vec_kernel<16>), gv, dim3(256), 0, st, lg, centers, pixels, HW, base);
    else hipLaunchKernelGGL((bins_fwd_vec_kernel<8>), gv, dim3(256), 0, st, lg, centers, pixels, HW, base);
  } else if (dtype == ADN_BF16) hipLaunchKernelGGL((bins_fwd_kernel<uint16_t>), grid, dim3(256), 0, st, logits, centers, pixels, HW, nb, base);
  else hipLaunchKernelGGL((bins_fwd_kernel<float>), grid, dim3(256), 0, st, logits, centers, pixels, HW, nb, base);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int64_t adn_bins_bwd_workspace_bytes(int32_t B, int32_t HW, int32_t nb) {
  if (B <= 0 || HW <= 0 || nb <= 0) return -1;
  return (int64_t)B * pool_parts(HW) * nb * 4;
}

extern "C" int adn_bins_bwd(const void* logits, const float* centers, const float* base, const float* dbase,
                            const float* dmean, int32_t B, int32_t HW, int32_t nb, int32_t dtype, void* dlogits,
                            float* dcent, int32_t dcent_accumulate, void* workspace, int64_t workspace_bytes, void* stream) {
  ADN_CHECK_ARG(logits && centers && base && dbase && dlogits && dcent && B > 0 && HW > 0 && nb > 0 && nb <= kMaxBins,
                "adn_bins_bwd: bad arguments");
  ADN_CHECK_ARG(dtype == ADN_F32 || dtype == ADN_BF16, "adn_bins_bwd: bad dtype %d", dtype);
  ADN_CHECK_ARG(workspace && workspace_bytes >= adn_bins_bwd_workspace_bytes(B, HW, nb), "adn_bins_bwd: workspace too small");
  ADN_CHECK_ARG(!dcent_accumulate, "adn_bins_bwd: accumulate into dcent is done by the caller's small-terms kernel");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int P = pool_parts(HW);
  float* part = reinterpret_cast<float*>(workspace);
  const dim3 grid(P, B);
  if (dtype == ADN_BF16)
    hipLaunchKernelGGL((bins_bwd_kernel<uint16_t>), grid, dim3(256), 0, st, logits, centers, base, dbase, dmean, 1.f / HW, HW, nb, dlogits, part);
  else
    hipLaunchKernelGGL((bins_bwd_kernel<float>), grid, dim3(256), 0, st, logits, centers, base, dbase, dmean, 1.f / HW, HW, nb, dlogits, part);
  ADN_CHECK_LAUNCH();
  hipLaunchKernelGGL(rowsum_final, dim3((unsigned)adn_cdiv(nb, 256), B), dim3(256), 0, st, part, P, nb, 1.0f, dcent);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_distill_pix_stats(const float* base, const float* resid, const float* gt, const float* teacher,
                                     int64_t n, float max_depth, float* final_out, double* stats, void* workspace,
                                     int64_t workspace_bytes, void* stream) {
  ADN_CHECK_ARG(base && resid && gt && final_out && stats && n > 0, "adn_distill_pix_stats: bad arguments");
  ADN_CHECK_ARG(workspace && workspace_bytes >= 1024 * 4 * 8, "adn_distill_pix_stats: workspace too small (32 KiB)");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int nbk = (int)adn_cdiv(n, 2048);
  if (nbk > 1024) nbk = 1024;
  if (nbk < 1) nbk = 1;
  double* part = reinterpret_cast<double*>(workspace);
  hipLaunchKernelGGL(distill_pix_stats_kernel, dim3(nbk), dim3(256), 0, st, base, resid, gt, teacher, n, max_depth,
                     final_out, part);
  ADN_CHECK_LAUNCH();
  hipLaunchKernelGGL(sum4_kernel, dim3(1), dim3(64), 0, st, part, nbk, stats);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_distill_pix_grad(const float* base, const float* resid, const float* gt, const float* teacher,
                                    int64_t n, float max_depth, const double* stats, float lambda_task,
                                    float lambda_response, float lambda_sparse, float* dbase, float* dres, void* stream) {
  ADN_CHECK_ARG(base && resid && gt && stats && dbase && dres && n > 0, "adn_distill_pix_grad: bad arguments");
  hipLaunchKernelGGL(distill_pix_grad_kernel, dim3(blocks_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), base,
                     resid, gt, teacher, n, max_depth, stats, lambda_task, lambda_response, lambda_sparse, dbase, dres);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_featcos_grad(const void* a, const void* r, const float* stats, int32_t B, int32_t HW, int32_t C,
                                int32_t dtype, float coef, void* ga, void* stream) {
  ADN_CHECK_ARG(a && r && stats && ga && B > 0 && HW > 0 && C > 0, "adn_featcos_grad: bad arguments");
  ADN_CHECK_ARG(dtype == ADN_F32 || dtype == ADN_BF16, "adn_featcos_grad: bad dtype %d", dtype);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t n = (int64_t)B * HW * C;
  if (dtype == ADN_BF16) hipLaunchKernelGGL((featcos_grad_kernel<uint16_t>), dim3(blocks_for(n)), dim3(256), 0, st, a, r, stats, HW, C, coef, ga, n);
  else hipLaunchKernelGGL((featcos_grad_kernel<float>), dim3(blocks_for(n)), dim3(256), 0, st, a, r, stats, HW, C, coef, ga, n);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_distill_small(const AdnDistillSmall* d, void* stream) {
  ADN_CHECK_ARG(d && d->mean_student && d->centers_student && d->pix_stats && d->terms && d->dmean && d->dcent,
                "adn_distill_small: null operand");
  ADN_CHECK_ARG(d->B > 0 && d->nb > 0 && d->nb <= 256, "adn_distill_small: bad dims");
  ADN_CHECK_ARG(!d->has_teacher || (d->mean_teacher && d->centers_teacher), "adn_distill_small: teacher operands missing");
  SmallParams p;
  p.ms = d->mean_student; p.mt = d->mean_teacher; p.cs = d->centers_student; p.ct = d->centers_teacher;
  for (int i = 0; i < 5; ++i) {
    p.fstats[i] = d->feat_stats[i];
    p.fC[i] = d->feat_channels[i];
    ADN_CHECK_ARG(!d->has_teacher || (p.fstats[i] && p.fC[i] > 0), "adn_distill_small: feature statistics %d missing", i);
  }
  p.pstats = d->pix_stats;
  p.B = d->B; p.nb = d->nb; p.has_teacher = d->has_teacher;
  p.temperature = d->temperature;
  p.lt = d->lambda_task; p.lr = d->lambda_response; p.lf = d->lambda_feature; p.lb = d->lambda_bin; p.ls = d->lambda_sparse;
  p.terms = d->terms; p.dmean = d->dmean; p.dcent = d->dcent;
  hipLaunchKernelGGL(distill_small_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}
