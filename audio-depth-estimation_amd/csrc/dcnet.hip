// Memory-bound kernels of the DoubleConv U-Net family (RGBDepthNet / binaural / AdaBins decoders):
// MaxPool2d(2), bilinear x2 upsample (align_corners=True) + pad, ReLU+BatchNorm backward statistics,
// the 1x1 single-channel depth head, and the L1 + total-variation depth loss.
// All activations NHWC in dtype T (f32 / bf16); 8 channels (16/32 bytes) per thread when C % 8 == 0,
// scalar otherwise.  HBM roofline kernels: every tensor is read/written exactly once per pass.
#include "adn_common.h"
#include "mx8_quant.h"
#include "epilogue.h"

namespace {

constexpr int kMaxBlocks = 8192;
inline unsigned blocks_for(int64_t n, int per_block = 256) {
  int64_t b = adn_cdiv(n, per_block);
  if (b > kMaxBlocks) b = kMaxBlocks;
  if (b < 1) b = 1;
  return (unsigned)b;
}

template <typename T, int V>
__device__ __forceinline__ void loadv(const T* p, int64_t idx, float* f) {
  if constexpr (V == 8) load8<T>(p, idx, f);
  else f[0] = ElemTraits<T>::load(p + idx);
}
template <typename T, int V>
__device__ __forceinline__ void storev(T* p, int64_t idx, const float* f) {
  if constexpr (V == 8) store8<T>(p, idx, f);
  else ElemTraits<T>::store(p + idx, f[0]);
}

// ---------------------------------------------------------------------------------------------------
// MaxPool2d(2) (binaural_attention_model.py:47-50).  src [B][H][W][C] -> dst [B][H/2][W/2][C] (floor).
template <typename T, int V>
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const T* src, T* dst, int B, int H, int W, int C,
                                                           uint2* q8 = nullptr, uint8_t* qsc = nullptr) {
  const int Ho = H >> 1, Wo = W >> 1, ncg = C / V;
  const int64_t work = (int64_t)B * Ho * Wo * ncg;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < work; idx += (int64_t)gridDim.x * 256) {
    const int cg = (int)(idx % ncg);
    int64_t pix = idx / ncg;
    const int ox = (int)(pix % Wo);
    pix /= Wo;
    const int oy = (int)(pix % Ho);
    const int b = (int)(pix / Ho);
    const int64_t base = (((int64_t)b * H + 2 * oy) * W + 2 * ox) * C + cg * V;
    float v0[V], v1[V], v2[V], v3[V], m[V];
    loadv<T, V>(src, base, v0);
    loadv<T, V>(src, base + C, v1);
    loadv<T, V>(src, base + (int64_t)W * C, v2);
    loadv<T, V>(src, base + (int64_t)W * C + C, v3);
#pragma unroll
    for (int k = 0; k < V; ++k) m[k] = fmaxf(fmaxf(v0[k], v1[k]), fmaxf(v2[k], v3[k]));
    storev<T, V>(dst, (((int64_t)b * Ho + oy) * Wo + ox) * C + cg * V, m);
    if constexpr (V == 8 && sizeof(T) == 2) {
      if (q8) {                  // MX-fp8 copy of the bf16 output for the fp8 conv path (the max of bf16 values is exact)
        int byte;
        q8[idx] = mx_quant8(m, byte);
        if ((threadIdx.x & 3) == 0) qsc[idx >> 2] = (uint8_t)byte;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// ConvTranspose2d(k 2, s 2) of the bilinear=False `Up` (binaural_attention_model.py:64-67) is a 1x1 GEMM to
// [B][H][W][4][C] (tap t = 2i + j, then channel) followed by this pixel shuffle:
//   spatial[b][2y + i][2x + j][c] = packed[b][y][x][2i + j][c]        (inverse: the gather of the gradient)
template <typename T, int V>
__global__ __launch_bounds__(256) void pixel_shuffle2_kernel(const T* src, T* dst, int B, int H, int W, int C, int inverse) {
  const int ncg = C / V;
  const int64_t work = (int64_t)B * H * W * 4 * ncg;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < work; idx += (int64_t)gridDim.x * 256) {
    const int cg = (int)(idx % ncg);
    int64_t r = idx / ncg;
    const int t = (int)(r & 3);
    r >>= 2;
    const int x = (int)(r % W);
    r /= W;
    const int y = (int)(r % H);
    const int b = (int)(r / H);
    const int64_t packed = ((((int64_t)b * H + y) * W + x) * 4 + t) * C + cg * V;
    const int64_t spatial = (((int64_t)b * 2 * H + 2 * y + (t >> 1)) * 2 * W + 2 * x + (t & 1)) * C + cg * V;
    float v[V];
    loadv<T, V>(src, inverse ? spatial : packed, v);
    storev<T, V>(dst, inverse ? packed : spatial, v);
  }
}

// ---- fused tail of a gradient's LAST writer (round 2): when the kernel that completes d loss / d y of a ConvBNReLU output is
// a max-pool or upsample backward, it also applies the ReLU mask (y > 0) and accumulates the BatchNorm-backward sums
// (sum g, sum g * xhat) -- the separate relu_bwd_stats pass (read g, y, z; write g) shrinks to one extra read of z (and y).
// Thread t keeps channel group t % ncg for its whole grid-stride loop (the stride is a multiple of 256 and ncg | 256);
// partials [gridDim.x][2][C] for adn_bn_bwd_finalize.
struct TailStats {
  const void* y;        // activated forward tensor of the SOURCE record (mask); max-pool already holds it
  const void* z;        // raw conv output of the source record, nullptr = tail not fused
  const float* mean;
  const float* istd;
  float* partials;
};

template <typename T, int V>
__device__ __forceinline__ void tail_apply(const TailStats& ts, int64_t e, const float* mu, const float* is, const float* yv,
                                           float* g, float* s1, float* s2) {
  float zv[V];
  loadv<T, V>(reinterpret_cast<const T*>(ts.z), e, zv);
#pragma unroll
  for (int k = 0; k < V; ++k) {
    g[k] = yv[k] > 0.f ? g[k] : 0.f;
    s1[k] += g[k];
    s2[k] += g[k] * ((zv[k] - mu[k]) * is[k]);
  }
}
// per-thread channel constants: the thread's channel group is fixed over its grid-stride loop
template <int V>
__device__ __forceinline__ void tail_consts(const TailStats& ts, int ncg, float* mu, float* is) {
#pragma unroll
  for (int k = 0; k < V; ++k) mu[k] = is[k] = 0.f;
  if (ts.z) {
    const int c0 = (int)(((int64_t)blockIdx.x * 256 + threadIdx.x) % ncg) * V;
#pragma unroll
    for (int k = 0; k < V; ++k) {
      mu[k] = ts.mean[c0 + k];
      is[k] = ts.istd[c0 + k];
    }
  }
}

// block reduction of the per-thread sums (V == 8): threads t, t + ncg, t + 2 ncg, ... share a channel group
__device__ __forceinline__ void tail_reduce8(const TailStats& ts, int C, int ncg, const float* s1, const float* s2) {
  __shared__ float tail_red[256 * 16];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    tail_red[threadIdx.x * 16 + k] = s1[k];
    tail_red[threadIdx.x * 16 + 8 + k] = s2[k];
  }
  __syncthreads();
  float* po = ts.partials + (int64_t)blockIdx.x * 2 * C;
  const int per = 256 / ncg;
  for (int t = threadIdx.x; t < ncg * 16; t += 256) {
    const int gl = t >> 4, k = t & 15;
    float sum = 0.f;
    for (int r = 0; r < per; ++r) sum += tail_red[(r * ncg + gl) * 16 + k];
    po[(k >> 3) * C + gl * 8 + (k & 7)] = sum;
  }
}

// Backward: the gradient of a window goes to its FIRST maximum in scan order (strict >), as torch does.
// gsrc (+)= routed gdst; pixels of a trailing odd row/column get 0.
template <typename T, int V>
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const T* gdst, const T* y, T* gsrc, int B, int H, int W,
                                                           int C, int accumulate, TailStats ts = TailStats{}) {
  const int Ho = H >> 1, Wo = W >> 1, Hc = (H + 1) >> 1, Wc = (W + 1) >> 1, ncg = C / V;
  const int64_t work = (int64_t)B * Hc * Wc * ncg;
  float ts1[V], ts2[V], tmu[V], tis[V];
#pragma unroll
  for (int k = 0; k < V; ++k) ts1[k] = ts2[k] = 0.f;
  tail_consts<V>(ts, ncg, tmu, tis);
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < work; idx += (int64_t)gridDim.x * 256) {
    const int cg = (int)(idx % ncg);
    int64_t pix = idx / ncg;
    const int wx = (int)(pix % Wc);
    pix /= Wc;
    const int wy = (int)(pix % Hc);
    const int b = (int)(pix / Hc);
    const int64_t base = (((int64_t)b * H + 2 * wy) * W + 2 * wx) * C + cg * V;
    if (wy < Ho && wx < Wo) {
      const int64_t off[4] = {0, (int64_t)C, (int64_t)W * C, (int64_t)W * C + C};
      float v[4][V], gd[V], o[V];
#pragma unroll
      for (int q = 0; q < 4; ++q) loadv<T, V>(y, base + off[q], v[q]);
      loadv<T, V>(gdst, (((int64_t)b * Ho + wy) * Wo + wx) * C + cg * V, gd);
      int arg[V];
#pragma unroll
      for (int k = 0; k < V; ++k) {
        float m = v[0][k];
        int a = 0;
#pragma unroll
        for (int q = 1; q < 4; ++q)
          if (v[q][k] > m) {
            m = v[q][k];
            a = q;
          }
        arg[k] = a;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (accumulate) loadv<T, V>(gsrc, base + off[q], o);
#pragma unroll
        for (int k = 0; k < V; ++k) o[k] = (accumulate ? o[k] : 0.f) + (arg[k] == q ? gd[k] : 0.f);
        if (ts.z) tail_apply<T, V>(ts, base + off[q], tmu, tis, v[q], o, ts1, ts2);
        storev<T, V>(gsrc, base + off[q], o);
      }
    } else if (ts.z) {
      // trailing odd row / column: no pooled gradient; the pixel keeps its accumulated gradient (or 0), masked
      for (int dy = 0; dy < 2; ++dy)
        for (int dx = 0; dx < 2; ++dx)
          if (2 * wy + dy < H && 2 * wx + dx < W) {
            const int64_t e = base + ((int64_t)dy * W + dx) * C;
            float o[V], yv[V];
            if (accumulate) loadv<T, V>(gsrc, e, o);
            else {
#pragma unroll
              for (int k = 0; k < V; ++k) o[k] = 0.f;
            }
            loadv<T, V>(y, e, yv);
            tail_apply<T, V>(ts, e, tmu, tis, yv, o, ts1, ts2);
            storev<T, V>(gsrc, e, o);
          }
    } else if (!accumulate) {
      float zero[V];
#pragma unroll
      for (int k = 0; k < V; ++k) zero[k] = 0.f;
      for (int dy = 0; dy < 2; ++dy)
        for (int dx = 0; dx < 2; ++dx)
          if (2 * wy + dy < H && 2 * wx + dx < W) storev<T, V>(gsrc, base + ((int64_t)dy * W + dx) * C, zero);
    }
  }
  if constexpr (V == 8) {
    if (ts.z) tail_reduce8(ts, C, ncg, ts1, ts2);
  }
}

// ---------------------------------------------------------------------------------------------------
// nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) followed by F.pad to the skip's size
// (binaural_attention_model.py:62,69-75).  src [B][Hi][Wi][C] -> dst [B][Ho][Wo][C]; the 2Hi x 2Wi image
// sits at (padT, padL), zeros elsewhere.  Source coordinate = o * (in-1)/(out-1) in f32, as torch computes it.
__device__ __forceinline__ void up_src(int o, int in, float r, int& i0, int& ip, float& l1) {
  const float f = r * (float)o;
  i0 = (int)f;
  if (i0 > in - 1) i0 = in - 1;
  ip = i0 < in - 1 ? 1 : 0;
  l1 = fminf(fmaxf(f - (float)i0, 0.f), 1.f);
}

template <typename T, int V>
__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(const T* src, T* dst, int B, int Hi, int Wi, int Ho,
                                                             int Wo, int padT, int padL, int C, uint2* q8 = nullptr,
                                                             uint8_t* qsc = nullptr) {
  const int ncg = C / V;
  const int Hu = 2 * Hi, Wu = 2 * Wi;
  const float rh = Hu > 1 ? (float)(Hi - 1) / (float)(Hu - 1) : 0.f;
  const float rw = Wu > 1 ? (float)(Wi - 1) / (float)(Wu - 1) : 0.f;
  const int64_t work = (int64_t)B * Ho * Wo * ncg;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < work; idx += (int64_t)gridDim.x * 256) {
    const int cg = (int)(idx % ncg);
    int64_t pix = idx / ncg;
    const int ox = (int)(pix % Wo);
    pix /= Wo;
    const int oy = (int)(pix % Ho);
    const int b = (int)(pix / Ho);
    const int uy = oy - padT, ux = ox - padL;
    float o[V];
    if ((unsigned)uy < (unsigned)Hu && (unsigned)ux < (unsigned)Wu) {
      int y0, yp, x0, xp;
      float ly, lx;
      up_src(uy, Hi, rh, y0, yp, ly);
      up_src(ux, Wi, rw, x0, xp, lx);
      const int64_t base = (((int64_t)b * Hi + y0) * Wi + x0) * C + cg * V;
      float a[V], bb[V], c[V], d[V];
      loadv<T, V>(src, base, a);
      loadv<T, V>(src, base + (int64_t)xp * C, bb);
      loadv<T, V>(src, base + (int64_t)yp * Wi * C, c);
      loadv<T, V>(src, base + ((int64_t)yp * Wi + xp) * C, d);
#pragma unroll
      for (int k = 0; k < V; ++k)
        o[k] = (1.f - ly) * ((1.f - lx) * a[k] + lx * bb[k]) + ly * ((1.f - lx) * c[k] + lx * d[k]);
    } else {
#pragma unroll
      for (int k = 0; k < V; ++k) o[k] = 0.f;
    }
    storev<T, V>(dst, idx * V, o);
    if constexpr (V == 8 && sizeof(T) == 2) {
      if (q8) {                  // MX-fp8 copy of exactly what the bf16 tensor holds
#pragma unroll
        for (int k = 0; k < V; ++k) o[k] = bf16_bits_to_f32(f32_to_bf16_bits(o[k]));
        int byte;
        q8[idx] = mx_quant8(o, byte);
        if ((threadIdx.x & 3) == 0) qsc[idx >> 2] = (uint8_t)byte;
      }
    }
  }
}

// Backward as a gather (deterministic, no atomics): source pixel (iy, ix) collects every upsampled pixel
// whose two taps include it, with exactly the forward's weights.
__device__ __forceinline__ int up_candidates(int i, int in, int out, float r, int* o_idx, float* wgt) {
  // outputs o with floor(r*o) in {i-1, i}: r*o in [i-1, i+1)
  int lo = 0, hi = out - 1;
  if (r > 0.f) {
    lo = (int)floorf((float)(i - 1) / r) - 1;
    hi = (int)ceilf((float)(i + 1) / r) + 1;
    if (lo < 0) lo = 0;
    if (hi > out - 1) hi = out - 1;
  }
  int n = 0;
  for (int o = lo; o <= hi && n < 8; ++o) {
    int i0, ip;
    float l1;
    up_src(o, in, r, i0, ip, l1);
    float w = 0.f;
    if (i0 == i) w += 1.f - l1;
    if (ip && i0 + 1 == i) w += l1;
    if (w != 0.f) {
      o_idx[n] = o;
      wgt[n] = w;
      ++n;
    }
  }
  return n;
}

template <typename T, int V>
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const T* gdst, T* gsrc, int B, int Hi, int Wi, int Ho,
                                                             int Wo, int padT, int padL, int C, int accumulate,
                                                             TailStats ts = TailStats{}) {
  const int ncg = C / V;
  float ts1[V], ts2[V], tmu[V], tis[V];
#pragma unroll
  for (int k = 0; k < V; ++k) ts1[k] = ts2[k] = 0.f;
  tail_consts<V>(ts, ncg, tmu, tis);
  const int Hu = 2 * Hi, Wu = 2 * Wi;
  const float rh = Hu > 1 ? (float)(Hi - 1) / (float)(Hu - 1) : 0.f;
  const float rw = Wu > 1 ? (float)(Wi - 1) / (float)(Wu - 1) : 0.f;
  const int64_t work = (int64_t)B * Hi * Wi * ncg;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < work; idx += (int64_t)gridDim.x * 256) {
    const int cg = (int)(idx % ncg);
    int64_t pix = idx / ncg;
    const int ix = (int)(pix % Wi);
    pix /= Wi;
    const int iy = (int)(pix % Hi);
    const int b = (int)(pix / Hi);
    int oy[8], ox[8];
    float wy[8], wx[8];
    const int ny = up_candidates(iy, Hi, Hu, rh, oy, wy);
    const int nx = up_candidates(ix, Wi, Wu, rw, ox, wx);
    float acc[V];
    if (accumulate) loadv<T, V>(gsrc, idx * V, acc);
    else {
#pragma unroll
      for (int k = 0; k < V; ++k) acc[k] = 0.f;
    }
    for (int a = 0; a < ny; ++a) {
      const int dy = oy[a] + padT;
      if ((unsigned)dy >= (unsigned)Ho) continue;
      for (int c = 0; c < nx; ++c) {
        const int dx = ox[c] + padL;
        if ((unsigned)dx >= (unsigned)Wo) continue;
        float g[V];
        loadv<T, V>(gdst, (((int64_t)b * Ho + dy) * Wo + dx) * C + cg * V, g);
        const float w = wy[a] * wx[c];
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] += w * g[k];
      }
    }
    if (ts.z) {
      float yv[V];
      loadv<T, V>(reinterpret_cast<const T*>(ts.y), idx * V, yv);
      tail_apply<T, V>(ts, idx * V, tmu, tis, yv, acc, ts1, ts2);
    }
    storev<T, V>(gsrc, idx * V, acc);
  }
  if constexpr (V == 8) {
    if (ts.z) tail_reduce8(ts, C, ncg, ts1, ts2);
  }
}

// ---------------------------------------------------------------------------------------------------
// ReLU + BatchNorm backward, pass 1: g <- g * (y > 0) in place, per-channel partial sums of g and g*xhat
// (xhat = (z - mean) * istd) into partials [gridDim.x][2][C] for adn_bn_bwd_finalize.
template <typename T>
__global__ __launch_bounds__(256) void relu_bwd_stats_kernel(T* g, const T* y, const T* z, const float* mean,
                                                             const float* istd, int64_t pixels, int C,
                                                             float* partials) {
  __shared__ float red[256 * 16];
  const int64_t rows_per_block = (pixels + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < pixels ? r0 + rows_per_block : pixels;
  float* po = partials + (int64_t)blockIdx.x * 2 * C;
  if ((C & 7) == 0) {
    const int ncg = C >> 3;
    // column groups in passes of at most 256 (C <= 2048 needs one pass)
    for (int cg0 = 0; cg0 < ncg; cg0 += 256) {
      const int span = ncg - cg0 < 256 ? ncg - cg0 : 256;
      const int rpi = 256 / span;                      // rows per iteration
      const int r = threadIdx.x / span, cgl = threadIdx.x - r * span;
      const int c = (cg0 + cgl) * 8;
      float s1[8], s2[8], mu[8], is[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) s1[k] = s2[k] = 0.f;
      if (r < rpi) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          mu[k] = mean[c + k];
          is[k] = istd[c + k];
        }
        for (int64_t row = r0 + r; row < r1; row += rpi) {
          const int64_t e = row * C + c;
          float gv[8], yv[8], zv[8];
          load8<T>(g, e, gv);
          load8<T>(y, e, yv);
          load8<T>(z, e, zv);
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            gv[k] = yv[k] > 0.f ? gv[k] : 0.f;
            s1[k] += gv[k];
            s2[k] += gv[k] * ((zv[k] - mu[k]) * is[k]);
          }
          store8<T>(g, e, gv);
        }
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        red[threadIdx.x * 16 + k] = s1[k];
        red[threadIdx.x * 16 + 8 + k] = s2[k];
      }
      __syncthreads();
      // thread t < span*16 sums slot (cgl = t / 16, k = t % 16) over the rpi row groups
      for (int t = threadIdx.x; t < span * 16; t += 256) {
        const int gl = t >> 4, k = t & 15;
        float s = 0.f;
        for (int rr = 0; rr < rpi; ++rr) s += red[(rr * span + gl) * 16 + k];
        const int cc = (cg0 + gl) * 8 + (k & 7);
        po[(k >> 3) * C + cc] = s;
      }
    }
  } else {
    for (int c = threadIdx.x; c < C; c += 256) {
      const float mu = mean[c], is = istd[c];
      float s1 = 0.f, s2 = 0.f;
      for (int64_t row = r0; row < r1; ++row) {
        const int64_t e = row * C + c;
        float gv = ElemTraits<T>::load(g + e);
        gv = ElemTraits<T>::load(y + e) > 0.f ? gv : 0.f;
        s1 += gv;
        s2 += gv * ((ElemTraits<T>::load(z + e) - mu) * is);
        ElemTraits<T>::store(g + e, gv);
      }
      po[c] = s1;
      po[C + c] = s2;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// 1x1 conv to ONE channel + output activation (outc + clamp / sigmoid*max_depth:
// rgb_depth_model.py:195-209, binaural_attention_model.py:330-337).  One pixel per LPP-lane group.
//   act 0: out = clamp(z, 0, max_depth)      act 1: out = clamp(sigmoid(z) * max_depth, 0, max_depth)
//   act 2: out = tanh(z) * max_depth         act 3: out = z (the clamp follows the final resize, adn_clamp_range)
template <typename T, int V>
__global__ __launch_bounds__(256) void head1x1_fwd_kernel(const T* x, const float* w, const float* bias,
                                                          int64_t pixels, int C, int lpp, int act, float maxd,
                                                          float* zpre, float* out) {
  const int lig = threadIdx.x & (lpp - 1);
  const int gpb = 256 / lpp;
  const int grp = threadIdx.x / lpp;
  const float b = bias ? bias[0] : 0.f;
  for (int64_t p0 = (int64_t)blockIdx.x * gpb; p0 < pixels; p0 += (int64_t)gridDim.x * gpb) {
    const int64_t pix = p0 + grp;
    float s = 0.f;
    if (pix < pixels) {
      for (int c = lig * V; c < C; c += lpp * V) {
        float xv[V];
        loadv<T, V>(x, pix * C + c, xv);
#pragma unroll
        for (int k = 0; k < V; ++k) s += xv[k] * w[c + k];
      }
    }
    for (int o = lpp >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (pix < pixels && lig == 0) {
      const float zz = s + b;
      zpre[pix] = zz;
      if (act == 2) {
        out[pix] = tanhf(zz) * maxd;
      } else if (act == 3) {
        out[pix] = zz;
      } else {
        float o_ = act == 1 ? maxd / (1.f + __expf(-zz)) : zz;
        out[pix] = fminf(fmaxf(o_, 0.f), maxd);
      }
    }
  }
}

// torch.clamp(x, 0, max_depth) after the final resize (rgb_depth_model.py:209) and its backward (the gradient passes
// on the closed interval, as torch.clamp does)
__global__ __launch_bounds__(256) void clamp_range_kernel(const float* x, const float* g, int64_t n, float maxd, float* out) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const float v = x[e];
    out[e] = g ? ((v >= 0.f && v <= maxd) ? g[e] : 0.f) : fminf(fmaxf(v, 0.f), maxd);
  }
}

constexpr int kHeadIt = 4;   // channel passes per lane (C <= lpp * V * kHeadIt)

// Backward: dz = gout * act'(z); gx[pix][c] = dz * w[c]; partial sums of dz * x[pix][c] (dW) and dz (db)
// into partials [gridDim.x][C + 1].
template <typename T, int V>
__global__ __launch_bounds__(256) void head1x1_bwd_kernel(const float* gout, const float* zpre, const T* x,
                                                          const float* w, int64_t pixels, int C, int lpp, int act,
                                                          float maxd, T* gx, float* partials) {
  __shared__ float red[256 * (kHeadIt * V + 1)];
  constexpr int SL = kHeadIt * V + 1;
  const int lig = threadIdx.x & (lpp - 1);
  const int gpb = 256 / lpp;
  const int grp = threadIdx.x / lpp;
  float acc[kHeadIt * V];
  float accb = 0.f;
#pragma unroll
  for (int k = 0; k < kHeadIt * V; ++k) acc[k] = 0.f;
  for (int64_t p0 = (int64_t)blockIdx.x * gpb; p0 < pixels; p0 += (int64_t)gridDim.x * gpb) {
    const int64_t pix = p0 + grp;
    if (pix >= pixels) continue;
    const float zz = zpre[pix];
    float d;
    if (act == 1) {
      const float s = 1.f / (1.f + __expf(-zz));
      d = maxd * s * (1.f - s);
    } else if (act == 2) {
      const float t = tanhf(zz);
      d = maxd * (1.f - t * t);
    } else if (act == 3) {
      d = 1.f;
    } else {
      d = (zz >= 0.f && zz <= maxd) ? 1.f : 0.f;
    }
    const float dz = gout[pix] * d;
    if (lig == 0) accb += dz;
#pragma unroll
    for (int it = 0; it < kHeadIt; ++it) {
      const int c = (it * lpp + lig) * V;
      if (c < C) {
        float xv[V], gv[V];
        loadv<T, V>(x, pix * C + c, xv);
#pragma unroll
        for (int k = 0; k < V; ++k) {
          acc[it * V + k] += dz * xv[k];
          gv[k] = dz * w[c + k];
        }
        storev<T, V>(gx, pix * C + c, gv);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < kHeadIt * V; ++k) red[threadIdx.x * SL + k] = acc[k];
  red[threadIdx.x * SL + kHeadIt * V] = accb;
  __syncthreads();
  float* po = partials + (int64_t)blockIdx.x * (C + 1);
  for (int c = threadIdx.x; c <= C; c += 256) {
    float s = 0.f;
    if (c == C) {
      for (int g2 = 0; g2 < gpb; ++g2) s += red[(g2 * lpp) * SL + kHeadIt * V];
    } else {
      const int cgi = c / V, k = c - cgi * V;
      const int it = cgi / lpp, l = cgi - it * lpp;
      for (int g2 = 0; g2 < gpb; ++g2) s += red[(g2 * lpp + l) * SL + it * V + k];
    }
    po[c] = s;
  }
}

// out[j] = sum_p partials[p][j], f64 accumulation, one wave per column.
__global__ __launch_bounds__(256) void colsum_kernel(const float* partials, int64_t P, int n, float* out0, int n0,
                                                     float* out1) {
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (j >= n) return;
  double s = 0.0;
  for (int64_t r = lane; r < P; r += 64) s += (double)partials[r * n + j];
  s = wave_sum_d(s);
  if (lane == 0) {
    if (j < n0) out0[j] = (float)s;
    else if (out1) out1[j - n0] = (float)s;
  }
}

// ---------------------------------------------------------------------------------------------------
// DepthLoss of train_rgb_depth.py:43-87: lambda_l1 * mean|p - g| + lambda_smooth * (mean|dx p| + mean|dy p|),
// unmasked.  stats (f64[4]) = [sum|p-g|, sum|dx|, sum|dy|, 0].
__global__ __launch_bounds__(256) void l1tv_stats_kernel(const float* pred, const float* gt, int64_t n, int H, int W,
                                                         double* partial) {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int x = (int)(e % W);
    const int y = (int)((e / W) % H);
    const float p = pred[e];
    s0 += (double)fabsf(p - gt[e]);
    if (x < W - 1) s1 += (double)fabsf(p - pred[e + 1]);
    if (y < H - 1) s2 += (double)fabsf(p - pred[e + W]);
  }
  __shared__ double sm[3][4];
  s0 = wave_sum_d(s0);
  s1 = wave_sum_d(s1);
  s2 = wave_sum_d(s2);
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sm[0][wv] = s0;
    sm[1][wv] = s1;
    sm[2][wv] = s2;
  }
  __syncthreads();
  if (threadIdx.x < 3)
    partial[(int64_t)blockIdx.x * 3 + threadIdx.x] =
        sm[threadIdx.x][0] + sm[threadIdx.x][1] + sm[threadIdx.x][2] + sm[threadIdx.x][3];
}

__global__ __launch_bounds__(64) void l1tv_stats_final_kernel(const double* partial, int nb, double* stats) {
  for (int q = 0; q < 3; ++q) {
    double s = 0.0;
    for (int r = threadIdx.x; r < nb; r += 64) s += partial[(int64_t)r * 3 + q];
    s = wave_sum_d(s);
    if (threadIdx.x == 0) stats[q] = s;
  }
  if (threadIdx.x == 0) stats[3] = 0.0;
}

__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

// counts: N = n * world, Nx = B*H*(W-1) * world, Ny = B*(H-1)*W * world (world = data-parallel replicas whose
// stats were summed).  loss_out optional (block 0 writes it).
__global__ __launch_bounds__(256) void l1tv_finish_kernel(const float* pred, const float* gt, int64_t n, int H, int W,
                                                          const double* stats, double inv_n, double inv_nx,
                                                          double inv_ny, float l1w, float sw, float* loss_out,
                                                          float* grad) {
  if (blockIdx.x == 0 && threadIdx.x == 0 && loss_out)
    loss_out[0] = (float)((double)l1w * stats[0] * inv_n + (double)sw * (stats[1] * inv_nx + stats[2] * inv_ny));
  const float c0 = (float)((double)l1w * inv_n), cx = (float)((double)sw * inv_nx), cy = (float)((double)sw * inv_ny);
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int x = (int)(e % W);
    const int y = (int)((e / W) % H);
    const float p = pred[e];
    float g = c0 * sgn(p - gt[e]);
    if (x < W - 1) g += cx * sgn(p - pred[e + 1]);
    if (x > 0) g -= cx * sgn(pred[e - 1] - p);
    if (y < H - 1) g += cy * sgn(p - pred[e + W]);
    if (y > 0) g -= cy * sgn(pred[e - W] - p);
    grad[e] = g;
  }
}

inline int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

}  // namespace

#define ADN_DISPATCH_V(KERNEL, TYPE, grid, st, ...)                                              \
  do {                                                                                           \
    if ((C & 7) == 0) hipLaunchKernelGGL((KERNEL<TYPE, 8>), grid, dim3(256), 0, st, __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<TYPE, 1>), grid, dim3(256), 0, st, __VA_ARGS__);              \
  } while (0)

extern "C" int adn_maxpool2_fwd(const void* src, void* dst, int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype,
                                void* stream) {
  ADN_CHECK_ARG(src && dst && B > 0 && H > 1 && W > 1 && C > 0, "adn_maxpool2_fwd: bad arguments");
  ADN_CHECK_ARG(dtype == ADN_F32 || dtype == ADN_BF16, "adn_maxpool2_fwd: bad dtype %d", dtype);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t work = (int64_t)B * (H / 2) * (W / 2) * ((C & 7) == 0 ? C / 8 : C);
  if (dtype == ADN_BF16) {
    auto s = reinterpret_cast<const uint16_t*>(src);
    auto d = reinterpret_cast<uint16_t*>(dst);
    ADN_DISPATCH_V(maxpool2_fwd_kernel, uint16_t, dim3(blocks_for(work)), st, s, d, B, H, W, C);
  } else {
    auto s = reinterpret_cast<const float*>(src);
    auto d = reinterpret_cast<float*>(dst);
    ADN_DISPATCH_V(maxpool2_fwd_kernel, float, dim3(blocks_for(work)), st, s, d, B, H, W, C);
  }
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

// bf16 forward kernels that also write the MX-fp8 copy of their output (C % 32 == 0), for the fp8 conv path
extern "C" int adn_maxpool2_fwd_mx8(const void* src, void* dst, int32_t B, int32_t H, int32_t W, int32_t C, void* out8,
                                    void* out_scales, void* stream) {
  ADN_CHECK_ARG(src && dst && out8 && out_scales && B > 0 && H > 1 && W > 1 && C > 0 && C % 32 == 0,
                "adn_maxpool2_fwd_mx8: bad arguments (C=%d must be a multiple of 32)", C);
  const int64_t work = (int64_t)B * (H / 2) * (W / 2) * (C / 8);
  hipLaunchKernelGGL((maxpool2_fwd_kernel<uint16_t, 8>), dim3(blocks_for(work)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const uint16_t*>(src), reinterpret_cast<uint16_t*>(dst), B, H, W, C,
                     reinterpret_cast<uint2*>(out8), reinterpret_cast<uint8_t*>(out_scales));
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_upsample2x_fwd_mx8(const void* src, void* dst, int32_t B, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo,
                                      int32_t C, void* out8, void* out_scales, void* stream) {
  ADN_CHECK_ARG(src && dst && out8 && out_scales && B > 0 && Hi > 0 && Wi > 0 && C > 0 && C % 32 == 0,
                "adn_upsample2x_fwd_mx8: bad arguments (C=%d must be a multiple of 32)", C);
  ADN_CHECK_ARG(Ho >= 2 * Hi && Wo >= 2 * Wi, "adn_upsample2x_fwd_mx8: target %dx%d smaller than 2x source %dx%d", Ho, Wo, Hi, Wi);
  const int padT = (Ho - 2 * Hi) / 2, padL = (Wo - 2 * Wi) / 2;
  const int64_t work = (int64_t)B * Ho * Wo * (C / 8);
  hipLaunchKernelGGL((upsample2x_fwd_kernel<uint16_t, 8>), dim3(blocks_for(work)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const uint16_t*>(src),
                     reinterpret_cast<uint16_t*>(dst), B, Hi, Wi, Ho, Wo, padT, padL, C, reinterpret_cast<uint2*>(out8),
                     reinterpret_cast<uint8_t*>(out_scales));
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

// Grid of the tail-fused backward kernels for `work` thread items (= partial rows they write); 0 when the channel count does not
// allow the fusion (needs C % 8 == 0 and C / 8 a divisor of 256).
extern "C" int64_t adn_tail_stats_blocks(int64_t work, int32_t C) {
  if (C <= 0 || (C & 7) != 0 || 256 % (C / 8) != 0 || work <= 0) return 0;
  int64_t b = adn_cdiv(work, 256);
  if (b > 2048) b = 2048;
  return b;
}

extern "C" int adn_maxpool2_bwd_tail(const void* gdst, const void* y, void* gsrc, int32_t B, int32_t H, int32_t W, int32_t C,
                                     int32_t accumulate, int32_t dtype, const void* z, const float* mean, const float* istd,
                                     float* partials, void* stream) {
  ADN_CHECK_ARG(gdst && y && gsrc && z && mean && istd && partials && B > 0 && H > 1 && W > 1, "adn_maxpool2_bwd_tail: bad arguments");
  ADN_CHECK_ARG(dtype == ADN_F32 || dtype == ADN_BF16, "adn_maxpool2_bwd_tail: bad dtype %d", dtype);
  const int64_t work = (int64_t)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / 8);
  const int64_t blocks = adn_tail_stats_blocks(work, C);
  ADN_CHECK_ARG(blocks > 0, "adn_maxpool2_bwd_tail: C=%d not supported", C);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  TailStats ts{y, z, mean, istd, partials};
  if (dtype == ADN_BF16)
    hipLaunchKernelGGL((maxpool2_bwd_kernel<uint16_t, 8>), dim3((unsigned)blocks), dim3(256), 0, st,
                       reinterpret_cast<const uint16_t*>(gdst), reinterpret_cast<const uint16_t*>(y),
                       reinterpret_cast<uint16_t*>(gsrc), B, H, W, C, accumulate, ts);
  else
    hipLaunchKernelGGL((maxpool2_bwd_kernel<float, 8>), dim3((unsigned)blocks), dim3(256), 0, st,
                       reinterpret_cast<const float*>(gdst), reinterpret_cast<const float*>(y), reinterpret_cast<float*>(gsrc), B,
                       H, W, C, accumulate, ts);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_upsample2x_bwd_tail(const void* gdst, void* gsrc, int32_t B, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo,
                                       int32_t C, int32_t accumulate, int32_t dtype, const void* y, const void* z,
                                       const float* mean, const float* istd, float* partials, void* stream) {
  ADN_CHECK_ARG(gdst && gsrc && y && z && mean && istd && partials && B > 0 && Hi > 0 && Wi > 0, "adn_upsample2x_bwd_tail: bad arguments");
  ADN_CHECK_ARG(Ho >= 2 * Hi && Wo >= 2 * Wi, "adn_upsample2x_bwd_tail: target %dx%d smaller than 2x source %dx%d", Ho, Wo, Hi, Wi);
  ADN_CHECK_ARG(dtype == ADN_F32 || dtype == ADN_BF16, "adn_upsample2x_bwd_tail: bad dtype %d", dtype);
  const int padT = (Ho - 2 * Hi) / 2, padL = (Wo - 2 * Wi) / 2;
  const int64_t work = (int64_t)B * Hi * Wi * (C / 8);
  const int64_t blocks = adn_tail_stats_blocks(work, C);
  ADN_CHECK_ARG(blocks > 0, "adn_upsample2x_bwd_tail: C=%d not supported", C);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  TailStats ts{y, z, mean, istd, partials};
  if (dtype == ADN_BF16)
    hipLaunchKernelGGL((upsample2x_bwd_kernel<uint16_t, 8>), dim3((unsigned)blocks), dim3(256), 0, st,
                       reinterpret_cast<const uint16_t*>(gdst), reinterpret_cast<uint16_t*>(gsrc), B, Hi, Wi, Ho, Wo, padT, padL, C,
                       accumulate, ts);
  else
    hipLaunchKernelGGL((upsample2x_bwd_kernel<float, 8>), dim3((unsigned)blocks), dim3(256), 0, st,
                       reinterpret_cast<const float*>(gdst), reinterpret_cast<float*>(gsrc), B, Hi, Wi, Ho, Wo, padT, padL, C,
                       accumulate, ts);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_maxpool2_bwd(const void* gdst, const void* y, void* gsrc, int32_t B, int32_t H, int32_t W,
                                int32_t C, int32_t accumulate, int32_t dtype, void* stream) {
  ADN_CHECK_ARG(gdst && y && gsrc && B > 0 && H > 1 && W > 1 && C > 0, "adn_maxpool2_bwd: bad arguments");
  ADN_CHECK_ARG(dtype == ADN_F32 || dtype == ADN_BF16, "adn_maxpool2_bwd: bad dtype %d", dtype);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t work = (int64_t)B * ((H + 1) / 2) * ((W + 1) / 2) * ((C & 7) == 0 ? C / 8 : C);
  if (dtype == ADN_BF16) {
    auto gd = reinterpret_cast<const uint16_t*>(gdst);
    auto yy = reinterpret_cast<const uint16_t*>(y);
    auto gs = reinterpret_cast<uint16_t*>(gsrc);
    ADN_DISPATCH_V(maxpool2_bwd_kernel, uint16_t, dim3(blocks_for(work)), st, gd, yy, gs, B, H, W, C, accumulate);
  } else {
    auto gd = reinterpret_cast<const float*>(gdst);
    auto yy = reinterpret_cast<const float*>(y);
    auto gs = reinterpret_cast<float*>(gsrc);
    ADN_DISPATCH_V(maxpool2_bwd_kernel, float, dim3(blocks_for(work)), st, gd, yy, gs, B, H, W, C, accumulate);
  }
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_upsample2x_fwd(const void* src, void* dst, int32_t B, int32_t Hi, int32_t Wi, int32_t Ho,
                                  int32_t Wo, int32_t C, int32_t dtype, void* stream) {
  ADN_CHECK_ARG(src && dst && B > 0 && Hi > 0 && Wi > 0 && C > 0, "adn_upsample2x_fwd: bad arguments");
  ADN_CHECK_ARG(Ho >= 2 * Hi && Wo >= 2 * Wi, "adn_upsample2x_fwd: target %dx%d smaller than 2x source %dx%d", Ho, Wo,
                Hi, Wi);
  ADN_CHECK_ARG(dtype == ADN_F32 || dtype == ADN_BF16, "adn_upsample2x_fwd: bad dtype %d", dtype);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int padT = (Ho - 2 * Hi) / 2, padL = (Wo - 2 * Wi) / 2;      // F.pad(diff // 2, diff - diff // 2)
  const int64_t work = (int64_t)B * Ho * Wo * ((C & 7) == 0 ? C / 8 : C);
  if (dtype == ADN_BF16) {
    auto s = reinterpret_cast<const uint16_t*>(src);
    auto d = reinterpret_cast<uint16_t*>(dst);
    ADN_DISPATCH_V(upsample2x_fwd_kernel, uint16_t, dim3(blocks_for(work)), st, s, d, B, Hi, Wi, Ho, Wo, padT, padL, C);
  } else {
    auto s = reinterpret_cast<const float*>(src);
    auto d = reinterpret_cast<float*>(dst);
    ADN_DISPATCH_V(upsample2x_fwd_kernel, float, dim3(blocks_for(work)), st, s, d, B, Hi, Wi, Ho, Wo, padT, padL, C);
  }
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_upsample2x_bwd(const void* gdst, void* gsrc, int32_t B, int32_t Hi, int32_t Wi, int32_t Ho,
                                  int32_t Wo, int32_t C, int32_t accumulate, int32_t dtype, void* stream) {
  ADN_CHECK_ARG(gdst && gsrc && B > 0 && Hi > 0 && Wi > 0 && C > 0, "adn_upsample2x_bwd: bad arguments");
  ADN_CHECK_ARG(Ho >= 2 * Hi && Wo >= 2 * Wi, "adn_upsample2x_bwd: target %dx%d smaller than 2x source %dx%d", Ho, Wo,
                Hi, Wi);
  ADN_CHECK_ARG(dtype == ADN_F32 || dtype == ADN_BF16, "adn_upsample2x_bwd: bad dtype %d", dtype);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int padT = (Ho - 2 * Hi) / 2, padL = (Wo - 2 * Wi) / 2;
  const int64_t work = (int64_t)B * Hi * Wi * ((C & 7) == 0 ? C / 8 : C);
  if (dtype == ADN_BF16) {
    auto gd = reinterpret_cast<const uint16_t*>(gdst);
    auto gs = reinterpret_cast<uint16_t*>(gsrc);
    ADN_DISPATCH_V(upsample2x_bwd_kernel, uint16_t, dim3(blocks_for(work)), st, gd, gs, B, Hi, Wi, Ho, Wo, padT, padL, C,
                    accumulate);
  } else {
    auto gd = reinterpret_cast<const float*>(gdst);
    auto gs = reinterpret_cast<float*>(gsrc);
    ADN_DISPATCH_V(upsample2x_bwd_kernel, float, dim3(blocks_for(work)), st, gd, gs, B, Hi, Wi, Ho, Wo, padT, padL, C,
                    accumulate);
  }
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_pixel_shuffle2(const void* src, void* dst, int32_t B, int32_t H, int32_t W, int32_t C, int32_t inverse,
                                  int32_t dtype, void* stream) {
  ADN_CHECK_ARG(src && dst && B > 0 && H > 0 && W > 0 && C > 0, "adn_pixel_shuffle2: bad arguments");
  ADN_CHECK_ARG(dtype == ADN_F32 || dtype == ADN_BF16, "adn_pixel_shuffle2: bad dtype %d", dtype);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t work = (int64_t)B * H * W * 4 * ((C & 7) == 0 ? C / 8 : C);
  if (dtype == ADN_BF16) {
    auto sp = reinterpret_cast<const uint16_t*>(src);
    auto dp = reinterpret_cast<uint16_t*>(dst);
    ADN_DISPATCH_V(pixel_shuffle2_kernel, uint16_t, dim3(blocks_for(work)), st, sp, dp, B, H, W, C, inverse);
  } else {
    auto sp = reinterpret_cast<const float*>(src);
    auto dp = reinterpret_cast<float*>(dst);
    ADN_DISPATCH_V(pixel_shuffle2_kernel, float, dim3(blocks_for(work)), st, sp, dp, B, H, W, C, inverse);
  }
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_clamp_range(const float* x, const float* g, int64_t n, float max_depth, float* out, void* stream) {
  ADN_CHECK_ARG(x && out && n > 0, "adn_clamp_range: bad arguments");
  hipLaunchKernelGGL(clamp_range_kernel, dim3(blocks_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, g, n,
                     max_depth, out);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int64_t adn_relu_bwd_stats_num_partials(int64_t pixels, int32_t C) {
  if (pixels <= 0 || C <= 0) return -1;
  int64_t p = adn_cdiv(pixels, 256);
  if (p > 2048) p = 2048;
  return p;
}

extern "C" int adn_relu_bwd_stats(void* g, const void* y, const void* z, const float* mean, const float* istd,
                                  int64_t pixels, int32_t C, int32_t dtype, float* partials, void* stream) {
  ADN_CHECK_ARG(g && y && z && mean && istd && partials && pixels > 0 && C > 0, "adn_relu_bwd_stats: bad arguments");
  ADN_CHECK_ARG(dtype == ADN_F32 || dtype == ADN_BF16, "adn_relu_bwd_stats: bad dtype %d", dtype);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const unsigned P = (unsigned)adn_relu_bwd_stats_num_partials(pixels, C);
  if (dtype == ADN_BF16)
    hipLaunchKernelGGL((relu_bwd_stats_kernel<uint16_t>), dim3(P), dim3(256), 0, st, reinterpret_cast<uint16_t*>(g),
                       reinterpret_cast<const uint16_t*>(y), reinterpret_cast<const uint16_t*>(z), mean, istd, pixels,
                       C, partials);
  else
    hipLaunchKernelGGL((relu_bwd_stats_kernel<float>), dim3(P), dim3(256), 0, st, reinterpret_cast<float*>(g),
                       reinterpret_cast<const float*>(y), reinterpret_cast<const float*>(z), mean, istd, pixels, C,
                       partials);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

static int head_lpp(int C, int V) {
  int l = next_pow2((int)adn_cdiv(C, V));
  if (l > 64) l = 64;
  return l;
}

extern "C" int adn_head1x1_fwd(const void* x, const float* w, const float* bias, int64_t pixels, int32_t C,
                               int32_t dtype, int32_t act, float max_depth, float* zpre, float* out, void* stream) {
  ADN_CHECK_ARG(x && w && zpre && out && pixels > 0 && C > 0, "adn_head1x1_fwd: bad arguments");
  ADN_CHECK_ARG(dtype == ADN_F32 || dtype == ADN_BF16, "adn_head1x1_fwd: bad dtype %d", dtype);
  ADN_CHECK_ARG(act >= 0 && act <= 3, "adn_head1x1_fwd: bad act %d", act);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int V = (C & 7) == 0 ? 8 : 1;
  const int lpp = head_lpp(C, V);
  const dim3 grid(blocks_for(pixels, 256 / lpp));
  if (dtype == ADN_BF16) {
    auto xx = reinterpret_cast<const uint16_t*>(x);
    ADN_DISPATCH_V(head1x1_fwd_kernel, uint16_t, grid, st, xx, w, bias, pixels, C, lpp, act, max_depth, zpre, out);
  } else {
    auto xx = reinterpret_cast<const float*>(x);
    ADN_DISPATCH_V(head1x1_fwd_kernel, float, grid, st, xx, w, bias, pixels, C, lpp, act, max_depth, zpre, out);
  }
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int64_t adn_head1x1_bwd_workspace_bytes(int64_t pixels, int32_t C) {
  if (pixels <= 0 || C <= 0) return -1;
  const int V = (C & 7) == 0 ? 8 : 1;
  const int lpp = head_lpp(C, V);
  int64_t nb = blocks_for(pixels, 256 / lpp);
  if (nb > 1024) nb = 1024;
  return nb * (C + 1) * (int64_t)sizeof(float);
}

extern "C" int adn_head1x1_bwd(const float* gout, const float* zpre, const void* x, const float* w, int64_t pixels,
                               int32_t C, int32_t dtype, int32_t act, float max_depth, void* gx, float* dw, float* db,
                               void* workspace, int64_t workspace_bytes, void* stream) {
  ADN_CHECK_ARG(gout && zpre && x && w && gx && dw && pixels > 0 && C > 0, "adn_head1x1_bwd: bad arguments");
  ADN_CHECK_ARG(dtype == ADN_F32 || dtype == ADN_BF16, "adn_head1x1_bwd: bad dtype %d", dtype);
  ADN_CHECK_ARG(act >= 0 && act <= 3, "adn_head1x1_bwd: bad act %d", act);
  const int V = (C & 7) == 0 ? 8 : 1;
  const int lpp = head_lpp(C, V);
  ADN_CHECK_ARG(C <= lpp * V * kHeadIt, "adn_head1x1_bwd: C = %d too large (max %d)", C, lpp * V * kHeadIt);
  const int64_t need = adn_head1x1_bwd_workspace_bytes(pixels, C);
  ADN_CHECK_ARG(workspace && workspace_bytes >= need, "adn_head1x1_bwd: workspace too small (%lld < %lld)",
                (long long)workspace_bytes, (long long)need);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const unsigned nb = (unsigned)(need / ((C + 1) * sizeof(float)));
  float* partials = reinterpret_cast<float*>(workspace);
  if (dtype == ADN_BF16) {
    auto xx = reinterpret_cast<const uint16_t*>(x);
    auto gg = reinterpret_cast<uint16_t*>(gx);
    ADN_DISPATCH_V(head1x1_bwd_kernel, uint16_t, dim3(nb), st, gout, zpre, xx, w, pixels, C, lpp, act, max_depth, gg, partials);
  } else {
    auto xx = reinterpret_cast<const float*>(x);
    auto gg = reinterpret_cast<float*>(gx);
    ADN_DISPATCH_V(head1x1_bwd_kernel, float, dim3(nb), st, gout, zpre, xx, w, pixels, C, lpp, act, max_depth, gg, partials);
  }
  ADN_CHECK_LAUNCH();
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)adn_cdiv(C + 1, 4)), dim3(256), 0, st, partials, (int64_t)nb, C + 1,
                     dw, C, db);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int64_t adn_l1tv_workspace_bytes(int64_t n) {
  if (n <= 0) return -1;
  int64_t nb = adn_cdiv(n, 256 * 8);
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  return nb * 3 * (int64_t)sizeof(double);
}

extern "C" int adn_l1tv_stats(const float* pred, const float* gt, int32_t B, int32_t H, int32_t W, double* stats,
                              void* workspace, int64_t workspace_bytes, void* stream) {
  ADN_CHECK_ARG(pred && gt && stats && B > 0 && H > 1 && W > 1, "adn_l1tv_stats: bad arguments");
  const int64_t n = (int64_t)B * H * W;
  const int64_t need = adn_l1tv_workspace_bytes(n);
  ADN_CHECK_ARG(workspace && workspace_bytes >= need, "adn_l1tv_stats: workspace too small");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nb = (int)(need / (3 * sizeof(double)));
  hipLaunchKernelGGL(l1tv_stats_kernel, dim3(nb), dim3(256), 0, st, pred, gt, n, H, W,
                     reinterpret_cast<double*>(workspace));
  ADN_CHECK_LAUNCH();
  hipLaunchKernelGGL(l1tv_stats_final_kernel, dim3(1), dim3(64), 0, st, reinterpret_cast<const double*>(workspace), nb,
                     stats);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_l1tv_finish(const float* pred, const float* gt, int32_t B, int32_t H, int32_t W,
                               const double* stats, int32_t replicas, float lambda_l1, float lambda_smooth,
                               float* loss_out, float* grad, void* stream) {
  ADN_CHECK_ARG(pred && gt && stats && grad && B > 0 && H > 1 && W > 1 && replicas > 0,
                "adn_l1tv_finish: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t n = (int64_t)B * H * W;
  const double inv_n = 1.0 / ((double)n * replicas);
  const double inv_nx = 1.0 / ((double)B * H * (W - 1) * replicas);
  const double inv_ny = 1.0 / ((double)B * (H - 1) * W * replicas);
  hipLaunchKernelGGL(l1tv_finish_kernel, dim3(blocks_for(n)), dim3(256), 0, st, pred, gt, n, H, W, stats, inv_n,
                     inv_nx, inv_ny, lambda_l1, lambda_smooth, loss_out, grad);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}
