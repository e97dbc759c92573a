// Memory-bound helpers around the implicit GEMMs: weight packing, NCHW<->NHWC boundary
// conversion, BatchNorm finalize / apply passes.  All are streaming kernels (HBM roofline),
// 16-byte accesses where the layout allows, grid capped at ~2048 blocks with grid-stride loops.
#include "epilogue.h"
#include "mx8_quant.h"

namespace {

constexpr int kMaxBlocks = 4096;
inline unsigned blocks_for(int64_t n, int per_block = 256) {
  int64_t b = adn_cdiv(n, per_block);
  if (b > kMaxBlocks) b = kMaxBlocks;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// BatchNorm apply kernels (bn_act / bn_bwd_apply): 1024 workgroups -- a thread keeps its 8 channels' coefficients in
// registers over >= 4x more 16-byte chunks than at 4096 (bn_bwd_apply of the headline step 133 -> 115 us, of RGBDepthNet
// 1.48 -> 1.02 ms together with the hoisted coefficient loads; 8192: 134 / 1.08, 2048: 122 / 1.02).  ADN_BN_BLOCKS overrides.
inline unsigned bn_blocks_for(int64_t n) {
  static const int cap = getenv("ADN_BN_BLOCKS") ? atoi(getenv("ADN_BN_BLOCKS")) : 1024;
  int64_t b = adn_cdiv(n, 256);
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// master [X][4][4][Y] f32 -> s2 [X][16][Ypad] (cast, zero padded channels): coalesced both sides
template <typename T>
__global__ __launch_bounds__(256) void pack_s2_kernel(const float* master, int X, int Y, int Ypad, T* s2) {
  const int64_t np = (int64_t)X * 16 * Ypad;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < np; e += (int64_t)gridDim.x * 256) {
    const int y = (int)(e % Ypad);
    const int64_t xt = e / Ypad;
    ElemTraits<T>::store(s2 + e, y < Y ? master[xt * Y + y] : 0.0f);
  }
}

// master [X][16 taps][Y] f32 -> t2 [4 phases][Y][4 taps][X]: a per-tap [X][Y] -> [Y][X] transpose through
// a 32x33 LDS tile so that both the reads (y fastest) and the writes (x fastest) are coalesced.
template <typename T>
__global__ __launch_bounds__(256) void pack_t2_kernel(const float* master, int X, int Y, T* t2) {
  __shared__ float tile[32][33];
  const int tap = blockIdx.y;                       // kh*4 + kw
  const int kh = tap >> 2, kw = tap & 3;
  // inverse of adn_t2_kh: kh=1 -> (ph 0, t 0), kh=3 -> (0,1), kh=0 -> (1,0), kh=2 -> (1,1)
  const int ph = (kh & 1) ? 0 : 1, ty = (kh == 3 || kh == 2) ? 1 : 0;
  const int pw = (kw & 1) ? 0 : 1, tx = (kw == 3 || kw == 2) ? 1 : 0;
  const int phase = ph * 2 + pw, t = ty * 2 + tx;
  const int tiles_y = (Y + 31) / 32;
  const int x0 = (blockIdx.x / tiles_y) * 32, y0 = (blockIdx.x % tiles_y) * 32;
  const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int x = x0 + ly + 8 * r, y = y0 + lx;
    tile[ly + 8 * r][lx] = (x < X && y < Y) ? master[((int64_t)x * 16 + tap) * Y + y] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int y = y0 + ly + 8 * r, x = x0 + lx;
    if (x < X && y < Y) ElemTraits<T>::store(t2 + (((int64_t)phase * Y + y) * 4 + t) * X + x, tile[lx][ly + 8 * r]);
  }
}

// one thread per pixel: C coalesced plane reads, one contiguous Cpad-element write
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* src, T* dst, int B, int C, int Cpad,
                                                           int64_t HW, int64_t Ctot) {
  const int64_t npix = (int64_t)B * HW;
  for (int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x; pix < npix; pix += (int64_t)gridDim.x * 256) {
    const int64_t b = pix / HW, hw = pix - b * HW;
    for (int c0 = 0; c0 < Cpad; c0 += 8) {
      float f[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) f[k] = (c0 + k < C) ? src[(b * Ctot + c0 + k) * HW + hw] : 0.0f;
      if (c0 + 8 <= Cpad && (Cpad & 7) == 0) {
        store8<T>(dst, pix * Cpad + c0, f);
      } else {
        for (int k = 0; k < 8 && c0 + k < Cpad; ++k) ElemTraits<T>::store(dst + pix * Cpad + c0 + k, f[k]);
      }
    }
  }
}

// every layer's T2 pack in one launch: blockIdx -> layer by a linear scan of the (tiny) table.  64 x 64 tiles: 256-byte
// reads of the f32 master rows, 128-byte writes of the bf16 pack rows.
template <typename S, typename T>
__global__ __launch_bounds__(256) void pack_t2_multi_kernel(const S* flat_master, const int64_t* table, int layers,
                                                            T* t2_base) {
  __shared__ float tile[64][65];
  int l = 0;
  while (l + 1 < layers && (int64_t)blockIdx.x >= table[(l + 1) * 5 + 4]) ++l;
  const S* master = flat_master + table[l * 5 + 0];
  const int X = (int)table[l * 5 + 1], Y = (int)table[l * 5 + 2];
  T* t2 = t2_base + table[l * 5 + 3];
  const int local = (int)((int64_t)blockIdx.x - table[l * 5 + 4]);
  const int tiles_y = (Y + 63) / 64;
  const int tiles_xy = ((X + 63) / 64) * tiles_y;
  const int tap = local / tiles_xy, txy = local % tiles_xy;
  const int kh = tap >> 2, kw = tap & 3;
  const int ph = (kh & 1) ? 0 : 1, ty = (kh == 3 || kh == 2) ? 1 : 0;
  const int pw = (kw & 1) ? 0 : 1, tx = (kw == 3 || kw == 2) ? 1 : 0;
  const int phase = ph * 2 + pw, t = ty * 2 + tx;
  const int x0 = (txy / tiles_y) * 64, y0 = (txy % tiles_y) * 64;
  if ((X & 7) == 0 && (Y & 7) == 0) {
    // vector form: 8 consecutive y per load (16 / 32 bytes), 8 consecutive x per store; a 64 x 64 tile = 512 chunks each way
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int item = threadIdx.x + 256 * it;
      const int xx = item >> 3, c = item & 7;
      const int x = x0 + xx, y = y0 + 8 * c;
      float f[8];
      if (x < X && y < Y) {
        load8<S>(master, ((int64_t)x * 16 + tap) * Y + y, f);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) tile[xx][8 * c + e] = f[e];
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int item = threadIdx.x + 256 * it;
      const int yy = item >> 3, c = item & 7;
      const int y = y0 + yy, x = x0 + 8 * c;
      if (x < X && y < Y) {
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = tile[8 * c + e][yy];
        store8<T>(t2, (((int64_t)phase * Y + y) * 4 + t) * X + x, f);
      }
    }
    return;
  }
  const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int x = x0 + ly + 4 * r, y = y0 + lx;
    tile[ly + 4 * r][lx] = (x < X && y < Y) ? ElemTraits<S>::load(master + ((int64_t)x * 16 + tap) * Y + y) : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int y = y0 + ly + 4 * r, x = x0 + lx;
    if (x < X && y < Y) ElemTraits<T>::store(t2 + (((int64_t)phase * Y + y) * 4 + t) * X + x, tile[lx][ly + 4 * r]);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T* src, float* dst, int B, int C, int64_t HW) {
  const int64_t n = (int64_t)B * C * HW;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int64_t hw = e % HW;
    const int64_t bc = e / HW;
    const int c = (int)(bc % C);
    const int64_t b = bc / C;
    dst[e] = ElemTraits<T>::load(src + (b * HW + hw) * C + c);
  }
}

// master [X][T][Y] f32 -> out [X][row_stride] = [X][T][Ypad] (zero padded channels) + zero tail up to row_stride
template <typename T>
__global__ __launch_bounds__(256) void pack_rows_kernel(const float* master, int X, int Tn, int Y, int Ypad,
                                                        int row_stride, T* out) {
  const int64_t n = (int64_t)X * row_stride;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int k = (int)(e % row_stride);
    const int64_t x = e / row_stride;
    const int t = k / Ypad, y = k - t * Ypad;
    ElemTraits<T>::store(out + e, (t < Tn && y < Y) ? master[(x * Tn + t) * Y + y] : 0.0f);
  }
}

// master [X][T][Y] f32 -> out [Y][row_stride] with out[y][t'*X + x] = master[x][t][y], t' = flip ? T-1-t : t
// (input-gradient operand of a stride-1 conv: transposed channels, spatially flipped taps); 32x33 LDS tile
// transpose per tap, rows zero padded to row_stride.
template <typename T>
__global__ __launch_bounds__(256) void pack_transpose_taps_kernel(const float* master, int X, int Tn, int Y, int flip,
                                                                  int row_stride, T* out) {
  __shared__ float tile[32][33];
  const int t = blockIdx.y;
  const int td = flip ? Tn - 1 - t : t;
  const int tiles_y = (Y + 31) / 32;
  const int x0 = (blockIdx.x / tiles_y) * 32, y0 = (blockIdx.x % tiles_y) * 32;
  const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int x = x0 + ly + 8 * r, y = y0 + lx;
    tile[ly + 8 * r][lx] = (x < X && y < Y) ? master[((int64_t)x * Tn + t) * Y + y] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int y = y0 + ly + 8 * r, x = x0 + lx;
    if (x < X && y < Y) ElemTraits<T>::store(out + (int64_t)y * row_stride + (int64_t)td * X + x, tile[lx][ly + 8 * r]);
  }
  // zero tail of the rows (only the blocks of tap 0 / x-tile 0 do it)
  if (t == 0 && x0 == 0) {
    const int tail0 = Tn * X;
    for (int r = ly; r < 32; r += 8) {
      const int y = y0 + r;
      if (y < Y)
        for (int k = tail0 + lx; k < row_stride; k += 32) ElemTraits<T>::store(out + (int64_t)y * row_stride + k, 0.0f);
    }
  }
}

// sum of the P partial rows of one channel in f64.  WIDE = false: one wave per channel (4 channels per workgroup);
// WIDE = true (P > 256: the 256^2 / 128^2 layers of the DoubleConv nets hold thousands of partial rows): the whole
// workgroup works on ONE channel, so a lane walks P / 256 instead of P / 64 strided rows (24 -> 8 us at P = 8192)
template <bool WIDE>
__device__ __forceinline__ bool bn_partial_sums(const float* partials, int64_t P, int C, int& c, double& s1, double& s2) {
  __shared__ double sh[2][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  c = WIDE ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave;
  s1 = 0.0;
  s2 = 0.0;
  if (!WIDE && c >= C) return false;
  const int step = WIDE ? 256 : 64;
  for (int64_t r = WIDE ? threadIdx.x : lane; r < P; r += step) {
    s1 += (double)partials[(r * 2 + 0) * C + c];
    s2 += (double)partials[(r * 2 + 1) * C + c];
  }
  s1 = wave_sum_d(s1);
  s2 = wave_sum_d(s2);
  if (WIDE) {
    if (lane == 0) {
      sh[0][wave] = s1;
      sh[1][wave] = s2;
    }
    __syncthreads();
    s1 = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
    s2 = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
    return threadIdx.x == 0;
  }
  return lane == 0;
}

template <bool WIDE>
__global__ __launch_bounds__(256) void bn_fwd_finalize_kernel(const float* partials, int64_t P, int C, double count,
                                                              const float* gamma, const float* beta, float eps,
                                                              float momentum, float* rmean, float* rvar,
                                                              int64_t* nbt, float* mean, float* istd, float* scale,
                                                              float* shift) {
  int c;
  double s1, s2;
  if (bn_partial_sums<WIDE>(partials, P, C, c, s1, s2)) {
    const double mu = s1 / count;
    double var = s2 / count - mu * mu;
    if (var < 0.0) var = 0.0;
    const double is = 1.0 / sqrt(var + (double)eps);
    const double g = gamma ? (double)gamma[c] : 1.0, b = beta ? (double)beta[c] : 0.0;
    mean[c] = (float)mu;
    istd[c] = (float)is;
    scale[c] = (float)(g * is);
    shift[c] = (float)(b - mu * g * is);
    if (rmean) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      rmean[c] = (float)((1.0 - momentum) * (double)rmean[c] + momentum * mu);
      rvar[c] = (float)((1.0 - momentum) * (double)rvar[c] + momentum * unbiased);
    }
    if (nbt && c == 0) nbt[0] += 1;
  }
}

__global__ __launch_bounds__(256) void bn_eval_affine_kernel(const float* gamma, const float* beta, const float* rm,
                                                             const float* rv, float eps, int C, float* scale,
                                                             float* shift) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float is = 1.0f / sqrtf(rv[c] + eps);
  const float g = gamma ? gamma[c] : 1.0f, b = beta ? beta[c] : 0.0f;
  scale[c] = g * is;
  shift[c] = b - rm[c] * g * is;
}

// y = z*scale+shift -> leaky / relu copies.  8 channels (one or two 16-byte chunks) per thread.
template <typename T>
__global__ __launch_bounds__(256) void bn_act_kernel(const T* z, int64_t pixels, int C, const float* scale,
                                                     const float* shift, float slope, T* out_leaky, T* out_relu,
                                                     uint2* q8 = nullptr, uint8_t* qsc = nullptr) {
  const int64_t n = pixels * C;
  if ((C & 7) == 0) {
    const int64_t groups = n >> 3;
    // channel group of this thread, advanced incrementally (no 64-bit modulo per iteration)
    const unsigned cpg = (unsigned)C >> 3;
    const unsigned step = (unsigned)(((int64_t)gridDim.x * 256) % cpg);
    unsigned cgp = (unsigned)(((int64_t)blockIdx.x * 256 + threadIdx.x) % cpg);
    // The grid stride is a multiple of the channel-group count whenever C is a power of two (every layer of the five
    // nets): the thread then stays on the SAME 8 channels and their coefficients are loaded once, not per 16-byte chunk
    // (the per-chunk loads were 2/3 of the kernel's L1 traffic)
    float sc[8], sh[8];
    auto load_coef = [&](int c) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        sc[k] = scale[c + k];
        sh[k] = shift[c + k];
      }
    };
    const bool fixed = step == 0;
    if (fixed) load_coef((int)(cgp << 3));
    for (int64_t gidx = (int64_t)blockIdx.x * 256 + threadIdx.x; gidx < groups; gidx += (int64_t)gridDim.x * 256) {
      const int64_t e = gidx << 3;
      if (!fixed) {
        load_coef((int)(cgp << 3));
        cgp += step;
        if (cgp >= cpg) cgp -= cpg;
      }
      float v[8], o[8];
      load8<T>(z, e, v);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = v[k] * sc[k] + sh[k];
      if (out_leaky) {
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = v[k] > 0.f ? v[k] : v[k] * slope;
        store8<T>(out_leaky, e, o);
      }
      if (out_relu) {
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = fmaxf(v[k], 0.f);
        store8<T>(out_relu, e, o);
        if constexpr (sizeof(T) == 2) {
          if (q8) {                                  // MX-fp8 copy of exactly what the bf16 tensor holds (C % 32 == 0)
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = bf16_bits_to_f32(f32_to_bf16_bits(o[k]));
            int byte;
            q8[gidx] = mx_quant8(o, byte);
            if ((threadIdx.x & 3) == 0) qsc[gidx >> 2] = (uint8_t)byte;
          }
        }
      }
    }
  } else {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
      const int c = (int)(e % C);
      const float y = ElemTraits<T>::load(z + e) * scale[c] + shift[c];
      if (out_leaky) ElemTraits<T>::store(out_leaky + e, y > 0.f ? y : y * slope);
      if (out_relu) ElemTraits<T>::store(out_relu + e, fmaxf(y, 0.f));
    }
  }
}

template <bool WIDE>
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* partials, int64_t P, int C, double count,
                                                              float* dgamma, float* dbeta, float* coef) {
  int c;
  double s1, s2;
  if (bn_partial_sums<WIDE>(partials, P, C, c, s1, s2)) {
    if (dbeta) dbeta[c] = (float)s1;
    if (dgamma) dgamma[c] = (float)s2;
    coef[c] = (float)(s1 / count);
    coef[C + c] = (float)(s2 / count);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(T* g, const T* z, int64_t pixels, int C,
                                                           const float* scale, const float* mean, const float* istd,
                                                           const float* coef, uint2* q8 = nullptr, uint8_t* qsc = nullptr) {
  const int64_t n = pixels * C;
  if ((C & 7) == 0) {
    const int64_t groups = n >> 3;
    const unsigned cpg = (unsigned)C >> 3;
    const unsigned step = (unsigned)(((int64_t)gridDim.x * 256) % cpg);
    unsigned cgp = (unsigned)(((int64_t)blockIdx.x * 256 + threadIdx.x) % cpg);
    // (coefficients of the thread's 8 channels loaded once when the grid stride keeps it on them: see bn_act_kernel)
    float cm[8], ci[8], cs[8], c0[8], c1[8];
    auto load_coef = [&](int c) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        cm[k] = mean[c + k];
        ci[k] = istd[c + k];
        cs[k] = scale[c + k];
        c0[k] = coef[c + k];
        c1[k] = coef[C + c + k];
      }
    };
    const bool fixed = step == 0;
    if (fixed) load_coef((int)(cgp << 3));
    for (int64_t gidx = (int64_t)blockIdx.x * 256 + threadIdx.x; gidx < groups; gidx += (int64_t)gridDim.x * 256) {
      const int64_t e = gidx << 3;
      if (!fixed) {
        load_coef((int)(cgp << 3));
        cgp += step;
        if (cgp >= cpg) cgp -= cpg;
      }
      float gv[8], zv[8];
      load8<T>(g, e, gv);
      load8<T>(z, e, zv);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float xh = (zv[k] - cm[k]) * ci[k];
        gv[k] = cs[k] * (gv[k] - c0[k] - xh * c1[k]);
      }
      store8<T>(g, e, gv);
      if constexpr (sizeof(T) == 2) {
        if (q8) {
#pragma unroll
          for (int k = 0; k < 8; ++k) gv[k] = bf16_bits_to_f32(f32_to_bf16_bits(gv[k]));
          int byte;
          q8[gidx] = mx_quant8(gv, byte);
          if ((threadIdx.x & 3) == 0) qsc[gidx >> 2] = (uint8_t)byte;
        }
      }
    }
  } else {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
      const int c = (int)(e % C);
      const float xh = (ElemTraits<T>::load(z + e) - mean[c]) * istd[c];
      ElemTraits<T>::store(g + e, scale[c] * (ElemTraits<T>::load(g + e) - coef[c] - xh * coef[C + c]));
    }
  }
}


// ---- small tensors (the innermost U-Net levels): finalize + apply in ONE launch -------------------------------------
// A workgroup owns 32 channels.  Phase 1: it sums the P partial rows of its channels (thread = channel x 8 row
// groups, 128-byte coalesced rows, f64) -- every row-slice workgroup of the same channels repeats this (the partials are
// L2 resident) instead of waiting for a separate finalize launch; phase 2: it applies the affine map to its rows
// (thread = 16-byte chunk x 64 rows per pass).  Replaces bn_*_finalize + bn_act / bn_bwd_apply (2 launches of 6-9 us
// each at these sizes, launch-latency bound) by one.
__device__ __forceinline__ void bn_block_sums(const float* partials, int64_t P, int C, int c0, double (*sh)[32][32],
                                              double& s1, double& s2) {
  // thread = 4 channels (one 16-byte load per partial row and statistic) x 32 row groups, 4 rows in flight per thread
  const int cq = threadIdx.x & 7, rg = threadIdx.x >> 3;
  double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
  const float* base = partials + c0 + cq * 4;
#pragma unroll 4
  for (int64_t r = rg; r < P; r += 32) {
    const f32x4_t x = *reinterpret_cast<const f32x4_t*>(base + (r * 2 + 0) * C);
    const f32x4_t y = *reinterpret_cast<const f32x4_t*>(base + (r * 2 + 1) * C);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      a[k] += (double)x[k];
      b[k] += (double)y[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    sh[0][rg][cq * 4 + k] = a[k];
    sh[1][rg][cq * 4 + k] = b[k];
  }
  __syncthreads();
  s1 = 0.0;
  s2 = 0.0;
  if (threadIdx.x < 32) {
#pragma unroll 8
    for (int g = 0; g < 32; ++g) {
      s1 += sh[0][g][threadIdx.x];
      s2 += sh[1][g][threadIdx.x];
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_fwd_fused_kernel(const float* partials, int64_t P, int C, double count,
                                                           const float* gamma, const float* beta, float eps,
                                                           float momentum, float* rmean, float* rvar, int64_t* nbt,
                                                           float* mean, float* istd, float* scale, float* shift,
                                                           const T* z, int64_t pixels, float slope, T* out_leaky,
                                                           T* out_relu) {
  __shared__ double sh[2][32][32];
  __shared__ float aff[2][32];
  const int c0 = blockIdx.x * 32;
  double s1, s2;
  bn_block_sums(partials, P, C, c0, sh, s1, s2);
  if (threadIdx.x < 32) {
    const int c = c0 + threadIdx.x;
    const double mu = s1 / count;
    double var = s2 / count - mu * mu;
    if (var < 0.0) var = 0.0;
    const double is = 1.0 / sqrt(var + (double)eps);
    const double g = gamma ? (double)gamma[c] : 1.0, b = beta ? (double)beta[c] : 0.0;
    const float sc = (float)(g * is), shf = (float)(b - mu * g * is);
    aff[0][threadIdx.x] = sc;
    aff[1][threadIdx.x] = shf;
    if (blockIdx.y == 0) {                  // one row-slice workgroup publishes the statistics
      mean[c] = (float)mu;
      istd[c] = (float)is;
      scale[c] = sc;
      shift[c] = shf;
      if (rmean) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        rmean[c] = (float)((1.0 - momentum) * (double)rmean[c] + momentum * mu);
        rvar[c] = (float)((1.0 - momentum) * (double)rvar[c] + momentum * unbiased);
      }
      if (nbt && c == 0) nbt[0] += 1;
    }
  }
  __syncthreads();
  const int chunk = threadIdx.x & 3;
  float sc[8], shf[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    sc[k] = aff[0][chunk * 8 + k];
    shf[k] = aff[1][chunk * 8 + k];
  }
  for (int64_t r = (int64_t)blockIdx.y * 64 + (threadIdx.x >> 2); r < pixels; r += (int64_t)gridDim.y * 64) {
    const int64_t e = r * C + c0 + chunk * 8;
    float v[8], o[8];
    load8<T>(z, e, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = v[k] * sc[k] + shf[k];
    if (out_leaky) {
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = v[k] > 0.f ? v[k] : v[k] * slope;
      store8<T>(out_leaky, e, o);
    }
    if (out_relu) {
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = fmaxf(v[k], 0.f);
      store8<T>(out_relu, e, o);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_fused_kernel(const float* partials, int64_t P, int C, double count,
                                                           float* dgamma, float* dbeta, T* g, const T* z,
                                                           int64_t pixels, const float* scale, const float* mean,
                                                           const float* istd) {
  __shared__ double sh[2][32][32];
  __shared__ float cf[5][32];            // scale, mean, istd, sum g / n, sum g xhat / n
  const int c0 = blockIdx.x * 32;
  double s1, s2;
  bn_block_sums(partials, P, C, c0, sh, s1, s2);
  if (threadIdx.x < 32) {
    const int c = c0 + threadIdx.x;
    if (blockIdx.y == 0) {
      if (dbeta) dbeta[c] = (float)s1;
      if (dgamma) dgamma[c] = (float)s2;
    }
    cf[0][threadIdx.x] = scale[c];
    cf[1][threadIdx.x] = mean[c];
    cf[2][threadIdx.x] = istd[c];
    cf[3][threadIdx.x] = (float)(s1 / count);
    cf[4][threadIdx.x] = (float)(s2 / count);
  }
  __syncthreads();
  const int chunk = threadIdx.x & 3;
  float k0[8], k1[8], k2[8], k3[8], k4[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    k0[k] = cf[0][chunk * 8 + k];
    k1[k] = cf[1][chunk * 8 + k];
    k2[k] = cf[2][chunk * 8 + k];
    k3[k] = cf[3][chunk * 8 + k];
    k4[k] = cf[4][chunk * 8 + k];
  }
  for (int64_t r = (int64_t)blockIdx.y * 64 + (threadIdx.x >> 2); r < pixels; r += (int64_t)gridDim.y * 64) {
    const int64_t e = r * C + c0 + chunk * 8;
    float gv[8], zv[8];
    load8<T>(g, e, gv);
    load8<T>(z, e, zv);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float xh = (zv[k] - k1[k]) * k2[k];
      gv[k] = k0[k] * (gv[k] - k3[k] - xh * k4[k]);
    }
    store8<T>(g, e, gv);
  }
}

}  // namespace

extern "C" int adn_pack_weights(const float* master, int32_t X, int32_t Y, int32_t y_pad, int32_t dtype,
                                void* s2_out, void* t2_out, void* stream) {
  ADN_CHECK_ARG(master && X > 0 && Y > 0 && y_pad >= Y, "adn_pack_weights: bad arguments");
  ADN_CHECK_ARG(dtype == ADN_F32 || dtype == ADN_BF16, "adn_pack_weights: bad dtype %d", dtype);
  ADN_CHECK_ARG(s2_out || t2_out, "adn_pack_weights: no output");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t n = (int64_t)X * 16 * y_pad;
  if (s2_out) {
    if (dtype == ADN_BF16)
      hipLaunchKernelGGL((pack_s2_kernel<uint16_t>), dim3(blocks_for(n)), dim3(256), 0, st, master, X, Y, y_pad,
                         reinterpret_cast<uint16_t*>(s2_out));
    else
      hipLaunchKernelGGL((pack_s2_kernel<float>), dim3(blocks_for(n)), dim3(256), 0, st, master, X, Y, y_pad,
                         reinterpret_cast<float*>(s2_out));
    ADN_CHECK_LAUNCH();
  }
  if (t2_out) {
    const dim3 grid((unsigned)(adn_cdiv(X, 32) * adn_cdiv(Y, 32)), 16);
    if (dtype == ADN_BF16)
      hipLaunchKernelGGL((pack_t2_kernel<uint16_t>), grid, dim3(256), 0, st, master, X, Y,
                         reinterpret_cast<uint16_t*>(t2_out));
    else
      hipLaunchKernelGGL((pack_t2_kernel<float>), grid, dim3(256), 0, st, master, X, Y,
                         reinterpret_cast<float*>(t2_out));
    ADN_CHECK_LAUNCH();
  }
  return ADN_OK;
}

extern "C" int adn_nchw_slice_to_nhwc(const float* src, void* dst, int32_t B, int32_t C_total, int32_t c_lo, int32_t C,
                                      int32_t c_pad, int32_t H, int32_t W, int32_t dtype, void* stream) {
  ADN_CHECK_ARG(src && dst && B > 0 && C > 0 && c_pad >= C && H > 0 && W > 0 && c_lo >= 0 && c_lo + C <= C_total,
                "adn_nchw_slice_to_nhwc: bad arguments");
  ADN_CHECK_ARG(dtype == ADN_F32 || dtype == ADN_BF16, "adn_nchw_slice_to_nhwc: bad dtype %d", dtype);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t n = (int64_t)B * H * W;
  const float* s0 = src + (int64_t)c_lo * H * W;
  if (dtype == ADN_BF16)
    hipLaunchKernelGGL((nchw_to_nhwc_kernel<uint16_t>), dim3(blocks_for(n)), dim3(256), 0, st, s0,
                       reinterpret_cast<uint16_t*>(dst), B, C, c_pad, (int64_t)H * W, (int64_t)C_total);
  else
    hipLaunchKernelGGL((nchw_to_nhwc_kernel<float>), dim3(blocks_for(n)), dim3(256), 0, st, s0,
                       reinterpret_cast<float*>(dst), B, C, c_pad, (int64_t)H * W, (int64_t)C_total);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_nchw_to_nhwc(const float* src, void* dst, int32_t B, int32_t C, int32_t c_pad, int32_t H,
                                int32_t W, int32_t dtype, void* stream) {
  return adn_nchw_slice_to_nhwc(src, dst, B, C, 0, C, c_pad, H, W, dtype, stream);
}

extern "C" int adn_nhwc_to_nchw(const void* src, float* dst, int32_t B, int32_t C, int32_t H, int32_t W,
                                int32_t dtype, void* stream) {
  ADN_CHECK_ARG(src && dst && B > 0 && C > 0 && H > 0 && W > 0, "adn_nhwc_to_nchw: bad arguments");
  ADN_CHECK_ARG(dtype == ADN_F32 || dtype == ADN_BF16, "adn_nhwc_to_nchw: bad dtype %d", dtype);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t n = (int64_t)B * C * H * W;
  if (dtype == ADN_BF16)
    hipLaunchKernelGGL((nhwc_to_nchw_kernel<uint16_t>), dim3(blocks_for(n)), dim3(256), 0, st,
                       reinterpret_cast<const uint16_t*>(src), dst, B, C, (int64_t)H * W);
  else
    hipLaunchKernelGGL((nhwc_to_nchw_kernel<float>), dim3(blocks_for(n)), dim3(256), 0, st,
                       reinterpret_cast<const float*>(src), dst, B, C, (int64_t)H * W);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_bn_fwd_finalize(const float* partials, int64_t P, int32_t C, int64_t count, const float* gamma,
                                   const float* beta, float eps, float momentum, float* running_mean,
                                   float* running_var, int64_t* num_batches_tracked, float* mean, float* istd,
                                   float* scale, float* shift, void* stream) {
  ADN_CHECK_ARG(partials && P > 0 && C > 0 && count > 0 && mean && istd && scale && shift,
                "adn_bn_fwd_finalize: bad arguments");
  ADN_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "adn_bn_fwd_finalize: running stats mismatch");
  if (P > 256)
    hipLaunchKernelGGL(bn_fwd_finalize_kernel<true>, dim3((unsigned)C), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), partials, P, C, (double)count, gamma, beta, eps, momentum,
                       running_mean, running_var, num_batches_tracked, mean, istd, scale, shift);
  else
    hipLaunchKernelGGL(bn_fwd_finalize_kernel<false>, dim3((unsigned)adn_cdiv(C, 4)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), partials, P, C, (double)count, gamma, beta, eps, momentum,
                       running_mean, running_var, num_batches_tracked, mean, istd, scale, shift);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, float eps, int32_t C, float* scale, float* shift,
                                  void* stream) {
  ADN_CHECK_ARG(running_mean && running_var && scale && shift && C > 0, "adn_bn_eval_affine: bad arguments");
  hipLaunchKernelGGL(bn_eval_affine_kernel, dim3((unsigned)adn_cdiv(C, 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), gamma, beta, running_mean, running_var, eps, C, scale,
                     shift);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_bn_act(const void* z, int64_t pixels, int32_t C, int32_t dtype, const float* scale,
                          const float* shift, float slope, void* out_leaky, void* out_relu, void* stream) {
  ADN_CHECK_ARG(z && pixels > 0 && C > 0 && scale && shift && (out_leaky || out_relu), "adn_bn_act: bad arguments");
  ADN_CHECK_ARG(dtype == ADN_F32 || dtype == ADN_BF16, "adn_bn_act: bad dtype %d", dtype);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t work = (C & 7) == 0 ? pixels * C / 8 : pixels * C;
  if (dtype == ADN_BF16)
    hipLaunchKernelGGL((bn_act_kernel<uint16_t>), dim3(bn_blocks_for(work)), dim3(256), 0, st,
                       reinterpret_cast<const uint16_t*>(z), pixels, C, scale, shift, slope,
                       reinterpret_cast<uint16_t*>(out_leaky), reinterpret_cast<uint16_t*>(out_relu));
  else
    hipLaunchKernelGGL((bn_act_kernel<float>), dim3(bn_blocks_for(work)), dim3(256), 0, st,
                       reinterpret_cast<const float*>(z), pixels, C, scale, shift, slope,
                       reinterpret_cast<float*>(out_leaky), reinterpret_cast<float*>(out_relu));
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

// BN apply + ReLU with an additional MX-fp8 copy (e4m3 + E8M0 per 32 channels) of the bf16 output, for the fp8 conv path.
extern "C" int adn_bn_act_mx8(const void* z, int64_t pixels, int32_t C, const float* scale, const float* shift,
                              void* out_relu, void* out8, void* out_scales, void* stream) {
  ADN_CHECK_ARG(z && pixels > 0 && C > 0 && C % 32 == 0 && scale && shift && out_relu && out8 && out_scales,
                "adn_bn_act_mx8: bad arguments (C=%d must be a multiple of 32)", C);
  hipLaunchKernelGGL((bn_act_kernel<uint16_t>), dim3(bn_blocks_for(pixels * C / 8)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const uint16_t*>(z), pixels, C, scale, shift, 0.f,
                     (uint16_t*)nullptr, reinterpret_cast<uint16_t*>(out_relu), reinterpret_cast<uint2*>(out8),
                     reinterpret_cast<uint8_t*>(out_scales));
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_bn_bwd_apply_mx8(void* g, const void* z, int64_t pixels, int32_t C, const float* scale, const float* mean,
                                    const float* istd, const float* coef, void* out8, void* out_scales, void* stream) {
  ADN_CHECK_ARG(g && z && pixels > 0 && C > 0 && C % 32 == 0 && scale && mean && istd && coef && out8 && out_scales,
                "adn_bn_bwd_apply_mx8: bad arguments (C=%d must be a multiple of 32)", C);
  hipLaunchKernelGGL((bn_bwd_apply_kernel<uint16_t>), dim3(bn_blocks_for(pixels * C / 8)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), reinterpret_cast<uint16_t*>(g),
                     reinterpret_cast<const uint16_t*>(z), pixels, C, scale, mean, istd, coef,
                     reinterpret_cast<uint2*>(out8), reinterpret_cast<uint8_t*>(out_scales));
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

// Pre-reduction of BatchNorm partial rows for the layers that hold thousands of them (the 256^2 / 512^2 levels of the
// DoubleConv nets: one row per 128-pixel GEMM tile = 16 384 rows at B = 32, 256^2): the finalize kernels walk the rows with one
// workgroup per channel (4-byte reads at a 2 C float stride, 17-20 us per launch there).  Here a workgroup owns 32 channels x
// one slice of rows and reads whole 128-byte row segments; `slices` rows [slices][2][C] go to `out_rows`, which the
// finalize kernel then sums.  f64 accumulation inside a slice.
__global__ __launch_bounds__(256) void bn_partials_reduce_kernel(const float* partials, int64_t P, int C, int slices,
                                                                 float* out_rows) {
  __shared__ double sh[32][8][8];
  const int cq = threadIdx.x & 7, rl = threadIdx.x >> 3;            // 8 lanes x 4 channels, 32 row lanes
  const int c0 = blockIdx.x * 32 + cq * 4;
  const int64_t R = (P + slices - 1) / slices;
  const int64_t r0 = (int64_t)blockIdx.y * R;
  const int64_t r1 = r0 + R < P ? r0 + R : P;
  double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int64_t r = r0 + rl; r < r1; r += 32) {
    const f32x4_t x = *reinterpret_cast<const f32x4_t*>(partials + (r * 2 + 0) * C + c0);
    const f32x4_t y = *reinterpret_cast<const f32x4_t*>(partials + (r * 2 + 1) * C + c0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      a[k] += (double)x[k];
      a[4 + k] += (double)y[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) sh[rl][cq][k] = a[k];
  __syncthreads();
  if (threadIdx.x < 64) {                                           // 8 lanes x 8 values
    const int q = threadIdx.x >> 3, k = threadIdx.x & 7;
    double t = 0.0;
    for (int r = 0; r < 32; ++r) t += sh[r][q][k];
    out_rows[((int64_t)blockIdx.y * 2 + (k >> 2)) * C + blockIdx.x * 32 + q * 4 + (k & 3)] = (float)t;
  }
}

extern "C" int adn_bn_partials_reduce(const float* partials, int64_t P, int32_t C, int32_t slices, float* out_rows,
                                      void* stream) {
  ADN_CHECK_ARG(partials && out_rows && P > 0 && C > 0 && C % 32 == 0 && slices > 0 && slices <= 1024,
                "adn_bn_partials_reduce: bad arguments (C=%d must be a multiple of 32)", C);
  hipLaunchKernelGGL(bn_partials_reduce_kernel, dim3(C / 32, slices), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     partials, P, C, slices, out_rows);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_bn_bwd_finalize(const float* partials, int64_t P, int32_t C, int64_t count, float* dgamma,
                                   float* dbeta, float* coef, void* stream) {
  ADN_CHECK_ARG(partials && P > 0 && C > 0 && count > 0 && coef, "adn_bn_bwd_finalize: bad arguments");
  if (P > 256)
    hipLaunchKernelGGL(bn_bwd_finalize_kernel<true>, dim3((unsigned)C), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), partials, P, C, (double)count, dgamma, dbeta, coef);
  else
    hipLaunchKernelGGL(bn_bwd_finalize_kernel<false>, dim3((unsigned)adn_cdiv(C, 4)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), partials, P, C, (double)count, dgamma, dbeta, coef);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_bn_bwd_apply(void* g, const void* z, int64_t pixels, int32_t C, int32_t dtype, const float* scale,
                                const float* mean, const float* istd, const float* coef, void* stream) {
  ADN_CHECK_ARG(g && z && pixels > 0 && C > 0 && scale && mean && istd && coef, "adn_bn_bwd_apply: bad arguments");
  ADN_CHECK_ARG(dtype == ADN_F32 || dtype == ADN_BF16, "adn_bn_bwd_apply: bad dtype %d", dtype);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t work = (C & 7) == 0 ? pixels * C / 8 : pixels * C;
  if (dtype == ADN_BF16)
    hipLaunchKernelGGL((bn_bwd_apply_kernel<uint16_t>), dim3(bn_blocks_for(work)), dim3(256), 0, st,
                       reinterpret_cast<uint16_t*>(g), reinterpret_cast<const uint16_t*>(z), pixels, C, scale, mean,
                       istd, coef);
  else
    hipLaunchKernelGGL((bn_bwd_apply_kernel<float>), dim3(bn_blocks_for(work)), dim3(256), 0, st,
                       reinterpret_cast<float*>(g), reinterpret_cast<const float*>(z), pixels, C, scale, mean, istd,
                       coef);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}


// row slices per channel block of the fused small-tensor kernels: enough workgroups to cover the chip's CUs a few waves deep
static inline unsigned bn_fused_slices(int64_t pixels, int32_t C) {
  int64_t sl = adn_cdiv(pixels, 256);
  const int64_t cap = adn_cdiv(512, C / 32);
  if (sl > cap) sl = cap;
  if (sl < 1) sl = 1;
  return (unsigned)sl;
}

extern "C" int adn_bn_fwd_fused(const float* partials, int64_t P, int32_t C, int64_t count, const float* gamma,
                                const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                int64_t* num_batches_tracked, float* mean, float* istd, float* scale, float* shift,
                                const void* z, int64_t pixels, int32_t dtype, float slope, void* out_leaky, void* out_relu,
                                void* stream) {
  ADN_CHECK_ARG(partials && P > 0 && C > 0 && count > 0 && mean && istd && scale && shift && z && pixels > 0 &&
                    (out_leaky || out_relu),
                "adn_bn_fwd_fused: bad arguments");
  ADN_CHECK_ARG(C % 32 == 0, "adn_bn_fwd_fused: C must be a multiple of 32 (got %d)", C);
  ADN_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "adn_bn_fwd_fused: running stats mismatch");
  ADN_CHECK_ARG(dtype == ADN_F32 || dtype == ADN_BF16, "adn_bn_fwd_fused: bad dtype %d", dtype);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)(C / 32), bn_fused_slices(pixels, C));
  if (dtype == ADN_BF16)
    hipLaunchKernelGGL((bn_fwd_fused_kernel<uint16_t>), grid, dim3(256), 0, st, partials, P, C, (double)count, gamma, beta,
                       eps, momentum, running_mean, running_var, num_batches_tracked, mean, istd, scale, shift,
                       reinterpret_cast<const uint16_t*>(z), pixels, slope, reinterpret_cast<uint16_t*>(out_leaky),
                       reinterpret_cast<uint16_t*>(out_relu));
  else
    hipLaunchKernelGGL((bn_fwd_fused_kernel<float>), grid, dim3(256), 0, st, partials, P, C, (double)count, gamma, beta,
                       eps, momentum, running_mean, running_var, num_batches_tracked, mean, istd, scale, shift,
                       reinterpret_cast<const float*>(z), pixels, slope, reinterpret_cast<float*>(out_leaky),
                       reinterpret_cast<float*>(out_relu));
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_bn_bwd_fused(const float* partials, int64_t P, int32_t C, int64_t count, float* dgamma, float* dbeta,
                                void* g, const void* z, int64_t pixels, int32_t dtype, const float* scale,
                                const float* mean, const float* istd, void* stream) {
  ADN_CHECK_ARG(partials && P > 0 && C > 0 && count > 0 && g && z && pixels > 0 && scale && mean && istd,
                "adn_bn_bwd_fused: bad arguments");
  ADN_CHECK_ARG(C % 32 == 0, "adn_bn_bwd_fused: C must be a multiple of 32 (got %d)", C);
  ADN_CHECK_ARG(dtype == ADN_F32 || dtype == ADN_BF16, "adn_bn_bwd_fused: bad dtype %d", dtype);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)(C / 32), bn_fused_slices(pixels, C));
  if (dtype == ADN_BF16)
    hipLaunchKernelGGL((bn_bwd_fused_kernel<uint16_t>), grid, dim3(256), 0, st, partials, P, C, (double)count, dgamma,
                       dbeta, reinterpret_cast<uint16_t*>(g), reinterpret_cast<const uint16_t*>(z), pixels, scale, mean,
                       istd);
  else
    hipLaunchKernelGGL((bn_bwd_fused_kernel<float>), grid, dim3(256), 0, st, partials, P, C, (double)count, dgamma, dbeta,
                       reinterpret_cast<float*>(g), reinterpret_cast<const float*>(z), pixels, scale, mean, istd);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_pack_t2_multi(const void* flat_master, int32_t master_dtype, const int64_t* table, int32_t layers,
                                 int64_t total_blocks, int32_t dtype, void* t2_base, void* stream) {
  ADN_CHECK_ARG(flat_master && table && layers > 0 && total_blocks > 0 && t2_base, "adn_pack_t2_multi: bad arguments");
  ADN_CHECK_ARG(dtype == ADN_F32 || dtype == ADN_BF16, "adn_pack_t2_multi: bad dtype %d", dtype);
  ADN_CHECK_ARG(master_dtype == ADN_F32 || (master_dtype == ADN_BF16 && dtype == ADN_BF16),
                "adn_pack_t2_multi: master dtype %d cannot feed a dtype-%d pack", master_dtype, dtype);
  ADN_CHECK_ARG(total_blocks < (1ll << 31), "adn_pack_t2_multi: too many blocks");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (master_dtype == ADN_BF16)
    hipLaunchKernelGGL((pack_t2_multi_kernel<uint16_t, uint16_t>), dim3((unsigned)total_blocks), dim3(256), 0, st,
                       reinterpret_cast<const uint16_t*>(flat_master), table, layers,
                       reinterpret_cast<uint16_t*>(t2_base));
  else if (dtype == ADN_BF16)
    hipLaunchKernelGGL((pack_t2_multi_kernel<float, uint16_t>), dim3((unsigned)total_blocks), dim3(256), 0, st,
                       reinterpret_cast<const float*>(flat_master), table, layers, reinterpret_cast<uint16_t*>(t2_base));
  else
    hipLaunchKernelGGL((pack_t2_multi_kernel<float, float>), dim3((unsigned)total_blocks), dim3(256), 0, st,
                       reinterpret_cast<const float*>(flat_master), table, layers, reinterpret_cast<float*>(t2_base));
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_pack_rows(const float* master, int32_t X, int32_t taps, int32_t Y, int32_t y_pad, int32_t row_stride,
                             int32_t dtype, void* out, void* stream) {
  ADN_CHECK_ARG(master && out && X > 0 && taps > 0 && Y > 0 && y_pad >= Y && row_stride >= taps * y_pad,
                "adn_pack_rows: bad arguments");
  ADN_CHECK_ARG(dtype == ADN_F32 || dtype == ADN_BF16, "adn_pack_rows: bad dtype %d", dtype);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t n = (int64_t)X * row_stride;
  if (dtype == ADN_BF16)
    hipLaunchKernelGGL((pack_rows_kernel<uint16_t>), dim3(blocks_for(n)), dim3(256), 0, st, master, X, taps, Y, y_pad,
                       row_stride, reinterpret_cast<uint16_t*>(out));
  else
    hipLaunchKernelGGL((pack_rows_kernel<float>), dim3(blocks_for(n)), dim3(256), 0, st, master, X, taps, Y, y_pad,
                       row_stride, reinterpret_cast<float*>(out));
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_pack_transpose_taps(const float* master, int32_t X, int32_t taps, int32_t Y, int32_t flip,
                                       int32_t row_stride, int32_t dtype, void* out, void* stream) {
  ADN_CHECK_ARG(master && out && X > 0 && taps > 0 && Y > 0 && row_stride >= taps * X,
                "adn_pack_transpose_taps: bad arguments");
  ADN_CHECK_ARG(dtype == ADN_F32 || dtype == ADN_BF16, "adn_pack_transpose_taps: bad dtype %d", dtype);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)(adn_cdiv(X, 32) * adn_cdiv(Y, 32)), taps);
  if (dtype == ADN_BF16)
    hipLaunchKernelGGL((pack_transpose_taps_kernel<uint16_t>), grid, dim3(256), 0, st, master, X, taps, Y, flip,
                       row_stride, reinterpret_cast<uint16_t*>(out));
  else
    hipLaunchKernelGGL((pack_transpose_taps_kernel<float>), grid, dim3(256), 0, st, master, X, taps, Y, flip,
                       row_stride, reinterpret_cast<float*>(out));
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}
