// Weight-gradient kernel of the k4/s2/p1 Conv2d / ConvTranspose2d pair (gfx950).
//
//   dW[r][tap][c] = sum_{m on the small grid} plain[m][r] * gath[b, 2i-1+ky, 2j-1+kx][c]
//
// The contraction runs over PIXELS while both operands are channel-contiguous (NHWC), so the MFMA
// fragments ("8 consecutive k per lane") are columns of the staged LDS tiles: the bf16 path reads
// them with ds_read_b64_tr_b16 (hardware transpose read, cdna_hip_programming.md T10), the exact
// f32 path with ds_read_b32.  LDS tiles are [pixel][128 channels] with the 32-byte granule index
// XOR-ed by f(row) = (row&3) | ((row>>3)&1)<<2, which makes both the transposed reads (8 rows x 32 B
// per half-wave) and the staging writes conflict free.  Output tile 128(r) x 128(tap,c columns);
// the pixel range is split over grid.z into f32 slabs that a second kernel sums (deterministic).
#include "adn_common.h"

namespace {

struct WParams {
  const void* plain0; const void* plain1; int R0, R1;
  const void* gath0;  const void* gath1;  int C0, C1;
  int B, Hs, Ws, Msmall;
  int steps;      // pixel steps in total
  int nsplit;
  int tiles_r, tiles_c;
  float* out;     // dW or slab base
  int64_t out_elems;
  int c_valid;    // gathered channels actually stored (compact [R][16][c_valid]); == C0+C1 normally
};

__device__ __forceinline__ int swz_f(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

template <typename T>
__global__ __launch_bounds__(256, 2) void wgrad_mfma_kernel(WParams p) {
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int BKP = sizeof(T) == 2 ? 64 : 32;       // pixels per step
  constexpr int ROWB = 128 * (int)sizeof(T);          // bytes per LDS row
  constexpr int CPRW = ROWB / 16;                     // 16-byte chunks per row
  constexpr int RPP = 256 / CPRW;                     // rows per loader pass
  constexpr int PASSES = BKP / RPP;                   // = 4
  constexpr int TILE = BKP * ROWB;                    // bytes per operand tile
  constexpr int LDC = 132;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ps = smem;                 // [2][TILE]
  char* Gs = smem + 2 * TILE;      // [2][TILE]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int tile_c = blockIdx.x % p.tiles_c;
  const int tile_r = blockIdx.x / p.tiles_c;
  const int split = blockIdx.z;
  const int Hs = p.Hs, Ws = p.Ws, Hl = 2 * Hs, Wl = 2 * Ws;
  const int C = p.C0 + p.C1;

  const int chunk = tid % CPRW;
  const int prow0 = tid / CPRW;

  // plain operand: channel range fixed per thread
  const int r_el = tile_r * 128 + chunk * EPC;
  const bool r_ok = r_el < p.R0 + p.R1;      // R may be a multiple of 64: upper half tile is zero
  const T* psrc;
  int Rsrc, roff;
  if (r_el < p.R0) {
    psrc = reinterpret_cast<const T*>(p.plain0);
    Rsrc = p.R0;
    roff = r_el;
  } else {
    psrc = reinterpret_cast<const T*>(p.plain1);
    Rsrc = p.R1;
    roff = r_el - p.R0;
  }
  // gathered operand: (tap, channel) fixed per thread
  const int gcol = tile_c * 128 + chunk * EPC;
  const int tap = gcol / C;
  const int cch = gcol - tap * C;
  const int ky = tap >> 2, kx = tap & 3;
  const T* gsrc;
  int Csrc, coff;
  if (cch < p.C0) {
    gsrc = reinterpret_cast<const T*>(p.gath0);
    Csrc = p.C0;
    coff = cch;
  } else {
    gsrc = reinterpret_cast<const T*>(p.gath1);
    Csrc = p.C1;
    coff = cch - p.C0;
  }

  const int s_begin = (int)(((int64_t)p.steps * split) / p.nsplit);
  const int s_end = (int)(((int64_t)p.steps * (split + 1)) / p.nsplit);

  u32x4_t rp[PASSES], rg[PASSES];
  auto load_step = [&](int s) {
#pragma unroll
    for (int k = 0; k < PASSES; ++k) {
      const int m = s * BKP + prow0 + RPP * k;
      u32x4_t vp = {0u, 0u, 0u, 0u}, vg = {0u, 0u, 0u, 0u};
      if (m < p.Msmall) {
        if (r_ok) vp = *reinterpret_cast<const u32x4_t*>(psrc + (int64_t)m * Rsrc + roff);
        const int b = m / (Hs * Ws);
        const int rem = m - b * (Hs * Ws);
        const int i = rem / Ws;
        const int j = rem - i * Ws;
        const int iy = 2 * i - 1 + ky, ix = 2 * j - 1 + kx;
        if ((unsigned)iy < (unsigned)Hl && (unsigned)ix < (unsigned)Wl) {
          const int64_t pix = ((int64_t)b * Hl + iy) * Wl + ix;
          vg = *reinterpret_cast<const u32x4_t*>(gsrc + pix * Csrc + coff);
        }
      }
      rp[k] = vp;
      rg[k] = vg;
    }
  };
  auto store_step = [&](int buf) {
#pragma unroll
    for (int k = 0; k < PASSES; ++k) {
      const int row = prow0 + RPP * k;
      const int phys = (((chunk >> 1) ^ swz_f(row)) << 1) | (chunk & 1);
      *reinterpret_cast<u32x4_t*>(Ps + buf * TILE + row * ROWB + phys * 16) = rp[k];
      *reinterpret_cast<u32x4_t*>(Gs + buf * TILE + row * ROWB + phys * 16) = rg[k];
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  if (s_begin < s_end) {
    load_step(s_begin);
    store_step(0);
  }
  __syncthreads();

  const int fi = lane & 15, fg = lane >> 4;
  for (int s = s_begin; s < s_end; ++s) {
    const int cur = (s - s_begin) & 1;
    const bool more = (s + 1) < s_end;
    if (more) load_step(s + 1);
    const char* Pb = Ps + cur * TILE;
    const char* Gb = Gs + cur * TILE;
    if constexpr (sizeof(T) == 2) {
      // lane (4q+pp of its 16-lane group) supplies row 8*fg+q (+4), 4 channels at 4*pp of the 16-block
      const int q = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
      for (int ks = 0; ks < BKP / 32; ++ks) {
        bf16x8_t af[4], bf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          s16x4_t lo, hi;
          {
            const int ch0 = wr * 64 + t * 16;  // channel block start (elements)
            const int row_lo = ks * 32 + 8 * fg + q, row_hi = row_lo + 4;
            const int g_lo = ((ch0 >> 4) ^ swz_f(row_lo)), g_hi = ((ch0 >> 4) ^ swz_f(row_hi));
            lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4_t*)(Pb + row_lo * ROWB + g_lo * 32 + pp * 8));
            hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4_t*)(Pb + row_hi * ROWB + g_hi * 32 + pp * 8));
            typedef __attribute__((ext_vector_type(8))) short s16x8_t;
            s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            af[t] = *reinterpret_cast<bf16x8_t*>(&v);
          }
          {
            const int ch0 = wc * 64 + t * 16;
            const int row_lo = ks * 32 + 8 * fg + q, row_hi = row_lo + 4;
            const int g_lo = ((ch0 >> 4) ^ swz_f(row_lo)), g_hi = ((ch0 >> 4) ^ swz_f(row_hi));
            lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4_t*)(Gb + row_lo * ROWB + g_lo * 32 + pp * 8));
            hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4_t*)(Gb + row_hi * ROWB + g_hi * 32 + pp * 8));
            typedef __attribute__((ext_vector_type(8))) short s16x8_t;
            s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            bf[t] = *reinterpret_cast<bf16x8_t*>(&v);
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < BKP / 4; ++ks) {
        const int row = ks * 4 + fg;
        float af[4], bf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int ca = wr * 64 + t * 16 + fi;   // channel (element) index in the 128-wide tile
          const int cb = wc * 64 + t * 16 + fi;
          af[t] = *reinterpret_cast<const float*>(Pb + row * ROWB + (((ca >> 3) ^ swz_f(row)) << 5) + (ca & 7) * 4);
          bf[t] = *reinterpret_cast<const float*>(Gb + row * ROWB + (((cb >> 3) ^ swz_f(row)) << 5) + (cb & 7) * 4);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    }
    if (more) store_step(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: tile -> LDS -> 16-byte row-contiguous f32 stores ----
  float* ct = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) ct[(wr * 64 + i * 16 + 4 * fg + r) * LDC + wc * 64 + j * 16 + fi] = acc[i][j][r];
  __syncthreads();
  float* out = p.out + (int64_t)split * p.out_elems;
  const int cq = tid & 31;    // float4 column group (32 per row)
  const int r0 = tid >> 5;    // 8 rows per pass
  const int R = p.R0 + p.R1;
  if (p.c_valid == C) {
    const int64_t ldo = (int64_t)16 * C;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int row = r0 + 8 * k;
      if (tile_r * 128 + row < R) {
        const f32x4_t v = *reinterpret_cast<const f32x4_t*>(ct + row * LDC + cq * 4);
        *reinterpret_cast<f32x4_t*>(out + (int64_t)(tile_r * 128 + row) * ldo + tile_c * 128 + cq * 4) = v;
      }
    }
  } else {
    // zero-padded gathered channels (edge layers): keep only c < c_valid, compact [R][16][c_valid]
    for (int k = 0; k < 16; ++k) {
      const int row = r0 + 8 * k;
      if (tile_r * 128 + row >= R) continue;
      for (int e = 0; e < 4; ++e) {
        const int gc = tile_c * 128 + cq * 4 + e;
        const int tp = gc / C, cc = gc - tp * C;
        if (cc < p.c_valid)
          out[((int64_t)(tile_r * 128 + row) * 16 + tp) * p.c_valid + cc] = ct[row * LDC + cq * 4 + e];
      }
    }
  }
}

// generic path: one thread per (output element, split)
template <typename T>
__global__ __launch_bounds__(256) void wgrad_direct_kernel(WParams p, int pix_per_split) {
  const int Hs = p.Hs, Ws = p.Ws, Hl = 2 * Hs, Wl = 2 * Ws;
  const int C = p.C0 + p.C1;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= p.out_elems) return;
  const int split = blockIdx.y;
  const int cv = p.c_valid;
  const int c = (int)(e % cv);
  const int tap = (int)((e / cv) % 16);
  const int r = (int)(e / ((int64_t)16 * cv));
  (void)C;
  const int ky = tap >> 2, kx = tap & 3;
  const T* ps = r < p.R0 ? reinterpret_cast<const T*>(p.plain0) + r : reinterpret_cast<const T*>(p.plain1) + (r - p.R0);
  const int Rs = r < p.R0 ? p.R0 : p.R1;
  const T* gs = c < p.C0 ? reinterpret_cast<const T*>(p.gath0) + c : reinterpret_cast<const T*>(p.gath1) + (c - p.C0);
  const int Cs = c < p.C0 ? p.C0 : p.C1;
  const int m0 = split * pix_per_split;
  int m1 = m0 + pix_per_split;
  if (m1 > p.Msmall) m1 = p.Msmall;
  float acc = 0.f;
  for (int m = m0; m < m1; ++m) {
    const int b = m / (Hs * Ws);
    const int rem = m - b * (Hs * Ws);
    const int i = rem / Ws, j = rem - i * Ws;
    const int iy = 2 * i - 1 + ky, ix = 2 * j - 1 + kx;
    if ((unsigned)iy >= (unsigned)Hl || (unsigned)ix >= (unsigned)Wl) continue;
    const int64_t pix = ((int64_t)b * Hl + iy) * Wl + ix;
    acc += ElemTraits<T>::load(ps + (int64_t)m * Rs) * ElemTraits<T>::load(gs + pix * Cs);
  }
  p.out[(int64_t)split * p.out_elems + e] = acc;
}

__global__ __launch_bounds__(256) void slab_sum_kernel(const float* slab, float* out, int64_t n, int nsplit) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    float v = 0.f;
    for (int s = 0; s < nsplit; ++s) v += slab[(int64_t)s * n + e];
    out[e] = v;
  }
}

struct WPlan {
  bool mfma;
  int nsplit, steps, tiles_r, tiles_c, pix_per_split;
  int64_t out_elems, slab_bytes;
};

void make_wplan(const AdnWgradDesc* d, WPlan* pl) {
  const int R = d->R0 + d->R1, C = d->C0 + d->C1;
  const int64_t msmall = (int64_t)d->B * d->Hs * d->Ws;
  const int cv = d->c_valid > 0 ? d->c_valid : C;
  pl->out_elems = (int64_t)R * 16 * cv;
  const int epc = d->dtype == ADN_BF16 ? 8 : 4;
  // sources are selected per 16-byte chunk, so a tile may straddle the two plain / gathered sources
  const bool aligned = (R % 64 == 0) && (d->R0 % epc == 0) && ((16 * C) % 128 == 0) && (C % epc == 0) &&
                       (d->C0 % epc == 0) && (C >= 128 ? (C % 128 == 0) : (128 % C == 0));
  pl->mfma = aligned;
  if (aligned) {
    const int bkp = d->dtype == ADN_BF16 ? 64 : 32;
    pl->steps = (int)adn_cdiv(msmall, bkp);
    pl->tiles_r = (int)adn_cdiv(R, 128);
    pl->tiles_c = 16 * C / 128;
    const int64_t tiles = (int64_t)pl->tiles_r * pl->tiles_c;
    int ns = (int)adn_cdiv(512, tiles);
    const int max_by_steps = pl->steps / 4 > 0 ? pl->steps / 4 : 1;
    if (ns > max_by_steps) ns = max_by_steps;
    if (ns > 256) ns = 256;
    if (ns < 1) ns = 1;
    pl->nsplit = ns;
    pl->pix_per_split = 0;
  } else {
    int ns = (int)adn_cdiv(msmall, 2048);
    if (ns > 1024) ns = 1024;
    if (ns < 1) ns = 1;
    pl->nsplit = ns;
    pl->pix_per_split = (int)adn_cdiv(msmall, ns);
    pl->steps = 0;
    pl->tiles_r = pl->tiles_c = 0;
  }
  pl->slab_bytes = pl->nsplit > 1 ? (int64_t)pl->nsplit * pl->out_elems * 4 : 0;
}

int wvalidate(const AdnWgradDesc* d) {
  ADN_CHECK_ARG(d != nullptr, "adn_wgrad: null descriptor");
  ADN_CHECK_ARG(d->dtype == ADN_F32 || d->dtype == ADN_BF16, "adn_wgrad: bad dtype %d", d->dtype);
  ADN_CHECK_ARG(d->B > 0 && d->Hs > 0 && d->Ws > 0, "adn_wgrad: bad shape");
  ADN_CHECK_ARG(d->R0 > 0 && d->R1 >= 0 && d->C0 > 0 && d->C1 >= 0, "adn_wgrad: bad channels");
  ADN_CHECK_ARG(d->plain0 && d->gath0 && d->dw, "adn_wgrad: null operand");
  ADN_CHECK_ARG((d->R1 == 0 || d->plain1) && (d->C1 == 0 || d->gath1), "adn_wgrad: null second source");
  ADN_CHECK_ARG((int64_t)d->B * d->Hs * d->Ws * 4 < (1ll << 31), "adn_wgrad: tensor too large");
  ADN_CHECK_ARG(d->c_valid >= 0 && d->c_valid <= d->C0 + d->C1, "adn_wgrad: bad c_valid %d", d->c_valid);
  return ADN_OK;
}

template <typename T>
int wrun(const AdnWgradDesc* d, const WPlan& pl, hipStream_t st) {
  WParams p;
  p.plain0 = d->plain0; p.plain1 = d->plain1; p.R0 = d->R0; p.R1 = d->R1;
  p.gath0 = d->gath0; p.gath1 = d->gath1; p.C0 = d->C0; p.C1 = d->C1;
  p.B = d->B; p.Hs = d->Hs; p.Ws = d->Ws; p.Msmall = d->B * d->Hs * d->Ws;
  p.steps = pl.steps; p.nsplit = pl.nsplit; p.tiles_r = pl.tiles_r; p.tiles_c = pl.tiles_c;
  p.out = pl.nsplit > 1 ? reinterpret_cast<float*>(d->workspace) : d->dw;
  p.out_elems = pl.out_elems;
  p.c_valid = d->c_valid > 0 ? d->c_valid : d->C0 + d->C1;
  if (pl.mfma) {
    constexpr int BKP = sizeof(T) == 2 ? 64 : 32;
    constexpr int stage = 4 * BKP * 128 * (int)sizeof(T);
    constexpr int epil = 128 * 132 * 4;
    constexpr int lds = stage > epil ? stage : epil;
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_mfma_kernel<T>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      attr_set = true;
    }
    hipLaunchKernelGGL((wgrad_mfma_kernel<T>), dim3(pl.tiles_r * pl.tiles_c, 1, pl.nsplit), dim3(256), lds, st, p);
  } else {
    hipLaunchKernelGGL((wgrad_direct_kernel<T>), dim3((unsigned)adn_cdiv(pl.out_elems, 256), pl.nsplit), dim3(256),
                       0, st, p, pl.pix_per_split);
  }
  ADN_CHECK_LAUNCH();
  if (pl.nsplit > 1) {
    int64_t blocks = adn_cdiv(pl.out_elems, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, st,
                       reinterpret_cast<const float*>(d->workspace), d->dw, pl.out_elems, pl.nsplit);
    ADN_CHECK_LAUNCH();
  }
  return ADN_OK;
}

}  // namespace

extern "C" int64_t adn_wgrad_workspace_bytes(const AdnWgradDesc* d) {
  if (wvalidate(d) != ADN_OK) return -1;
  WPlan pl;
  make_wplan(d, &pl);
  return pl.slab_bytes;
}

extern "C" int adn_wgrad(const AdnWgradDesc* d, void* stream) {
  int rc = wvalidate(d);
  if (rc != ADN_OK) return rc;
  WPlan pl;
  make_wplan(d, &pl);
  ADN_CHECK_ARG(pl.slab_bytes == 0 || (d->workspace && d->workspace_bytes >= pl.slab_bytes),
                "adn_wgrad: workspace too small (%lld < %lld)", (long long)d->workspace_bytes,
                (long long)pl.slab_bytes);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d->dtype == ADN_BF16) return wrun<uint16_t>(d, pl, st);
  return wrun<float>(d, pl, st);
}
