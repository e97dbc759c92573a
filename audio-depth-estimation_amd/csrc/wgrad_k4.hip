// Weight-gradient kernel of the k4/s2/p1 Conv2d / ConvTranspose2d pair (gfx950) -- the U-Net baseline's layers.
// (The stride-1 3x3 / 1x1 geometry of the DoubleConv nets lives in wgrad.hip: same structure with run-time tap
// geometry; keeping this k4 instantiation separate keeps its constant-folded addressing, measured 5-10 % faster.)
//
//   dW[r][tap][c] = sum_{m on the small grid} plain[m][r] * gath[b, 2i-1+ky, 2j-1+kx][c]
//
// The contraction runs over PIXELS while both operands are channel-contiguous (NHWC), so the MFMA
// fragments ("8 consecutive k per lane") are columns of the staged LDS tiles: the bf16 path reads
// them with ds_read_b64_tr_b16 (hardware transpose read, cdna_hip_programming.md T10), the exact
// f32 path with ds_read_b32.  LDS tiles are [pixel][128 channels] with the 32-byte granule index
// XOR-ed by f(row) = (row&3) | ((row>>3)&1)<<2, which makes both the transposed reads (8 rows x 32 B
// per half-wave) conflict free; the tiles are filled by LDS-DMA with the swizzle on the source address.  Output tile 128(r) x 128(tap,c columns);
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
  double* sq;     // optional: per-workgroup sum of dW^2 (only with nsplit == 1 and c_valid == C: this kernel writes the final dW)
};

// sum over the 256 threads of a workgroup, result valid in thread 0 (sh: 4 doubles)
__device__ __forceinline__ double wg4_block_sum_d(double v, double* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// 128 zero bytes: LDS-DMA source for rows beyond M / padded taps / the upper half of an R=64 tile
__device__ u32x4_t adn_wg4_zero_page[8];

__device__ __forceinline__ int swz_f(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

// (the body is shared by the one-problem kernel and the multi-problem launch of the small-image layers: `bid` is the
//  workgroup index inside its problem)
template <typename T, bool FAST>
__device__ __forceinline__ void wgrad_mfma_body(const WParams& p, const int bid) {
#if defined(__HIP_DEVICE_COMPILE__)   // buffer-resource builtins exist only in the device pass
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
  // XCD-contiguous order: all output tiles of one pixel split run on the same XCD and share the staged
  // operands through its L2 (blocks b and b+8 share an XCD)
  const int ntile = p.tiles_r * p.tiles_c;
  const int nblk = ntile * p.nsplit;
  const int bq = nblk >> 3, br = nblk & 7, bx = bid & 7;
  const int lid = (bx < br ? bx * (bq + 1) : br * (bq + 1) + (bx - br) * bq) + (bid >> 3);
  const int tile_c = lid % p.tiles_c;
  const int tile_r = (lid / p.tiles_c) % p.tiles_r;
  const int split = lid / ntile;
  const int Hs = p.Hs, Ws = p.Ws, Hl = 2 * Hs, Wl = 2 * Ws;
  const int C = p.C0 + p.C1;

  // LDS-DMA staging (global_load_lds_dwordx4): wave w writes 1 KiB = RPP/4 consecutive tile rows per pass,
  // lane l lands on row (l / CPRW) of that group, physical chunk (l % CPRW).  The granule swizzle is applied
  // on the SOURCE side: the lane fetches the logical chunk whose swizzled position is its physical chunk.
  // f(row) depends on the pass only through (row>>3)&1, which alternates with the pass for the f32 tile
  // (8 rows per pass), hence NV = 2 source variants there.
  const int pc = tid % CPRW;
  const int prow0 = tid / CPRW;
  constexpr int NV = sizeof(T) == 2 ? 1 : 2;
  const T* zero = reinterpret_cast<const T*>(adn_wg4_zero_page);
  const T* psrc[NV];
  int Rsrc[NV];
  bool r_ok[NV];
  const T* gsrc[NV];
  int Csrc[NV], ky[NV], kx[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int row = prow0 + RPP * v;
    const int lc = (((pc >> 1) ^ swz_f(row)) << 1) | (pc & 1);
    const int r_el = tile_r * 128 + lc * EPC;
    r_ok[v] = r_el < p.R0 + p.R1;      // R may be a multiple of 64: upper half tile is zero
    if (r_el < p.R0) {
      psrc[v] = reinterpret_cast<const T*>(p.plain0) + r_el;
      Rsrc[v] = p.R0;
    } else {
      psrc[v] = reinterpret_cast<const T*>(p.plain1) + (r_el - p.R0);
      Rsrc[v] = p.R1;
    }
    const int gcol = tile_c * 128 + lc * EPC;
    const int tap = gcol / C;
    const int cch = gcol - tap * C;
    ky[v] = tap >> 2;
    kx[v] = tap & 3;
    if (cch < p.C0) {
      gsrc[v] = reinterpret_cast<const T*>(p.gath0) + cch;
      Csrc[v] = p.C0;
    } else {
      gsrc[v] = reinterpret_cast<const T*>(p.gath1) + (cch - p.C0);
      Csrc[v] = p.C1;
    }
  }

  const int s_begin = (int)(((int64_t)p.steps * split) / p.nsplit);
  const int s_end = (int)(((int64_t)p.steps * (split + 1)) / p.nsplit);

  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  // ---- FAST addressing (power-of-two image, >= BKP pixels per image, tile-uniform sources) ----
  // address = descriptor base (SGPR) + voffset (per lane, constant over the pixel loop) + soffset (scalar per
  // step).  For pixel m = s*BKP + r the gathered pixel (2i, 2j) has linear index 4m - 2j, which splits into a
  // scalar part of s and a lane constant of r; border taps are recognised by scalar compares combined with
  // lane-constant masks, and get voffset 0x80000000 (hardware range check -> the DMA writes zeros).
  constexpr unsigned OOB = 0x80000000u;
  constexpr int ESZ = (int)sizeof(T);
  unsigned pvoff[PASSES], gvoff[PASSES];
  bool m_y0[PASSES], m_y1[PASSES], m_x0[PASSES], m_x1[PASSES], m_c[PASSES];
  __amdgpu_buffer_rsrc_t rsp, rsg;
  int Rs_u = 0, Cs_u = 0, lgWs = 0;
  if constexpr (FAST) {
    lgWs = 31 - __builtin_clz((unsigned)Ws);
    const bool psecond = tile_r * 128 >= p.R0;
    Rs_u = psecond ? p.R1 : p.R0;
    const char* pb = reinterpret_cast<const char*>(psecond ? p.plain1 : p.plain0);
    const bool gsecond = (C >= 128) && ((tile_c * 128) % C) >= p.C0;
    Cs_u = gsecond ? p.C1 : p.C0;
    const char* gb = reinterpret_cast<const char*>(gsecond ? p.gath1 : p.gath0) - (int64_t)(Wl + 1) * Cs_u * ESZ;
    rsp = __builtin_amdgcn_make_buffer_rsrc((void*)pb, 0, 0x7ffffff0, 0x00020000);
    rsg = __builtin_amdgcn_make_buffer_rsrc((void*)gb, 0, 0x7ffffff0, 0x00020000);
    const bool rows_in_line = Ws >= BKP;          // a step stays inside one image row
    const int q = rows_in_line ? 1 : BKP / Ws;    // image rows per step otherwise
#pragma unroll
    for (int k = 0; k < PASSES; ++k) {
      const int v = k % NV;
      const int r = prow0 + RPP * k;
      const int row = prow0 + RPP * v;
      const int lc = (((pc >> 1) ^ swz_f(row)) << 1) | (pc & 1);
      const int r_el = tile_r * 128 + lc * EPC;
      const int roff = r_el - (psecond ? p.R0 : 0);
      pvoff[k] = r_ok[v] ? (unsigned)((r * Rs_u + roff) * ESZ) : OOB;
      const int gcol = tile_c * 128 + lc * EPC;
      const int tap = gcol / C;
      const int cch = gcol - tap * C - (gsecond ? p.C0 : 0);
      const int jx = r & (Ws - 1);
      const int L = rows_in_line ? 2 * r : 4 * r - 2 * jx;
      gvoff[k] = (unsigned)(((L + ky[v] * Wl + kx[v]) * Cs_u + cch) * ESZ);
      const int a = r >> lgWs;                    // image row inside the step (0 when rows_in_line)
      m_y0[k] = (ky[v] == 0) && (rows_in_line || a == 0);
      m_y1[k] = (ky[v] == 3) && (rows_in_line || a == q - 1);
      m_x0[k] = rows_in_line && (kx[v] == 0) && (r == 0);
      m_x1[k] = rows_in_line && (kx[v] == 3) && (r == BKP - 1);
      m_c[k] = !rows_in_line && (((kx[v] == 0) && jx == 0) || ((kx[v] == 3) && jx == Ws - 1));
    }
  }

  auto issue_step = [&](int s, int buf) {
    char* pdst = Ps + buf * TILE + wave * 1024;
    char* gdst = Gs + buf * TILE + wave * 1024;
    if constexpr (FAST) {
      const int m0 = s * BKP;
      const bool rows_in_line = Ws >= BKP;
      const int q = rows_in_line ? 1 : BKP / Ws;
      const int sj = rows_in_line ? (m0 & (Ws - 1)) : 0;
      const int si = (m0 >> lgWs) & (Hs - 1);                 // first image row of the step
      const bool top = si == 0, bot = si == Hs - q;
      const bool left = rows_in_line && sj == 0, right = rows_in_line && sj == Ws - BKP;
      const int psoff = m0 * Rs_u * ESZ;
      const int gsoff = (4 * m0 - 2 * sj) * Cs_u * ESZ;
#pragma unroll
      for (int k = 0; k < PASSES; ++k) {
        const bool inval = m_c[k] || (top && m_y0[k]) || (bot && m_y1[k]) || (left && m_x0[k]) || (right && m_x1[k]);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsp, (lptr_t)(pdst + k * (RPP * ROWB)), 16, pvoff[k], psoff, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsg, (lptr_t)(gdst + k * (RPP * ROWB)), 16,
                                                 inval ? OOB : gvoff[k], gsoff, 0, 0);
      }
    } else {
#pragma unroll
      for (int k = 0; k < PASSES; ++k) {
        const int v = k % NV;
        const int m = s * BKP + prow0 + RPP * k;
        const T* pp = zero;
        const T* gg = zero;
        if (m < p.Msmall) {
          if (r_ok[v]) pp = psrc[v] + (int64_t)m * Rsrc[v];
          const int b = m / (Hs * Ws);
          const int rem = m - b * (Hs * Ws);
          const int i = rem / Ws;
          const int j = rem - i * Ws;
          const int iy = 2 * i - 1 + ky[v], ix = 2 * j - 1 + kx[v];
          if ((unsigned)iy < (unsigned)Hl && (unsigned)ix < (unsigned)Wl)
            gg = gsrc[v] + (((int64_t)b * Hl + iy) * Wl + ix) * Csrc[v];
        }
        __builtin_amdgcn_global_load_lds((gptr_t)pp, (lptr_t)(pdst + k * (RPP * ROWB)), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)gg, (lptr_t)(gdst + k * (RPP * ROWB)), 16, 0, 0);
      }
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  if (s_begin < s_end) issue_step(s_begin, 0);
  __syncthreads();   // drains the LDS-DMA (vmcnt(0)) in front of the barrier

  const int fi = lane & 15, fg = lane >> 4;
  for (int s = s_begin; s < s_end; ++s) {
    const int cur = (s - s_begin) & 1;
    if (s + 1 < s_end) issue_step(s + 1, cur ^ 1);
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
    double sq = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int row = r0 + 8 * k;
      if (tile_r * 128 + row < R) {
        const f32x4_t v = *reinterpret_cast<const f32x4_t*>(ct + row * LDC + cq * 4);
        *reinterpret_cast<f32x4_t*>(out + (int64_t)(tile_r * 128 + row) * ldo + tile_c * 128 + cq * 4) = v;
        sq += (double)(v[0] * v[0] + v[1] * v[1]) + (double)(v[2] * v[2] + v[3] * v[3]);   // as sqsum_partial_kernel
      }
    }
    if (p.sq) {     // uniform: the final dW leaves this kernel, its share of the gradient norm rides along
      __syncthreads();                              // everyone is done with the staged tile: reuse its first bytes
      sq = wg4_block_sum_d(sq, reinterpret_cast<double*>(smem));
      if (tid == 0) p.sq[lid] = sq;
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
#endif
}

template <typename T, bool FAST>
__global__ __launch_bounds__(256, 2) void wgrad_mfma_kernel(WParams p) {
  wgrad_mfma_body<T, FAST>(p, (int)blockIdx.x);
}

// Several independent weight-gradient problems in ONE launch (the small-image levels of the U-Net: six launches of
// 256-512 short workgroups each, whose time is launch ramp and dW write latency, not arithmetic).
constexpr int kWgradBatchMax = 8;
struct WBatch {
  WParams p[kWgradBatchMax];
  int first[kWgradBatchMax + 1];      // first workgroup of problem k; first[n] = grid size
  int n;
};
template <typename T, bool FAST>
__global__ __launch_bounds__(256, 2) void wgrad_mfma_batch_kernel(WBatch b) {
  int k = 0;
#pragma unroll
  for (int j = 1; j < kWgradBatchMax; ++j)
    if (j < b.n && (int)blockIdx.x >= b.first[j]) k = j;
  wgrad_mfma_body<T, FAST>(b.p[k], (int)blockIdx.x - b.first[k]);
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

__global__ __launch_bounds__(256) void slab_sum_kernel(const float* slab, float* out, int64_t n, int nsplit, double* sq) {
  // n is a multiple of 4 (R*16*c with R % 64 == 0); 4 independent accumulator chains hide the load latency
  const int64_t n4 = n >> 2;
  const f32x4_t* s4 = reinterpret_cast<const f32x4_t*>(slab);
  double sqs = 0.0;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (int64_t)gridDim.x * 256) {
    f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    int s = 0;
    for (; s + 4 <= nsplit; s += 4) {
      a0 += s4[(int64_t)(s + 0) * n4 + e];
      a1 += s4[(int64_t)(s + 1) * n4 + e];
      a2 += s4[(int64_t)(s + 2) * n4 + e];
      a3 += s4[(int64_t)(s + 3) * n4 + e];
    }
    for (; s < nsplit; ++s) a0 += s4[(int64_t)s * n4 + e];
    const f32x4_t v = (a0 + a1) + (a2 + a3);
    reinterpret_cast<f32x4_t*>(out)[e] = v;
    sqs += (double)(v[0] * v[0] + v[1] * v[1]) + (double)(v[2] * v[2] + v[3] * v[3]);      // as sqsum_partial_kernel
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t e = (n4 << 2) + threadIdx.x;
    float v = 0.f;
    for (int s = 0; s < nsplit; ++s) v += slab[(int64_t)s * n + e];
    out[e] = v;
    sqs += (double)v * v;
  }
  if (sq) {         // uniform: this launch writes the final dW, its share of the gradient norm rides along
    __shared__ double sh[4];
    sqs = wg4_block_sum_d(sqs, sh);
    if (threadIdx.x == 0) sq[blockIdx.x] = sqs;
  }
}

// workgroups of the slab sum: enough to stream at the HBM rate; fewer when every workgroup leaves a norm partial behind
inline int64_t slab_sum_blocks(int64_t out_elems, bool sq) {
  int64_t blocks = adn_cdiv(adn_cdiv(out_elems, 4), 256);
  const int64_t cap = sq ? 1024 : 4096;
  return blocks > cap ? cap : blocks;
}

struct WPlan {
  bool mfma;
  bool fast;
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
  pl->fast = false;
  if (aligned) {
    const int bkp = d->dtype == ADN_BF16 ? 64 : 32;
    auto pow2 = [](int x) { return x > 0 && (x & (x - 1)) == 0; };
    pl->fast = pow2(d->Hs) && pow2(d->Ws) && d->Hs * d->Ws >= bkp && (d->R1 == 0 || d->R0 % 128 == 0) &&
               (d->C1 == 0 || (d->C0 % 128 == 0 && C % 128 == 0)) &&
               msmall * 4 * (d->C0 > d->C1 ? d->C0 : d->C1) * 4 < (1ll << 31) &&
               msmall * (d->R0 > d->R1 ? d->R0 : d->R1) * 4 < (1ll << 31);     // 32-bit scalar byte offsets
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

void fill_wparams(const AdnWgradDesc* d, const WPlan& pl, WParams& p) {
  p.plain0 = d->plain0; p.plain1 = d->plain1; p.R0 = d->R0; p.R1 = d->R1;
  p.gath0 = d->gath0; p.gath1 = d->gath1; p.C0 = d->C0; p.C1 = d->C1;
  p.B = d->B; p.Hs = d->Hs; p.Ws = d->Ws; p.Msmall = d->B * d->Hs * d->Ws;
  p.steps = pl.steps; p.nsplit = pl.nsplit; p.tiles_r = pl.tiles_r; p.tiles_c = pl.tiles_c;
  p.out = pl.nsplit > 1 ? reinterpret_cast<float*>(d->workspace) : d->dw;
  p.out_elems = pl.out_elems;
  p.c_valid = d->c_valid > 0 ? d->c_valid : d->C0 + d->C1;
  p.sq = (pl.mfma && pl.nsplit == 1 && p.c_valid == d->C0 + d->C1) ? d->sq_partials : nullptr;
}

template <typename T>
int wrun(const AdnWgradDesc* d, const WPlan& pl, hipStream_t st) {
  WParams p;
  fill_wparams(d, pl, p);
  if (pl.mfma) {
    constexpr int BKP = sizeof(T) == 2 ? 64 : 32;
    constexpr int stage = 4 * BKP * 128 * (int)sizeof(T);
    constexpr int epil = 128 * 132 * 4;
    constexpr int lds = stage > epil ? stage : epil;
    ADN_SET_LDS_ONCE(lds, &wgrad_mfma_kernel<T, true>);
    ADN_SET_LDS_ONCE(lds, &wgrad_mfma_kernel<T, false>);
    if (pl.fast)
      hipLaunchKernelGGL((wgrad_mfma_kernel<T, true>), dim3(pl.tiles_r * pl.tiles_c * pl.nsplit), dim3(256), lds, st, p);
    else
      hipLaunchKernelGGL((wgrad_mfma_kernel<T, false>), dim3(pl.tiles_r * pl.tiles_c * pl.nsplit), dim3(256), lds, st, p);
  } else {
    hipLaunchKernelGGL((wgrad_direct_kernel<T>), dim3((unsigned)adn_cdiv(pl.out_elems, 256), pl.nsplit), dim3(256),
                       0, st, p, pl.pix_per_split);
  }
  ADN_CHECK_LAUNCH();
  if (pl.nsplit > 1) {
    const int64_t blocks = slab_sum_blocks(pl.out_elems, d->sq_partials != nullptr);
    hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, st,
                       reinterpret_cast<const float*>(d->workspace), d->dw, pl.out_elems, pl.nsplit, d->sq_partials);
    ADN_CHECK_LAUNCH();
  }
  return ADN_OK;
}

}  // namespace

bool adn_wgrad_k4p_plan(const AdnWgradDesc* d, int* nsplit, int64_t* out_elems);   // wgrad_s1p.hip (patch-staged kernels)
int adn_wgrad_k4p_launch(const AdnWgradDesc* d, int nsplit, int64_t out_elems, void* stream);

int adn_wgrad_k4p_batch_launch(const AdnWgradDesc* descs, int n, const int* nsplit, float* const* out, void* stream);

// Patch-staged layers as ONE launch with 1/n of the pixel splits each (n <= 4 layers): per problem the split count of a
// lone launch divided by n.  Slab regions are laid out back to back in descs[0].workspace.
namespace {
bool patch_batch_plan(const AdnWgradDesc* descs, int n, int* ns, int64_t* oe, int64_t* off, int64_t* total) {
  int64_t at = 0;
  for (int k = 0; k < n; ++k) {
    int s1;
    if (wvalidate(descs + k) != ADN_OK || !adn_wgrad_k4p_plan(descs + k, &s1, &oe[k])) return false;
    ns[k] = s1 / n > 1 ? s1 / n : 1;
    off[k] = at;
    if (ns[k] > 1) at += (int64_t)ns[k] * oe[k] * 4;
  }
  *total = at;
  return true;
}
}  // namespace

int64_t adn_wgrad_k4_patch_batch_workspace_bytes(const AdnWgradDesc* descs, int32_t n) {
  int ns[4];
  int64_t oe[4], off[4], total;
  if (!descs || n < 1 || n > 4 || !patch_batch_plan(descs, n, ns, oe, off, &total)) return -1;
  return total;
}

int adn_wgrad_k4_patch_batch(const AdnWgradDesc* descs, int32_t n, void* stream) {
  ADN_CHECK_ARG(descs && n >= 1 && n <= 4, "adn_wgrad_patch_batch: 1 .. 4 problems (got %d)", n);
  int ns[4];
  int64_t oe[4], off[4], total;
  ADN_CHECK_ARG(patch_batch_plan(descs, n, ns, oe, off, &total),
                "adn_wgrad_patch_batch: a problem is not a patch-staged layer (adn_wgrad_patch_batch_workspace_bytes < 0)");
  ADN_CHECK_ARG(total == 0 || (descs[0].workspace && descs[0].workspace_bytes >= total),
                "adn_wgrad_patch_batch: workspace too small (%lld < %lld)", (long long)descs[0].workspace_bytes, (long long)total);
  float* out[4];
  for (int k = 0; k < n; ++k)
    out[k] = ns[k] > 1 ? reinterpret_cast<float*>(reinterpret_cast<char*>(descs[0].workspace) + off[k]) : descs[k].dw;
  int rc = adn_wgrad_k4p_batch_launch(descs, n, ns, out, stream);
  if (rc != ADN_OK) return rc;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  for (int k = 0; k < n; ++k) {
    if (ns[k] <= 1) continue;
    const int64_t blocks = slab_sum_blocks(oe[k], descs[k].sq_partials != nullptr);
    hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, st, out[k], descs[k].dw, oe[k], ns[k],
                       descs[k].sq_partials);
    ADN_CHECK_LAUNCH();
  }
  return ADN_OK;
}

int64_t adn_wgrad_k4_workspace_bytes(const AdnWgradDesc* d) {
  if (wvalidate(d) != ADN_OK) return -1;
  {
    int ns;
    int64_t oe;
    if (adn_wgrad_k4p_plan(d, &ns, &oe)) return ns > 1 ? (int64_t)ns * oe * 4 : 0;
  }
  WPlan pl;
  make_wplan(d, &pl);
  return pl.slab_bytes;
}

// doubles adn_wgrad_k4 leaves in d->sq_partials: one per workgroup of the launch that writes the final dW
int32_t adn_wgrad_k4_sq_count(const AdnWgradDesc* d) {
  if (wvalidate(d) != ADN_OK) return 0;
  {
    int ns;
    int64_t oe;
    if (adn_wgrad_k4p_plan(d, &ns, &oe)) return ns > 1 ? (int32_t)slab_sum_blocks(oe, true) : 0;
  }
  WPlan pl;
  make_wplan(d, &pl);
  if (pl.nsplit > 1) return (int32_t)slab_sum_blocks(pl.out_elems, true);
  const int cv = d->c_valid > 0 ? d->c_valid : d->C0 + d->C1;
  if (pl.mfma && cv == d->C0 + d->C1) return pl.tiles_r * pl.tiles_c;
  return 0;
}

// 0: `d` cannot ride in adn_wgrad_k4_batch; 1 / 2: it can, in the general / power-of-two-image ("fast") form of the
// tap-staged MFMA kernel (a launch holds problems of ONE class).  Inside a batch every problem runs UNSPLIT -- the other
// problems of the launch provide the parallelism a lone launch gets from splitting the pixels -- so it writes the final dW
// (no slab sum) and tiles_r * tiles_c norm partials (adn_wgrad_k4_batch_sq_count).
int32_t adn_wgrad_k4_batchable(const AdnWgradDesc* d) {
  if (wvalidate(d) != ADN_OK || d->dtype != ADN_BF16) return 0;
  if (d->c_valid > 0 && d->c_valid != d->C0 + d->C1) return 0;
  int ns;
  int64_t oe;
  if (adn_wgrad_k4p_plan(d, &ns, &oe)) return 0;
  WPlan pl;
  make_wplan(d, &pl);
  if (!pl.mfma || pl.nsplit > 2) return 0;             // (a layer that wants many pixel splits is not a small one)
  return pl.fast ? 2 : 1;
}

int32_t adn_wgrad_k4_batch_sq_count(const AdnWgradDesc* d) {
  if (!adn_wgrad_k4_batchable(d)) return 0;
  WPlan pl;
  make_wplan(d, &pl);
  return pl.tiles_r * pl.tiles_c;
}

int adn_wgrad_k4_batch(const AdnWgradDesc* descs, int32_t n, void* stream) {
  ADN_CHECK_ARG(descs && n >= 1 && n <= kWgradBatchMax, "adn_wgrad_batch: 1 .. %d problems (got %d)", kWgradBatchMax, n);
  WBatch b;
  b.n = n;
  b.first[0] = 0;
  const int32_t cls = adn_wgrad_k4_batchable(descs);
  for (int k = 0; k < n; ++k) {
    const int32_t c = adn_wgrad_k4_batchable(descs + k);
    ADN_CHECK_ARG(c != 0, "adn_wgrad_batch: problem %d is not batchable (adn_wgrad_batchable)", k);
    ADN_CHECK_ARG(c == cls, "adn_wgrad_batch: problem %d is of class %d, problem 0 of class %d (one class per launch)", k, c, cls);
    WPlan pl;
    make_wplan(descs + k, &pl);
    pl.nsplit = 1;
    fill_wparams(descs + k, pl, b.p[k]);
    b.first[k + 1] = b.first[k] + pl.tiles_r * pl.tiles_c;
  }
  for (int k = n; k < kWgradBatchMax; ++k) {
    b.p[k] = b.p[0];
    b.first[k + 1] = b.first[n];
  }
  constexpr int lds = 128 * 132 * 4 > 4 * 64 * 128 * 2 ? 128 * 132 * 4 : 4 * 64 * 128 * 2;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (cls == 2) {
    ADN_SET_LDS_ONCE(lds, &wgrad_mfma_batch_kernel<uint16_t, true>);
    hipLaunchKernelGGL((wgrad_mfma_batch_kernel<uint16_t, true>), dim3((unsigned)b.first[n]), dim3(256), lds, st, b);
  } else {
    ADN_SET_LDS_ONCE(lds, &wgrad_mfma_batch_kernel<uint16_t, false>);
    hipLaunchKernelGGL((wgrad_mfma_batch_kernel<uint16_t, false>), dim3((unsigned)b.first[n]), dim3(256), lds, st, b);
  }
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

int adn_wgrad_k4(const AdnWgradDesc* d, void* stream) {
  int rc = wvalidate(d);
  if (rc != ADN_OK) return rc;
  {
    int ns;
    int64_t oe;
    if (adn_wgrad_k4p_plan(d, &ns, &oe)) {
      const int64_t need = ns > 1 ? (int64_t)ns * oe * 4 : 0;
      ADN_CHECK_ARG(need == 0 || (d->workspace && d->workspace_bytes >= need), "adn_wgrad: workspace too small (%lld < %lld)",
                    (long long)d->workspace_bytes, (long long)need);
      rc = adn_wgrad_k4p_launch(d, ns, oe, stream);
      if (rc != ADN_OK) return rc;
      if (ns > 1) {
        const int64_t blocks = slab_sum_blocks(oe, d->sq_partials != nullptr);
        hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                           reinterpret_cast<const float*>(d->workspace), d->dw, oe, ns, d->sq_partials);
        ADN_CHECK_LAUNCH();
      }
      return ADN_OK;
    }
  }
  WPlan pl;
  make_wplan(d, &pl);
  ADN_CHECK_ARG(pl.slab_bytes == 0 || (d->workspace && d->workspace_bytes >= pl.slab_bytes),
                "adn_wgrad: workspace too small (%lld < %lld)", (long long)d->workspace_bytes,
                (long long)pl.slab_bytes);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d->dtype == ADN_BF16) return wrun<uint16_t>(d, pl, st);
  return wrun<float>(d, pl, st);
}
