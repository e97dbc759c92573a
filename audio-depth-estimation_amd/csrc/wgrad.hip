// Weight-gradient kernel of the k4/s2/p1 Conv2d / ConvTranspose2d pair (gfx950).
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
  int c_valid;    // gathered channels actually stored (compact [R][taps][c_valid]); == C0+C1 normally
  int geom;       // 0: k4 s2 gather (16 taps, gathered tensor on the 2x grid); ADN_GEMM_S1: ks x ks, same grid
  int ks;
};

// 128 zero bytes: LDS-DMA source for rows beyond M / padded taps / the upper half of an R=64 tile
__device__ u32x4_t adn_wg_zero_page[8];

__device__ __forceinline__ int swz_f(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

template <typename T, bool FAST, bool MIXED, bool S1, bool HALF>
__global__ __launch_bounds__(256, 2) void wgrad_mfma_kernel(WParams p) {
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
  // HALF (R == 64): all four waves work on rows 0..63 -- 2 column halves x 2 halves of every pixel step -- instead of
  // two of them multiplying the zero upper half of the row tile; the two pixel halves are summed in the epilogue
  const int wr = HALF ? 0 : (wave >> 1), wc = wave & 1;
  const int kh = wave >> 1;
  // XCD-contiguous order: all output tiles of one pixel split run on the same XCD and share the staged
  // operands through its L2 (blocks b and b+8 share an XCD)
  const int ntile = p.tiles_r * p.tiles_c;
  const int nblk = ntile * p.nsplit;
  const int bq = nblk >> 3, br = nblk & 7, bx = blockIdx.x & 7;
  const int lid = (bx < br ? bx * (bq + 1) : br * (bq + 1) + (bx - br) * bq) + (blockIdx.x >> 3);
  const int tile_c = lid % p.tiles_c;
  const int tile_r = (lid / p.tiles_c) % p.tiles_r;
  const int split = lid / ntile;
  const int Hs = p.Hs, Ws = p.Ws;
  constexpr bool s1 = S1;                                       // compile-time geometry: the k4 path keeps its constants
  const int Hl = s1 ? Hs : 2 * Hs, Wl = s1 ? Ws : 2 * Ws;      // grid of the gathered tensor
  const int kside = s1 ? p.ks : 4, kpad = s1 ? (p.ks >> 1) : 1, gstr = s1 ? 1 : 2;
  const int ntap = kside * kside;
  const bool has_pad = !s1 || p.ks == 3;
  const int C = p.C0 + p.C1;

  // LDS-DMA staging (global_load_lds_dwordx4): wave w writes 1 KiB = RPP/4 consecutive tile rows per pass,
  // lane l lands on row (l / CPRW) of that group, physical chunk (l % CPRW).  The granule swizzle is applied
  // on the SOURCE side: the lane fetches the logical chunk whose swizzled position is its physical chunk.
  // f(row) depends on the pass only through (row>>3)&1, which alternates with the pass for the f32 tile
  // (8 rows per pass), hence NV = 2 source variants there.
  const int pc = tid % CPRW;
  const int prow0 = tid / CPRW;
  constexpr int NV = sizeof(T) == 2 ? 1 : 2;
  const T* zero = reinterpret_cast<const T*>(adn_wg_zero_page);
  const T* psrc[NV];
  int Rsrc[NV];
  bool r_ok[NV];
  const T* gsrc[NV];
  int Csrc[NV], ky[NV], kx[NV];
  bool col_ok[NV];
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
    col_ok[v] = S1 ? (tap < ntap) : true;   // S1: the last column tile may be partial (9*C is not a multiple of 128)
    ky[v] = tap / kside;
    kx[v] = tap - ky[v] * kside;
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
  __amdgpu_buffer_rsrc_t rsp, rsg, rsg1;
  bool lsec[PASSES];
  constexpr bool mixed = MIXED;
  int Rs_u = 0, Cs_u = 0, lgWs = 0;
  if constexpr (FAST) {
    lgWs = 31 - __builtin_clz((unsigned)Ws);
    const bool psecond = tile_r * 128 >= p.R0;
    Rs_u = psecond ? p.R1 : p.R0;
    const char* pb = reinterpret_cast<const char*>(psecond ? p.plain1 : p.plain0);
    // a 128-column tile normally lies inside one gathered source; when it does not (C0 = C1 = 64, or several
    // taps of a narrow two-source concat per tile) the source is a per-lane constant and the two descriptors
    // are selected by an exec-masked branch
    const bool gsecond = !mixed && (C >= 128) && ((tile_c * 128) % C) >= p.C0;
    Cs_u = gsecond ? p.C1 : p.C0;
    const char* gb = reinterpret_cast<const char*>(gsecond ? p.gath1 : p.gath0) - (int64_t)kpad * (Wl + 1) * Cs_u * ESZ;
    rsp = __builtin_amdgcn_make_buffer_rsrc((void*)pb, 0, 0x7ffffff0, 0x00020000);
    rsg = __builtin_amdgcn_make_buffer_rsrc((void*)gb, 0, 0x7ffffff0, 0x00020000);
    rsg1 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<const char*>(p.C1 ? p.gath1 : p.gath0) - (int64_t)kpad * (Wl + 1) * p.C1 * ESZ), 0,
        0x7ffffff0, 0x00020000);
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
      int cch = gcol - tap * C - (gsecond ? p.C0 : 0);
      int Cs_l = Cs_u;
      lsec[k] = false;
      if (mixed && cch >= p.C0) {
        lsec[k] = true;
        cch -= p.C0;
        Cs_l = p.C1;
      }
      const int jx = r & (Ws - 1);
      // linear index of the gathered pixel at tap (kpad,kpad): s2: 4m - 2j, s1: m  (scalar part + lane part)
      const int L = s1 ? r : (rows_in_line ? 2 * r : 4 * r - 2 * jx);
      gvoff[k] = (unsigned)(((L + ky[v] * Wl + kx[v]) * Cs_l + cch) * ESZ);
      const int a = r >> lgWs;                    // image row inside the step (0 when rows_in_line)
      const int klast = kside - 1;
      m_y0[k] = has_pad && (ky[v] == 0) && (rows_in_line || a == 0);
      m_y1[k] = has_pad && (ky[v] == klast) && (rows_in_line || a == q - 1);
      m_x0[k] = has_pad && rows_in_line && (kx[v] == 0) && (r == 0);
      m_x1[k] = has_pad && rows_in_line && (kx[v] == klast) && (r == BKP - 1);
      m_c[k] = !col_ok[v] ||
               (has_pad && !rows_in_line && (((kx[v] == 0) && jx == 0) || ((kx[v] == klast) && jx == Ws - 1)));
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
      const int gpix = s1 ? m0 : 4 * m0 - 2 * sj;
      const int gsoff = gpix * Cs_u * ESZ;
      const int gsoff1 = gpix * p.C1 * ESZ;
#pragma unroll
      for (int k = 0; k < PASSES; ++k) {
        const bool inval = m_c[k] || (top && m_y0[k]) || (bot && m_y1[k]) || (left && m_x0[k]) || (right && m_x1[k]);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsp, (lptr_t)(pdst + k * (RPP * ROWB)), 16, pvoff[k], psoff, 0, 0);
        const unsigned gv = inval ? OOB : gvoff[k];
        if constexpr (MIXED) {
          // LDS-DMA lands at M0 base + lane * 16 for ACTIVE lanes only, so the two masked halves compose
          if (lsec[k]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsg1, (lptr_t)(gdst + k * (RPP * ROWB)), 16, gv, gsoff1, 0, 0);
          else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsg, (lptr_t)(gdst + k * (RPP * ROWB)), 16, gv, gsoff, 0, 0);
        } else {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsg, (lptr_t)(gdst + k * (RPP * ROWB)), 16, gv, gsoff, 0, 0);
        }
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
          const int iy = gstr * i - kpad + ky[v], ix = gstr * j - kpad + kx[v];
          if (col_ok[v] && (unsigned)iy < (unsigned)Hl && (unsigned)ix < (unsigned)Wl)
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
        if (HALF && ks != kh) continue;
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
        if (HALF && (ks >> 2) != kh) continue;
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
      for (int r = 0; r < 4; ++r) ct[((HALF ? kh : wr) * 64 + i * 16 + 4 * fg + r) * LDC + wc * 64 + j * 16 + fi] = acc[i][j][r];
  __syncthreads();
  float* out = p.out + (int64_t)split * p.out_elems;
  const int cq = tid & 31;    // float4 column group (32 per row)
  const int r0 = tid >> 5;    // 8 rows per pass
  const int R = p.R0 + p.R1;
  if (p.c_valid == C) {
    const int64_t ldo = (int64_t)ntap * C;
    const bool cok = S1 ? (tile_c * 128 + cq * 4 < ntap * C) : true;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int row = r0 + 8 * k;
      if (cok && tile_r * 128 + row < R) {
        f32x4_t v = *reinterpret_cast<const f32x4_t*>(ct + row * LDC + cq * 4);
        if (HALF) v += *reinterpret_cast<const f32x4_t*>(ct + (row + 64) * LDC + cq * 4);     // second pixel half
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
        if (tp < ntap && cc < p.c_valid)
          out[((int64_t)(tile_r * 128 + row) * ntap + tp) * p.c_valid + cc] =
              ct[row * LDC + cq * 4 + e] + (HALF ? ct[(row + 64) * LDC + cq * 4 + e] : 0.f);
      }
    }
  }
#endif
}

// generic path: one thread per (output element, split)
template <typename T>
__global__ __launch_bounds__(256) void wgrad_direct_kernel(WParams p, int pix_per_split) {
  const int Hs = p.Hs, Ws = p.Ws;
  const bool s1 = p.geom == ADN_GEMM_S1;
  const int Hl = s1 ? Hs : 2 * Hs, Wl = s1 ? Ws : 2 * Ws;
  const int kside = s1 ? p.ks : 4, kpad = s1 ? (p.ks >> 1) : 1, gstr = s1 ? 1 : 2;
  const int ntap = kside * kside;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= p.out_elems) return;
  const int split = blockIdx.y;
  const int cv = p.c_valid;
  const int c = (int)(e % cv);
  const int tap = (int)((e / cv) % ntap);
  const int r = (int)(e / ((int64_t)ntap * cv));
  const int ky = tap / kside, kx = tap - ky * kside;
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
    const int iy = gstr * i - kpad + ky, ix = gstr * j - kpad + kx;
    if ((unsigned)iy >= (unsigned)Hl || (unsigned)ix >= (unsigned)Wl) continue;
    const int64_t pix = ((int64_t)b * Hl + iy) * Wl + ix;
    acc += ElemTraits<T>::load(ps + (int64_t)m * Rs) * ElemTraits<T>::load(gs + pix * Cs);
  }
  p.out[(int64_t)split * p.out_elems + e] = acc;
}

__global__ __launch_bounds__(256) void slab_sum_kernel(const float* slab, float* out, int64_t n, int nsplit) {
  // n is a multiple of 4 (R*16*c with R % 64 == 0); 4 independent accumulator chains hide the load latency
  const int64_t n4 = n >> 2;
  const f32x4_t* s4 = reinterpret_cast<const f32x4_t*>(slab);
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
    reinterpret_cast<f32x4_t*>(out)[e] = (a0 + a1) + (a2 + a3);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t e = (n4 << 2) + threadIdx.x;
    float v = 0.f;
    for (int s = 0; s < nsplit; ++s) v += slab[(int64_t)s * n + e];
    out[e] = v;
  }
}

struct WPlan {
  bool mfma;
  bool fast;
  bool mixed;    // fast path whose 128-column tiles straddle the two gathered sources
  int nsplit, steps, tiles_r, tiles_c, pix_per_split;
  int64_t out_elems, slab_bytes;
};

void make_wplan(const AdnWgradDesc* d, WPlan* pl) {
  const int R = d->R0 + d->R1, C = d->C0 + d->C1;
  const int64_t msmall = (int64_t)d->B * d->Hs * d->Ws;
  const int cv = d->c_valid > 0 ? d->c_valid : C;
  const int ntap = d->geom == ADN_GEMM_S1 ? d->ks * d->ks : 16;
  pl->out_elems = (int64_t)R * ntap * cv;
  const int epc = d->dtype == ADN_BF16 ? 8 : 4;
  // sources are selected per 16-byte chunk, so a tile may straddle the two plain / gathered sources
  // tap, channel and source are per-lane constants of a 16-byte chunk, so a 128-column tile may straddle taps
  // (C = 192, 96, ...) as well as the two gathered sources; only chunks must not straddle anything
  const bool aligned = (R % 64 == 0) && (d->R0 % epc == 0) && (C % epc == 0) && (d->C0 % epc == 0);
  pl->mfma = aligned;
  pl->fast = false;
  pl->mixed = false;
  if (aligned) {
    const int bkp = d->dtype == ADN_BF16 ? 64 : 32;
    const int64_t esz = d->dtype == ADN_BF16 ? 2 : 4;
    auto pow2 = [](int x) { return x > 0 && (x & (x - 1)) == 0; };
    pl->fast = pow2(d->Hs) && pow2(d->Ws) && d->Hs * d->Ws >= bkp && (d->R1 == 0 || d->R0 % 128 == 0) &&
               msmall * (d->geom == ADN_GEMM_S1 ? 1 : 4) * (d->C0 > d->C1 ? d->C0 : d->C1) * esz < (1ll << 31) &&
               msmall * (d->R0 > d->R1 ? d->R0 : d->R1) * esz < (1ll << 31);     // 32-bit scalar byte offsets
    pl->mixed = pl->fast && d->C1 > 0 && (C % 128 != 0 || (d->C0 % 128) != 0);
    pl->steps = (int)adn_cdiv(msmall, bkp);
    pl->tiles_r = (int)adn_cdiv(R, 128);
    pl->tiles_c = (int)adn_cdiv((int64_t)ntap * C, 128);
    const int64_t tiles = (int64_t)pl->tiles_r * pl->tiles_c;
    // one resident wave of workgroups: 256 CUs x 2 workgroups; never 513 (9 taps x 57 splits ran a 2x tail)
    int ns = (int)(512 / tiles);
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
  ADN_CHECK_ARG(d->geom == 0 || (d->geom == ADN_GEMM_S1 && (d->ks == 1 || d->ks == 3)), "adn_wgrad: bad geom/ks %d/%d",
                d->geom, d->ks);
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
  p.geom = d->geom;
  p.ks = d->ks;
  if (pl.mfma) {
    constexpr int BKP = sizeof(T) == 2 ? 64 : 32;
    constexpr int stage = 4 * BKP * 128 * (int)sizeof(T);
    constexpr int epil = 128 * 132 * 4;
    constexpr int lds = stage > epil ? stage : epil;
    const dim3 grid(pl.tiles_r * pl.tiles_c * pl.nsplit);
    const bool s1 = d->geom == ADN_GEMM_S1;
    const bool half = (d->R0 + d->R1) == 64;
#define ADN_WG_LAUNCH1(FAST_, MIXED_, S1_, HALF_)                                                                      \
  do {                                                                                                         \
    ADN_SET_LDS_ONCE(lds, &wgrad_mfma_kernel<T, FAST_, MIXED_, S1_, HALF_>);                                   \
    hipLaunchKernelGGL((wgrad_mfma_kernel<T, FAST_, MIXED_, S1_, HALF_>), grid, dim3(256), lds, st, p);        \
  } while (0)
#define ADN_WG_LAUNCH(FAST_, MIXED_, S1_)                  \
  do {                                                     \
    if (half) ADN_WG_LAUNCH1(FAST_, MIXED_, S1_, true);    \
    else ADN_WG_LAUNCH1(FAST_, MIXED_, S1_, false);        \
  } while (0)
    if (pl.fast && pl.mixed) {
      if (s1) ADN_WG_LAUNCH(true, true, true);
      else ADN_WG_LAUNCH(true, true, false);
    } else if (pl.fast) {
      if (s1) ADN_WG_LAUNCH(true, false, true);
      else ADN_WG_LAUNCH(true, false, false);
    } else {
      if (s1) ADN_WG_LAUNCH(false, false, true);
      else ADN_WG_LAUNCH(false, false, false);
    }
#undef ADN_WG_LAUNCH
#undef ADN_WG_LAUNCH1
  } else {
    hipLaunchKernelGGL((wgrad_direct_kernel<T>), dim3((unsigned)adn_cdiv(pl.out_elems, 256), pl.nsplit), dim3(256),
                       0, st, p, pl.pix_per_split);
  }
  ADN_CHECK_LAUNCH();
  if (pl.nsplit > 1) {
    int64_t blocks = adn_cdiv(adn_cdiv(pl.out_elems, 4), 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, st,
                       reinterpret_cast<const float*>(d->workspace), d->dw, pl.out_elems, pl.nsplit);
    ADN_CHECK_LAUNCH();
  }
  return ADN_OK;
}

}  // namespace

int64_t adn_wgrad_k4_workspace_bytes(const AdnWgradDesc* d);   // wgrad_k4.hip
int adn_wgrad_k4(const AdnWgradDesc* d, void* stream);
bool adn_wgrad_s1p_plan(const AdnWgradDesc* d, int* nsplit, int64_t* out_elems);   // wgrad_s1p.hip (patch-staged 3 x 3)
int adn_wgrad_s1p_launch(const AdnWgradDesc* d, int nsplit, int64_t out_elems, void* stream);

int32_t adn_wgrad_k4_sq_count(const AdnWgradDesc* d);   // wgrad_k4.hip
int32_t adn_wgrad_k4_batchable(const AdnWgradDesc* d);
int32_t adn_wgrad_k4_batch_sq_count(const AdnWgradDesc* d);
int64_t adn_wgrad_k4_patch_batch_workspace_bytes(const AdnWgradDesc* descs, int32_t n);
int adn_wgrad_k4_patch_batch(const AdnWgradDesc* descs, int32_t n, void* stream);
int adn_wgrad_k4_batch(const AdnWgradDesc* descs, int32_t n, void* stream);

// Norm partials ride along only in the k4 pair's kernels (the U-Net baseline: 54 M parameters, the gradient pass the
// fusion saves is 40 us of a 2.9 ms step); the stride-1 kernels of the DoubleConv nets report 0 = "not fused".
extern "C" int32_t adn_wgrad_sq_count(const AdnWgradDesc* d) {
  if (!d || d->geom == ADN_GEMM_S1) return 0;
  return adn_wgrad_k4_sq_count(d);
}

extern "C" int32_t adn_wgrad_batchable(const AdnWgradDesc* d) {
  if (!d || d->geom == ADN_GEMM_S1) return 0;
  return adn_wgrad_k4_batchable(d);
}

extern "C" int64_t adn_wgrad_patch_batch_workspace_bytes(const AdnWgradDesc* descs, int32_t n) {
  for (int k = 0; descs && k < n && k < 4; ++k)
    if (descs[k].geom == ADN_GEMM_S1) return -1;
  return adn_wgrad_k4_patch_batch_workspace_bytes(descs, n);
}

extern "C" int adn_wgrad_patch_batch(const AdnWgradDesc* descs, int32_t n, void* stream) {
  return adn_wgrad_k4_patch_batch(descs, n, stream);
}

extern "C" int32_t adn_wgrad_batch_sq_count(const AdnWgradDesc* d) {
  if (!d || d->geom == ADN_GEMM_S1) return 0;
  return adn_wgrad_k4_batch_sq_count(d);
}

extern "C" int adn_wgrad_batch(const AdnWgradDesc* descs, int32_t n, void* stream) {
  return adn_wgrad_k4_batch(descs, n, stream);
}

extern "C" int64_t adn_wgrad_workspace_bytes(const AdnWgradDesc* d) {
  if (wvalidate(d) != ADN_OK) return -1;
  if (d->geom != ADN_GEMM_S1) return adn_wgrad_k4_workspace_bytes(d);
  int ns;
  int64_t oe;
  if (adn_wgrad_s1p_plan(d, &ns, &oe)) return ns > 1 ? (int64_t)ns * oe * 4 : 0;
  WPlan pl;
  make_wplan(d, &pl);
  return pl.slab_bytes;
}

extern "C" int adn_wgrad(const AdnWgradDesc* d, void* stream) {
  int rc = wvalidate(d);
  if (rc != ADN_OK) return rc;
  if (d->geom != ADN_GEMM_S1) return adn_wgrad_k4(d, stream);
  {
    int ns;
    int64_t oe;
    if (adn_wgrad_s1p_plan(d, &ns, &oe)) {
      const int64_t need = ns > 1 ? (int64_t)ns * oe * 4 : 0;
      ADN_CHECK_ARG(need == 0 || (d->workspace && d->workspace_bytes >= need), "adn_wgrad: workspace too small (%lld < %lld)",
                    (long long)d->workspace_bytes, (long long)need);
      rc = adn_wgrad_s1p_launch(d, ns, oe, stream);
      if (rc != ADN_OK) return rc;
      if (ns > 1) {
        int64_t blocks = adn_cdiv(adn_cdiv(oe, 4), 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                           reinterpret_cast<const float*>(d->workspace), d->dw, oe, ns);
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
