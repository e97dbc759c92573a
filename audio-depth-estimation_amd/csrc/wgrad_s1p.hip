// Patch-staged weight-gradient kernel of the 3 x 3 stride-1 convolutions (DoubleConv nets; bf16, gfx950).
//
//   dW[r][tap][c] = sum over pixels p of  dz[p][r] * in[p + tap][c]
//
// Why: the tap-staged kernel (wgrad.hip) lays the 9 taps out as column tiles, so every workgroup stages its own 64-pixel
// slice of BOTH operands per tap tile: at the 64-channel top level of the nets that is 2.5 KB of LDS-DMA per pixel for
// 256 B of algorithmic input -- the layers sit at 300-340 TFLOP/s, bound by L2 -> LDS staging.
// Here a workgroup owns a 64(r) x 64(c) block of the weight gradient for ALL 9 taps and walks 8 x 16 pixel tiles of the
// images: per tile it stages dz[128 px][64 r] (16 KiB) and the 10 x 18 input patch [180 px][64 c] (22.5 KiB) ONCE and
// takes the 9 taps as shifted transposed reads of that patch: 308 B of LDS-DMA per pixel and block.
// The contraction index (pixels) is the slow index of both NHWC operands: fragments are read with ds_read_b64_tr_b16
// exactly as in wgrad.hip (lane 4q+pp of a 16-lane group supplies pixel row 8 fg + q (+4), 4 channels at 4 pp of a
// 16-channel block).  LDS rows are 128 bytes (64 channels = four 32-byte granules); granule ^= f(row) with
// f = ((row >> 1) & 1) | ((row >> 3) & 1) << 1 makes the transposed reads conflict free for ANY row shift (a half wave
// reads rows {s .. s+3, s+8 .. s+11}: same-parity rows differ in bit 1 or bit 3), applied on the DMA source side.
// Wave w owns the 16 input channels 16 w .. 16 w + 15 of the block for all 9 taps and all 4 row tiles: 36 accumulator
// tiles (144 VGPRs); per 32-pixel K-step it reads the 4 dz fragments once and one patch fragment per tap (4 MFMAs each).
// Pixel tiles are split over the grid into f32 slabs that slab_sum_kernel (wgrad.hip) adds in fixed order (deterministic).
#include <stdlib.h>

#include <mutex>
#include <type_traits>

#include "adn_common.h"

namespace {

struct SParams {
  const void* dz;
  const void* in0;
  const void* in1;
  int R, C0, C1;
  int B, H, W;
  int nsplit, nrb, ncb;
  int tiles_total, tpr, tpi;       // pixel tiles in total, per image row of tiles, per image
  float* out;
  int64_t out_elems;
};

__device__ __forceinline__ int swz4(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); }

__global__ __launch_bounds__(256, 2) void wgrad_s1_patch_kernel(SParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int TH = 8, TW = 16, MW = TW + 2, PPIX = (TH + 2) * MW;     // 180 patch pixels
  constexpr int DZBUF = 128 * 128;                                      // 128 pixels x 64 channels bf16
  constexpr int PPIECES = (PPIX * 128 + 1023) / 1024;                   // 23 one-KiB pieces (8 pixels each)
  constexpr int PBUF = PPIECES * 1024;
  constexpr int PK = (PPIECES + 3) / 4;                                 // 6 per wave
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* DZl = smem;                    // [2][DZBUF]
  char* Pl = smem + 2 * DZBUF;         // [2][PBUF]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-contiguous order: the (row block, column block) pairs of one pixel split are neighbours on one XCD
  const int nblk = p.nrb * p.ncb;
  const int ngrid = nblk * p.nsplit;
  const int gq = ngrid >> 3, gr = ngrid & 7, gx = blockIdx.x & 7;
  const int lid = (gx < gr ? gx * (gq + 1) : gr * (gq + 1) + (gx - gr) * gq) + (blockIdx.x >> 3);
  const int blk = lid % nblk, split = lid / nblk;
  const int rb = blk / p.ncb, cb = blk - rb * p.ncb;
  const int r0 = rb * 64, c0 = cb * 64;
  const int H = p.H, W = p.W, R = p.R;
  const bool second = c0 >= p.C0;
  const int Cs = second ? p.C1 : p.C0;
  const int coff = second ? c0 - p.C0 : c0;

  const int t_begin = (int)(((int64_t)p.tiles_total * split) / p.nsplit);
  const int t_end = (int)(((int64_t)p.tiles_total * (split + 1)) / p.nsplit);

  // ---- loader geometry (lane constants) ----
  // piece = 8 LDS rows of 128 bytes; lane l -> row 8 pi + (l >> 3), physical 16-byte chunk l & 7; the lane fetches the
  // LOGICAL chunk whose swizzled position that is
  unsigned dvo[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int m = 8 * (wave + 4 * k) + (lane >> 3);                     // tile pixel 0..127
    const int chunk = ((((lane & 7) >> 1) ^ swz4(m)) << 1) | (lane & 1);
    dvo[k] = (unsigned)((((m >> 4) * W + (m & 15)) * R + r0 + chunk * 8) * 2);
  }
  unsigned pvo[PK];
  unsigned pmask = 0;          // per piece k: bit 5k + {0: top row, 1: bottom row, 2: left column, 3: right column, 4: beyond the patch}
  const int bshift = W + 1;
#pragma unroll
  for (int k = 0; k < PK; ++k) {
    const int q = 8 * (wave + 4 * k) + (lane >> 3);
    const int hr = q / MW, mc = q - hr * MW;
    const int chunk = ((((lane & 7) >> 1) ^ swz4(q)) << 1) | (lane & 1);
    pvo[k] = (unsigned)(((hr * W + mc) * Cs + coff + chunk * 8) * 2);
    const unsigned bits = (hr == 0 ? 1u : 0u) | (hr == TH + 1 ? 2u : 0u) | (mc == 0 ? 4u : 0u) | (mc == MW - 1 ? 8u : 0u) |
                          ((q >= PPIX || wave + 4 * k >= PPIECES) ? 16u : 0u);
    pmask |= bits << (5 * k);
  }
  typedef __attribute__((address_space(3))) void* lptr_t;
  const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)p.dz, 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsp = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const char*>(second ? p.in1 : p.in0) - (int64_t)bshift * Cs * 2), 0, 0x7ffffff0, 0x00020000);

  // tile t -> (image, tile row, tile column), advanced incrementally
  int tb = t_begin / p.tpi;
  int trem = t_begin - tb * p.tpi;
  int ty = trem / p.tpr, tx = trem - ty * p.tpr;

  auto issue = [&](int buf) {
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int pix0 = (tb * H + oy0) * W + ox0;
    const unsigned edge = (oy0 == 0 ? 1u : 0u) | (oy0 + TH == H ? 2u : 0u) | (ox0 == 0 ? 4u : 0u) | (ox0 + TW == W ? 8u : 0u) | 16u;
    char* dd = DZl + buf * DZBUF + wave * 1024;
    char* pd = Pl + buf * PBUF + wave * 1024;
    const int dso = pix0 * R * 2, pso = pix0 * Cs * 2;
#pragma unroll
    for (int k = 0; k < 4; ++k) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsd, (lptr_t)(dd + k * 4096), 16, dvo[k], dso, 0, 0);
#pragma unroll
    for (int k = 0; k < PK; ++k) {
      if (wave + 4 * k >= PPIECES) continue;
      const bool inval = ((pmask >> (5 * k)) & edge) != 0;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsp, (lptr_t)(pd + k * 4096), 16, inval ? OOB : pvo[k], pso, 0, 0);
    }
  };
  auto advance = [&]() {
    if (++tx == p.tpr) {
      tx = 0;
      if (++ty * p.tpr == p.tpi) {
        ty = 0;
        ++tb;
      }
    }
  };

  f32x4_t acc[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[t][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int fg = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  auto tr_frag = [&](const char* base, int row_lo, int gran) -> bf16x8_t {
    const int row_hi = row_lo + 4;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4_t*)(base + row_lo * 128 + ((gran ^ swz4(row_lo)) << 5) + pp * 8));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4_t*)(base + row_hi * 128 + ((gran ^ swz4(row_hi)) << 5) + pp * 8));
    s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return *reinterpret_cast<bf16x8_t*>(&v);
  };

  if (t_begin < t_end) issue(0);
  for (int t = t_begin; t < t_end; ++t) {
    const int cur = (t - t_begin) & 1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // own DMA landed, own reads of the previous tile returned
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (t + 1 < t_end) {
      advance();
      issue(cur ^ 1);
    }
    const char* Db = DZl + cur * DZBUF;
    const char* Pb = Pl + cur * PBUF;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      // opaque per-K-step copies of the lane's row bases: keeps the ~80 swizzled fragment addresses of the unrolled
      // body from being hoisted out of the tile loop (they would spill)
      int mrow = 32 * ks + 8 * fg + qq;                            // dz tile row of this lane's low half
      int prow = (2 * ks + (fg >> 1)) * MW + ((8 * fg + qq) & 15); // patch pixel of the same output pixel at tap (0, 0)
      asm volatile("" : "+v"(mrow), "+v"(prow));
      bf16x8_t af[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = tr_frag(Db, mrow, i);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const bf16x8_t bfr = tr_frag(Pb, prow + (tap / 3) * MW + (tap % 3), wave);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[tap][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr, acc[tap][i], 0, 0, 0);
      }
    }
    // the MFMA intrinsic has no side effects: without a "use" here the IR sink pass moves the tile's MFMAs out of the loop body
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(acc[tap][i]));
  }

  // ---- epilogue: slab [64 r][9][C] block, accumulator tile (tap, i): rows = r (lane >> 4) * 4 + reg, column = c lane & 15 ----
  float* out = p.out + (int64_t)split * p.out_elems;
  const int C = p.C0 + p.C1;
  const int64_t ldo = (int64_t)9 * C;
  const int col = c0 + 16 * wave + (lane & 15);
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        out[(int64_t)(r0 + i * 16 + 4 * fg + r) * ldo + tap * C + col] = acc[tap][i][r];
#endif
}

// ---- the k4 s2 p1 pair of the U-Net baseline (conv: plain = dZ on the small grid, gathered = layer input on the 2x grid;
// transposed conv: plain = layer input, gathered = dZ) -----------------------------------------------------------------
//   dW[r][tap][c] = sum over small-grid pixels (i, j) of  plain[i, j][r] * gath[2i - 1 + ky, 2j - 1 + kx][c]
// Same idea on 8 x 8 small-grid tiles: a workgroup owns 64(r) x 32(c) x 16 taps and stages, per tile, plain[64 px][64 r]
// (8 KiB) and the 18 x 18 gathered patch of 32 channels (20.25 KiB) once -- the tap-staged kernel (wgrad_k4.hip) moves
// 16 KiB + 16 KiB per 64 pixels for every one of its 128-column (tap, c) tiles.  The patch is stored as four planes by
// (row parity, column parity) of 9 x 9 pixels, 64-byte pixels: tap (ky, kx) of output pixel (y, x) is plane
// (ky & 1, kx & 1), pixel (y + (ky >> 1), x + (kx >> 1)), so the 4 consecutive output columns a lane group transposes
// are 4 consecutive LDS rows for every tap.  Granule (16 channels) ^= parity of the plane row: a half wave reads 4 rows
// of plane row h and 4 of plane row h + 1 -> conflict free.  Wave w: channels 16 (w & 1) .. + 15, kernel rows
// ky = 2 (w >> 1), 2 (w >> 1) + 1, all kx: 8 taps x 4 row tiles = 32 accumulator tiles.
constexpr unsigned OOB_K4 = 0x80000000u;

// One LDS-DMA request (1 KiB per wave) as inline assembly.  Why not __builtin_amdgcn_raw_ptr_buffer_load_lds here: once the
// fragment reads of the K loop are "base register + immediate" the compiler sees that the DMA and the ds_read_b64_tr_b16
// builtins touch the same LDS array and orders EVERY read behind the youngest request with s_waitcnt vmcnt(0) -- the next
// tile's staging would run in series with this tile's MFMAs.  The barrier protocol (vmcnt(0) + s_barrier per tile, two
// buffers) already orders them; hidden in assembly the request is outside the compiler's bookkeeping, counted by hand.
// M0 = LDS destination of the wave (written in the same statement that uses it, cdna_hip_programming.md 5.7).
__device__ __forceinline__ void k4_dma(unsigned voffset, u32x4_t rsrc, unsigned lds_addr, int soffset) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds"
               :
               : "v"(voffset), "s"(rsrc), "s"(lds_addr), "s"(soffset)
               : "memory");
}

struct KParams4 {
  const void* plain0;
  const void* plain1;
  const void* gath0;
  const void* gath1;
  int R0, R1, C0, C1;
  int B, Hs, Ws;
  int nsplit, nrb, ncb;
  int tiles_total, tpr, tpi;
  float* out;
  int64_t out_elems;
};

// SPREAD: the next tile's LDS-DMA requests are issued BETWEEN the MFMAs of the current tile (two straight-line regions pinned
// with sched_group_barrier) instead of in a burst behind the barrier: a request costs the issuing wave ~100 cycles, which the
// other waves of the SIMD fill with their MFMAs (round 3; the same change was worth 8 ... 20 % in the implicit GEMM)
// (body shared by the one-problem kernel and the multi-problem launch; `bid` = workgroup index inside its problem)
template <bool SPREAD>
__device__ __forceinline__ void wgrad_k4_patch_body(const KParams4& p, const int bid) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int TH = 8, TW = 8, PW = 9, PLANE = PW * PW, PPIX = 4 * PLANE;       // 324 patch pixels of 64 bytes
  constexpr int DZBUF = 64 * 128;
  constexpr int PPIECES = (PPIX * 64 + 1023) / 1024;                             // 21 (16 pixels each)
  constexpr int PK = (PPIECES + 3) / 4;                                          // 6 per wave
  constexpr int PBUF = 4 * PK * 1024;                                            // (24 pieces: every wave issues PK requests)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* DZl = smem;                    // [2][DZBUF]
  char* Pl = smem + 2 * DZBUF;         // [2][PBUF]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nblk = p.nrb * p.ncb;
  const int ngrid = nblk * p.nsplit;
  const int gq = ngrid >> 3, gr = ngrid & 7, gx = bid & 7;
  const int lid = (gx < gr ? gx * (gq + 1) : gr * (gq + 1) + (gx - gr) * gq) + (bid >> 3);
  const int blk = lid % nblk, split = lid / nblk;
  const int rb = blk / p.ncb, cb = blk - rb * p.ncb;
  const int r0 = rb * 64, c0 = cb * 32;
  const int Hs = p.Hs, Ws = p.Ws, Hl = 2 * Hs, Wl = 2 * Ws;
  const bool psecond = r0 >= p.R0;
  const int Rs = psecond ? p.R1 : p.R0;
  const int roff = psecond ? r0 - p.R0 : r0;
  const bool gsecond = c0 >= p.C0;
  const int Cs = gsecond ? p.C1 : p.C0;
  const int coff = gsecond ? c0 - p.C0 : c0;

  const int t_begin = (int)(((int64_t)p.tiles_total * split) / p.nsplit);
  const int t_end = (int)(((int64_t)p.tiles_total * (split + 1)) / p.nsplit);

  unsigned dvo[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int m = 8 * (wave + 4 * k) + (lane >> 3);                     // tile pixel 0..63 (row of 128 bytes)
    const int chunk = ((((lane & 7) >> 1) ^ swz4(m)) << 1) | (lane & 1);
    dvo[k] = (unsigned)((((m >> 3) * Ws + (m & 7)) * Rs + roff + chunk * 8) * 2);
  }
  unsigned pvo[PK];
  unsigned pmask = 0;          // per piece: bit 5k + {0: patch row 0, 1: patch row 17, 2: patch column 0, 3: column 17, 4: beyond}
  const int bshift = Wl + 1;
#pragma unroll
  for (int k = 0; k < PK; ++k) {
    const int q = 16 * (wave + 4 * k) + (lane >> 2);                    // LDS pixel (row of 64 bytes)
    const int plane = q / PLANE, rem = q - plane * PLANE;
    const int hr = rem / PW, mc = rem - hr * PW;
    const int pr = 2 * hr + (plane >> 1), pc = 2 * mc + (plane & 1);
    const int chunk = ((((lane & 3) >> 1) ^ (hr & 1)) << 1) | (lane & 1);
    pvo[k] = (unsigned)(((pr * Wl + pc) * Cs + coff + chunk * 8) * 2);
    const unsigned bits = (pr == 0 ? 1u : 0u) | (pr == 2 * TH + 1 ? 2u : 0u) | (pc == 0 ? 4u : 0u) | (pc == 2 * TW + 1 ? 8u : 0u) |
                          ((q >= PPIX || wave + 4 * k >= PPIECES) ? 16u : 0u);
    pmask |= bits << (5 * k);
  }

  // the same descriptors as plain SGPR quadruples for the inline-assembly requests (see k4_dma)
  auto desc_of = [](const void* ptr) -> u32x4_t {
    const unsigned long long a = (unsigned long long)ptr;
    return u32x4_t{(unsigned)a, (unsigned)(a >> 32) & 0xffffu, 0x7ffffff0u, 0x00020000u};
  };
  const u32x4_t dsd = desc_of(psecond ? p.plain1 : p.plain0);
  const u32x4_t dsp = desc_of(reinterpret_cast<const char*>(gsecond ? p.gath1 : p.gath0) - (int64_t)bshift * Cs * 2);
  int tb = t_begin / p.tpi;
  int trem = t_begin - tb * p.tpi;
  int ty = trem / p.tpr, tx = trem - ty * p.tpr;

  // requests K0 .. K1-1 of the tile at (tb, ty, tx) into buffer `buf` (index 0, 1 = the plain operand, 2 .. = patch pieces);
  // straight-line: the pieces beyond the patch (wave + 4k >= 21) are issued out of range (zeros into the 3 padding KiB)
  auto issue = [&](int buf, auto K0, auto K1, bool live) {
    constexpr int k0 = decltype(K0)::value, k1 = decltype(K1)::value;
    // (opaque buffer index: with a compile-time destination the compiler orders every later LDS read of the kernel behind the
    //  request with s_waitcnt vmcnt(0) -- it cannot know that the barrier protocol keeps the two buffers apart)
    asm volatile("" : "+s"(buf));
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const unsigned edge = (oy0 == 0 ? 1u : 0u) | (oy0 + TH == Hs ? 2u : 0u) | (ox0 == 0 ? 4u : 0u) | (ox0 + TW == Ws ? 8u : 0u) | 16u;
    char* dd = DZl + buf * DZBUF + wave * 1024;
    char* pd = Pl + buf * PBUF + wave * 1024;
    const int dso = ((tb * Hs + oy0) * Ws + ox0) * Rs * 2;
    const int pso = ((tb * Hl + 2 * oy0) * Wl + 2 * ox0) * Cs * 2;
#pragma unroll
    for (int k = k0; k < k1; ++k) {
      if (k < 2) {
        k4_dma(live ? dvo[k] : OOB_K4, dsd, lds0 + (unsigned)(dd - smem) + k * 4096, dso);
      } else {
        const int kk = k - 2;
        const bool inval = !live || ((pmask >> (5 * kk)) & edge) != 0;
        k4_dma(inval ? OOB_K4 : pvo[kk], dsp, lds0 + (unsigned)(pd - smem) + kk * 4096, pso);
      }
    }
  };
  auto advance = [&]() {
    if (++tx == p.tpr) {
      tx = 0;
      if (++ty * p.tpr == p.tpi) {
        ty = 0;
        ++tb;
      }
    }
  };

  f32x4_t acc[8][4];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[t][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int fg = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
  const int wct = wave & 1, wky = 2 * (wave >> 1);
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  auto tr_pair = [&](const char* lo_p, const char* hi_p) -> bf16x8_t {
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)lo_p);
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)hi_p);
    s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return *reinterpret_cast<bf16x8_t*>(&v);
  };

  // ---- fragment addresses: per-lane base registers + immediates (round 3) ----
  // Round 2 recomputed every swizzled address per K-step (an opaque copy of the row stopped the compiler from hoisting all 48
  // of them into registers): ~6 VALU instructions per MFMA, the kernel was VALU-issue bound.  Both swizzles only depend on
  // bits the K-step does not touch, so one base per plain fragment column (4) and ONE for the patch suffice; the buffer
  // (tile parity), the K-step, the tap's plane / row / column are immediates of the transposed reads.
  unsigned abase[4];
  {
    const int m0 = 8 * fg + qq;                     // plain tile row of the low half at K-step 0 (the high half is + 4 rows)
#pragma unroll
    for (int i = 0; i < 4; ++i) abase[i] = (unsigned)(m0 * 128 + ((i ^ swz4(m0)) << 5) + pp * 8);
  }
  const int kyh = wky >> 1;                         // (ky >> 1) is the same for both kernel rows of this wave
  const unsigned pbase = (unsigned)(2 * DZBUF + ((fg + kyh) * PW + qq) * 64 + ((wct ^ ((fg + kyh) & 1)) << 5) + pp * 8);

  using I0 = std::integral_constant<int, 0>;
  using I8 = std::integral_constant<int, 2 + PK>;
  // one pixel tile out of buffer CUR; the next tile's requests go into buffer CUR ^ 1
  auto tile_body = [&](auto CUR, bool more) {
    constexpr int cur = decltype(CUR)::value;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (more) advance();
    if constexpr (!SPREAD) {
      if (more) issue(cur ^ 1, I0{}, I8{}, true);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t af[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const char* a = smem + abase[i] + (cur * DZBUF + ks * 32 * 128);
        af[i] = tr_pair(a, a + 4 * 128);
      }
      auto read_tap = [&](int tl) -> bf16x8_t {
        const int dky = tl >> 2, kx = tl & 3;          // ky = wky + dky
        const int imm = cur * PBUF + ((((dky << 1) | (kx & 1)) * PLANE + 4 * ks * PW + (kx >> 1)) * 64);
        const char* a = smem + pbase + imm;
        return tr_pair(a, a + 4 * 64);
      };
      // the tap fragments one tap ahead of the MFMAs; one request of the next tile behind every second tap (the inline
      // assembly keeps its place in the instruction stream: that IS the interleave)
      bf16x8_t bq[2];
      bq[0] = read_tap(0);
#pragma unroll
      for (int tl = 0; tl < 8; ++tl) {
        if (tl + 1 < 8) bq[(tl + 1) & 1] = read_tap(tl + 1);
        if constexpr (SPREAD) {
          if (tl & 1) {
            const int rq = 4 * ks + (tl >> 1);          // request 0 .. 7 of the next tile (2 plain + 6 patch pieces per wave)
            switch (rq) {
              case 0: issue(cur ^ 1, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, more); break;
              case 1: issue(cur ^ 1, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{}, more); break;
              case 2: issue(cur ^ 1, std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{}, more); break;
              case 3: issue(cur ^ 1, std::integral_constant<int, 3>{}, std::integral_constant<int, 4>{}, more); break;
              case 4: issue(cur ^ 1, std::integral_constant<int, 4>{}, std::integral_constant<int, 5>{}, more); break;
              case 5: issue(cur ^ 1, std::integral_constant<int, 5>{}, std::integral_constant<int, 6>{}, more); break;
              case 6: issue(cur ^ 1, std::integral_constant<int, 6>{}, std::integral_constant<int, 7>{}, more); break;
              default: issue(cur ^ 1, std::integral_constant<int, 7>{}, std::integral_constant<int, 8>{}, more); break;
            }
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[tl][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bq[tl & 1], acc[tl][i], 0, 0, 0);
      }
    }
  };
  if (t_begin < t_end) issue(0, I0{}, I8{}, true);
  for (int t = t_begin; t < t_end; t += 2) {
    tile_body(std::integral_constant<int, 0>{}, t + 1 < t_end);
    if (t + 1 < t_end) tile_body(std::integral_constant<int, 1>{}, t + 2 < t_end);
  }

  float* out = p.out + (int64_t)split * p.out_elems;
  const int C = p.C0 + p.C1;
  const int64_t ldo = (int64_t)16 * C;
  const int col = c0 + 16 * wct + (lane & 15);
#pragma unroll
  for (int tl = 0; tl < 8; ++tl) {
    const int tap = (wky + (tl >> 2)) * 4 + (tl & 3);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(int64_t)(r0 + i * 16 + 4 * fg + r) * ldo + tap * C + col] = acc[tl][i][r];
  }
#endif
}

template <bool SPREAD>
__global__ __launch_bounds__(256, 2) void wgrad_k4_patch_kernel(KParams4 p) {
  wgrad_k4_patch_body<SPREAD>(p, (int)blockIdx.x);
}

// Several layers in one launch, each with FEWER pixel splits than a lone launch would take: the slab bytes of a launch are
// (workgroups x 128 KB) whatever the layer, so three layers sharing 512 workgroups write a third of the slabs each.
constexpr int kK4BatchMax = 4;
struct K4Batch {
  KParams4 p[kK4BatchMax];
  int first[kK4BatchMax + 1];
  int n;
};
__global__ __launch_bounds__(256, 2) void wgrad_k4_patch_batch_kernel(K4Batch b) {
  int k = 0;
#pragma unroll
  for (int j = 1; j < kK4BatchMax; ++j)
    if (j < b.n && (int)blockIdx.x >= b.first[j]) k = j;
  wgrad_k4_patch_body<false>(b.p[k], (int)blockIdx.x - b.first[k]);
}

// ADN_WGRAD_PATCH=0 switches the patch-staged kernels off; ADN_WGRAD_WGS = workgroups a launch aims at (default 512 = one
// resident wave of 2 per CU; every workgroup writes its own f32 slab block, so the slab traffic is proportional to it).
int g_wgs = 512;
int g_spread = 0;      // ADN_WGRAD_SPREAD=1: the k4 patch kernel issues its requests between the MFMAs instead of in a burst
                       // behind the barrier (measured: 5 ... 15 % slower here -- each request re-derives its border mask)
bool patch_enabled() {
  static int on = 1;
  static std::once_flag once;
  std::call_once(once, [] {
    if (const char* e = getenv("ADN_WGRAD_PATCH")) on = atoi(e);
    if (const char* e = getenv("ADN_WGRAD_WGS")) g_wgs = atoi(e) > 0 ? atoi(e) : 512;
    if (const char* e = getenv("ADN_WGRAD_SPREAD")) g_spread = atoi(e);
  });
  return on != 0;
}

}  // namespace

// Eligibility + plan: bf16, 3 x 3, one plain source, every channel count a multiple of 64, images tileable by 8 x 16.
bool adn_wgrad_s1p_plan(const AdnWgradDesc* d, int* nsplit, int64_t* out_elems) {
  if (!patch_enabled()) return false;
  if (d->dtype != ADN_BF16 || d->geom != ADN_GEMM_S1 || d->ks != 3) return false;
  if (d->R1 != 0 || d->R0 % 64 != 0 || d->C0 % 64 != 0 || d->C1 % 64 != 0) return false;
  if (d->Hs % 8 != 0 || d->Ws % 16 != 0) return false;
  const int C = d->C0 + d->C1;
  if (d->c_valid != 0 && d->c_valid != C) return false;
  const int64_t pix = (int64_t)d->B * d->Hs * d->Ws;
  if (pix * d->R0 * 2 >= 0x7ff00000ll || pix * (d->C0 > d->C1 ? d->C0 : d->C1) * 2 >= 0x7ff00000ll) return false;
  const int64_t tiles = pix / 128;
  const int nblk = (d->R0 / 64) * (C / 64);
  int ns = g_wgs / nblk;                      // one resident wave of workgroups (256 CUs x 2)
  if (ns > tiles / 2) ns = (int)(tiles / 2);
  if (ns < 1) ns = 1;
  *nsplit = ns;
  *out_elems = (int64_t)d->R0 * 9 * C;
  return true;
}

int adn_wgrad_s1p_launch(const AdnWgradDesc* d, int nsplit, int64_t out_elems, void* stream) {
  SParams p;
  p.dz = d->plain0;
  p.in0 = d->gath0;
  p.in1 = d->gath1;
  p.R = d->R0;
  p.C0 = d->C0;
  p.C1 = d->C1;
  p.B = d->B;
  p.H = d->Hs;
  p.W = d->Ws;
  p.nsplit = nsplit;
  p.nrb = d->R0 / 64;
  p.ncb = (d->C0 + d->C1) / 64;
  p.tpr = d->Ws / 16;
  p.tpi = (d->Hs / 8) * p.tpr;
  p.tiles_total = d->B * p.tpi;
  p.out = nsplit > 1 ? reinterpret_cast<float*>(d->workspace) : d->dw;
  p.out_elems = out_elems;
  constexpr int lds = 2 * (128 * 128 + 23 * 1024);
  ADN_SET_LDS_ONCE(lds, &wgrad_s1_patch_kernel);
  hipLaunchKernelGGL(wgrad_s1_patch_kernel, dim3(p.nrb * p.ncb * nsplit), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), p);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

// k4 s2 p1 pair: bf16, every channel count a multiple of 64 (gathered: 32), small grid tileable by 8 x 8.
bool adn_wgrad_k4p_plan(const AdnWgradDesc* d, int* nsplit, int64_t* out_elems) {
  if (!patch_enabled()) return false;
  if (d->dtype != ADN_BF16 || d->geom != 0) return false;
  if (d->R0 % 64 != 0 || d->R1 % 64 != 0 || d->C0 % 32 != 0 || d->C1 % 32 != 0) return false;
  if (d->Hs % 8 != 0 || d->Ws % 8 != 0) return false;
  const int C = d->C0 + d->C1, R = d->R0 + d->R1;
  if (d->c_valid != 0 && d->c_valid != C) return false;
  const int64_t pix = (int64_t)d->B * d->Hs * d->Ws;
  if (pix * (d->R0 > d->R1 ? d->R0 : d->R1) * 2 >= 0x7ff00000ll || 4 * pix * (d->C0 > d->C1 ? d->C0 : d->C1) * 2 >= 0x7ff00000ll)
    return false;
  const int64_t tiles = pix / 64;
  const int nblk = (R / 64) * (C / 32);
  if (tiles < 64) return false;               // the innermost levels (<= 4 x 4 images at B = 32): tap-staged split-K kernel
  int ns = g_wgs / nblk;
  if (ns > tiles / 4) ns = (int)(tiles / 4);
  if (ns < 1) ns = 1;
  *nsplit = ns;
  *out_elems = (int64_t)R * 16 * C;
  return true;
}

static void fill_k4p(const AdnWgradDesc* d, int nsplit, int64_t out_elems, float* outp, KParams4& p) {
  p.plain0 = d->plain0;
  p.plain1 = d->plain1;
  p.gath0 = d->gath0;
  p.gath1 = d->gath1;
  p.R0 = d->R0;
  p.R1 = d->R1;
  p.C0 = d->C0;
  p.C1 = d->C1;
  p.B = d->B;
  p.Hs = d->Hs;
  p.Ws = d->Ws;
  p.nsplit = nsplit;
  p.nrb = (d->R0 + d->R1) / 64;
  p.ncb = (d->C0 + d->C1) / 32;
  p.tpr = d->Ws / 8;
  p.tpi = (d->Hs / 8) * p.tpr;
  p.tiles_total = d->B * p.tpi;
  p.out = outp;
  p.out_elems = out_elems;
}

// n <= 4 problems in one launch; nsplit[k] pixel splits and the slab (or, unsplit, dW) base out[k] per problem
int adn_wgrad_k4p_batch_launch(const AdnWgradDesc* descs, int n, const int* nsplit, float* const* out, void* stream) {
  ADN_CHECK_ARG(n >= 1 && n <= kK4BatchMax, "adn_wgrad_patch_batch: 1 .. %d problems", kK4BatchMax);
  K4Batch b;
  b.n = n;
  b.first[0] = 0;
  for (int k = 0; k < n; ++k) {
    const AdnWgradDesc* d = descs + k;
    fill_k4p(d, nsplit[k], (int64_t)(d->R0 + d->R1) * 16 * (d->C0 + d->C1), out[k], b.p[k]);
    b.first[k + 1] = b.first[k] + b.p[k].nrb * b.p[k].ncb * nsplit[k];
  }
  for (int k = n; k < kK4BatchMax; ++k) {
    b.p[k] = b.p[0];
    b.first[k + 1] = b.first[n];
  }
  constexpr int lds = 2 * (64 * 128 + 24 * 1024);
  ADN_SET_LDS_ONCE(lds, &wgrad_k4_patch_batch_kernel);
  hipLaunchKernelGGL(wgrad_k4_patch_batch_kernel, dim3((unsigned)b.first[n]), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), b);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

int adn_wgrad_k4p_launch(const AdnWgradDesc* d, int nsplit, int64_t out_elems, void* stream) {
  KParams4 p;
  p.plain0 = d->plain0;
  p.plain1 = d->plain1;
  p.gath0 = d->gath0;
  p.gath1 = d->gath1;
  p.R0 = d->R0;
  p.R1 = d->R1;
  p.C0 = d->C0;
  p.C1 = d->C1;
  p.B = d->B;
  p.Hs = d->Hs;
  p.Ws = d->Ws;
  p.nsplit = nsplit;
  p.nrb = (d->R0 + d->R1) / 64;
  p.ncb = (d->C0 + d->C1) / 32;
  p.tpr = d->Ws / 8;
  p.tpi = (d->Hs / 8) * p.tpr;
  p.tiles_total = d->B * p.tpi;
  p.out = nsplit > 1 ? reinterpret_cast<float*>(d->workspace) : d->dw;
  p.out_elems = out_elems;
  constexpr int lds = 2 * (64 * 128 + 24 * 1024);
  if (g_spread) {
    ADN_SET_LDS_ONCE(lds, &wgrad_k4_patch_kernel<true>);
    hipLaunchKernelGGL(wgrad_k4_patch_kernel<true>, dim3(p.nrb * p.ncb * nsplit), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), p);
  } else {
    ADN_SET_LDS_ONCE(lds, &wgrad_k4_patch_kernel<false>);
    hipLaunchKernelGGL(wgrad_k4_patch_kernel<false>, dim3(p.nrb * p.ncb * nsplit), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), p);
  }
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}
