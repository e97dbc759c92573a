// Implicit-GEMM convolution family for k4/s2/p1 Conv2d and ConvTranspose2d (gfx950).
//
//   S2:  out[m][n]        = sum_{tap=(ky,kx), c} in[b, 2oy-1+ky, 2ox-1+kx, c] * w[n][tap][c]     m on the small grid
//   T2:  out[phase, m][n] = sum_{t=(ty,tx), c}  in[b, i+DY(ph,ty), j+DY(pw,tx), c] * w[phase][n][t][c]
//        written to the large-grid pixel (2i+ph, 2j+pw)
//
// MFMA path: 128 x BN output tile per 256-thread workgroup (4 waves as 2x2, each 64 x BN/2),
// K-step = 128 bytes of K (64 bf16 / 32 f32); both operands go global -> LDS by LDS-DMA
// (buffer_load_dwordx4 ... lds: no VGPR staging, no ds_write, zero padding by the hardware range check),
// 2- or 3-stage ring: the DMA of the next step(s) is in flight while step s runs on the matrix cores.  LDS image is XOR swizzled (16-byte chunk ^= (row>>1)&7,
// applied on the per-lane SOURCE address since the DMA destination is lane-linear) so that the
// ds_read_b128 fragment reads are bank-conflict free.  Small-M layers use
// split-K into f32 slabs + a reduce pass.  The epilogue goes through LDS so that every global
// access of the epilogue is a 16-byte, row-contiguous access.
#include <stdarg.h>
#include <stdlib.h>

#include <mutex>

#include "epilogue.h"

namespace {

// Tuning / diagnosis knobs, read from the environment ONCE (not per launch):
//   ADN_IGEMM_BM / ADN_IGEMM_BN / ADN_IGEMM_NS  force the tile rows / columns, cap the split-K count
//   ADN_IGEMM_NOA / ADN_IGEMM_NOB               timing-only builds (make FLAGS_igemm=-DADN_TIMING_KNOBS; NOT in the shipped
//                                               library): the gathered / weight operand is read through a zero-record
//                                               descriptor (every load dropped, zeros in LDS; wrong results)
//   ADN_IGEMM_SKIP                              same builds: bit 0 / 1 = the operand's LDS-DMA is not issued at all
struct Tune {
  int bm = 0, bn = 0, ns = 0, noa = 0, nob = 0, skip = 0, patch = 1, tall = 1, pair = 1, tinycap = 4, bn_t2 = 0, onepx = 1, ring = 1, ring_sched = 1, ring_epi = 0, ring_geom = -1, ring_hs = 0, ring64n = 1;
};
const Tune& tune() {
  static Tune t;
  static std::once_flag once;
  std::call_once(once, [] {
    if (const char* e = getenv("ADN_IGEMM_BM")) t.bm = atoi(e);
    if (const char* e = getenv("ADN_IGEMM_BN")) t.bn = atoi(e);
    if (const char* e = getenv("ADN_IGEMM_NS")) t.ns = atoi(e);
#ifdef ADN_TIMING_KNOBS      // timing-only builds (wrong results): never compiled into the shipped library
    if (const char* e = getenv("ADN_IGEMM_NOA")) t.noa = atoi(e);
    if (const char* e = getenv("ADN_IGEMM_NOB")) t.nob = atoi(e);
    if (const char* e = getenv("ADN_IGEMM_SKIP")) t.skip = atoi(e);
#endif
    if (const char* e = getenv("ADN_IGEMM_PATCH")) t.patch = atoi(e);
    if (const char* e = getenv("ADN_IGEMM_TALL")) t.tall = atoi(e);
    if (const char* e = getenv("ADN_IGEMM_PAIR")) t.pair = atoi(e);
    if (const char* e = getenv("ADN_IGEMM_TINYCAP")) t.tinycap = atoi(e);
    if (const char* e = getenv("ADN_IGEMM_ONEPX")) t.onepx = atoi(e);
    if (const char* e = getenv("ADN_IGEMM_BN_T2")) t.bn_t2 = atoi(e);
    if (const char* e = getenv("ADN_IGEMM_RING")) t.ring = atoi(e);
    if (const char* e = getenv("ADN_IGEMM_RING_SCHED")) t.ring_sched = atoi(e);
    if (const char* e = getenv("ADN_IGEMM_RING_EPI")) t.ring_epi = atoi(e);      // A/B: ring kernel only for this epilogue (1 | 3)
    if (const char* e = getenv("ADN_IGEMM_RING_GEOM")) t.ring_geom = atoi(e);    // A/B: ring kernel only for this geometry (0 | 1)
    if (const char* e = getenv("ADN_IGEMM_RING_HS")) t.ring_hs = atoi(e);
    if (const char* e = getenv("ADN_IGEMM_RING_N64")) t.ring64n = atoi(e);       // A/B: 64-column layers on the ring kernel too        // A/B: ring kernel only at this small-grid height
  });
  return t;
}

struct KParams {
  const void* in0;
  const void* in1;
  const void* w;
  int B, Hs, Ws, C0, C1, N;
  int ks;         // S1 geometry: kernel side (1 or 3)
  int wstride;    // elements per packed-weight row (taps*Cin rounded up to the K-step)
  int Msmall;     // B*Hs*Ws
  int kpt;        // K-steps per tap
  int ksteps;     // total K-steps
  int nsplit;     // split-K factor (grid.z)
  int tiles_m, tiles_n;
  int epi;
  AdnEpiSeg seg[2];
  float* slab;    // split-K / generic scratch
  unsigned rec_a, rec_b;   // buffer-descriptor record bytes of the gathered / weight operands (0 = timing-only build)
  int skip;                // timing-only builds: bit 0 / 1 = do not even ISSUE the gathered / weight operand's LDS-DMA
  int onepx;               // 1 x 1 small-grid images (the innermost U-Net level): only the taps that can be in range are walked
  int lgW, lgH;            // log2 of Ws / Hs when both are powers of two (every U-Net level), else -1: the pixel decode
                           // of the tile prologue / epilogue then uses shifts instead of ~40-instruction integer divisions
};


__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  // bijective XCD-contiguous remap (cdna_hip_programming.md T1): blocks b and b+8 share an XCD.
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// In front of a raw s_barrier that hands an LDS buffer back to the LDS-DMA: this wave's DMA of the awaited stage has
// landed (vmcnt) AND its own ds_reads of the previous step have returned (lgkmcnt(0)).  Without the second half the
// scheduler leaves the last ds_reads of a step in flight across the barrier (their MFMAs sink below it); a fast wave's
// DMA into that buffer -- weights are L2 hits, a few hundred cycles -- then overtakes them in a busy LDS queue
// (seen as one wrong 64-row sub-tile in roughly one of 16 384 workgroups of the S1 patch kernel, tools/diag_s1.py).
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}

// Tile configurations (waves are BM/64 x NWN, each wave owns a 64 x BN/NWN sub-tile):
//   128 x 128, 2x2 waves, 2 LDS stages, 2 workgroups per CU   small-M / split-K / short-K layers
//   128 x  64, 2x2 waves, 2 stages                             narrow N with small M
//   256 x  64, 4x1 waves, 2 stages, 2 workgroups per CU        N = 64: every wave keeps a 64x64 sub-tile
//   256 x 128, 4x2 waves, 3 stages, 1 workgroup per CU         long K: the DMA of steps s+1 and s+2 stays in
//             flight across the per-step barrier (counted s_waitcnt vmcnt, raw s_barrier)
// (PRE = false, an instantiation without the backward-epilogue prefetch registers for forward / split-K launches, was timed
//  in round 3 on the small-image layers: L5 / L6 / D5 forward 20.7 / 16.6 / 32.3 -> 19.9 / 15.7 / 31.4 us, the rest +-1 %:
//  inside the noise, so the template parameter stays but only PRE = true is instantiated)
template <typename T, int BM, int BN, int NWN, int GEOM, bool WIDE, bool PRE = true>
__global__ __launch_bounds__(BM * NWN, 2) void igemm_mfma_kernel(KParams p) {
#if defined(__HIP_DEVICE_COMPILE__)   // buffer-resource builtins exist only in the device pass; the host needs the stub only
  constexpr int NTHR = BM * NWN;            // 64 threads per 64 x (BN/NWN) wave tile
  constexpr int NW = NTHR / 64;
  constexpr int STAGES = (BM == 256 && NWN == 2) ? 3 : 2;
  constexpr int EPC = 16 / (int)sizeof(T);  // elements per 16-byte chunk
  constexpr int BK = 8 * EPC;               // elements per K-step (128 bytes)
  constexpr int WN = BN / NWN;              // wave tile columns
  constexpr int NT = WN / 16;
  constexpr int MT = 4;
  constexpr int RPASS = NTHR / 8;           // tile rows filled per loader pass (8 rows per wave-instruction)
  constexpr int APASS = BM / RPASS;         // = 4
  constexpr int BPASS = BN / RPASS;
  constexpr int LOADS = APASS + BPASS;      // LDS-DMA instructions per wave per stage
  constexpr int STAGE_BYTES = (BM + BN) * 128;
  constexpr int LDC = BN + 4;               // epilogue tile leading dimension (floats)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  // XCD-contiguous block order with the 4 transposed-conv phases of a tile adjacent (they gather the same
  // input pixels: measured 5x re-fetch of the input from beyond L2 when phases were separate grid slices)
  constexpr int NPH = GEOM == ADN_GEMM_T2 ? 4 : 1;
  const int nwg = p.tiles_m * p.tiles_n * NPH;
  const int wg0 = xcd_remap(blockIdx.x, nwg);
  const int phase = wg0 % NPH;
  const int wg = wg0 / NPH;
  const int tile_n = wg % p.tiles_n;
  const int tile_m = wg / p.tiles_n;
  const int ph = phase >> 1, pw = phase & 1;
  const int split = blockIdx.z;

  const int Hs = p.Hs, Ws = p.Ws;
  const int Hl = 2 * Hs, Wl = 2 * Ws;
  const int Cin = p.C0 + p.C1;
  const bool pow2 = p.lgW >= 0;                      // kernel-uniform
  // small-grid pixel m -> (image b, row y, column x)
  auto decode = [&](int m, int& b, int& y, int& x) {
    if (pow2) {
      x = m & (Ws - 1);
      y = (m >> p.lgW) & (Hs - 1);
      b = m >> (p.lgW + p.lgH);
    } else {
      b = m / (Hs * Ws);
      const int rem = m - b * (Hs * Ws);
      y = rem / Ws;
      x = rem - y * Ws;
    }
  };

  // ---- loader geometry (LDS-DMA: buffer_load_dwordx4 ... lds writes wave-uniform base + lane*16) ----
  // Wave w fills rows j*RPASS + w*8 .. +8 of each tile with ONE 1-KiB wave-instruction: lane l lands on row
  // (l>>3), physical 16-byte chunk (l&7).  The XOR swizzle therefore goes on the SOURCE side: the lane
  // fetches logical chunk pc ^ ((row>>1)&7) (same for all j because RPASS rows keep (row>>1)&7).
  //
  // Addressing is split so that the K loop costs (almost) no vector instructions (the first version
  // spent 4.5 VALU per MFMA on 64-bit address arithmetic and was issue bound at ~25 % MFMA utilisation):
  //   address = descriptor base (SGPRs, shifted back by one row + one pixel so tap offsets are >= 0)
  //           + voffset (per lane, CONSTANT over the K loop: pixel (2y,2x) resp. (y,x) of the row + chunk)
  //           + soffset (scalar per K-step: tap shift + channel offset)
  // Padding taps and rows beyond M use voffset = 0x80000000: the hardware range check fails and the DMA
  // writes zeros into LDS (verified on gfx950), so zero padding needs no memory at all.
  constexpr unsigned OOB = 0x80000000u;
  constexpr int ESZ = (int)sizeof(T);
  const int pc = tid & 7;
  const int lrow = tid >> 3;
  const int lc = pc ^ ((lrow >> 1) & 7);
  const int Wg = (GEOM == ADN_GEMM_S2) ? Wl : Ws;
  const int ks = p.ks, kpad = p.ks >> 1;                     // S1: kernel side and padding
  const int ntaps = GEOM == ADN_GEMM_S2 ? 16 : (GEOM == ADN_GEMM_T2 ? 4 : ks * ks);
  const int ktot = p.wstride;
  // the descriptor base is shifted back so that every tap offset is >= 0: one row + one pixel for the
  // k4 geometries, kpad rows + kpad pixels for the same-grid S1 geometry
  const int bshift = (GEOM == ADN_GEMM_S1) ? kpad * (Ws + 1) : (Wg + 1);

  unsigned roff0[APASS], roff1[APASS], amask[APASS];
#pragma unroll
  for (int j = 0; j < APASS; ++j) {
    const int m = tile_m * BM + lrow + RPASS * j;
    const bool okm = m < p.Msmall;
    const int mm = okm ? m : 0;
    int b, y, x;
    decode(mm, b, y, x);
    unsigned mask = 0;
    int pix;
    // tap validity bit masks: a tap is padding only at the image border, so the 16 (S2) / 4 (T2) per-tap range checks
    // reduce to four border compares selecting constant bit groups
    if constexpr (GEOM == ADN_GEMM_S2) {
      pix = (b * Hl + 2 * y) * Wl + 2 * x;                   // tap (ky,kx) adds (ky-1)*Wl + (kx-1)
      mask = 0xffffu & ~((y == 0 ? 0x000fu : 0u) | (y == Hs - 1 ? 0xf000u : 0u) | (x == 0 ? 0x1111u : 0u) |
                         (x == Ws - 1 ? 0x8888u : 0u));
    } else if constexpr (GEOM == ADN_GEMM_T2) {
      pix = (b * Hs + y) * Ws + x;                            // tap (ty,tx) adds dy*Ws + dx
      // phase bit 0: tap index 1 reaches back (dy = -1), phase bit 1: tap index 0 reaches forward (dy = +1)
      const unsigned ybad = ph == 0 ? (y == 0 ? 0xcu : 0u) : (y == Hs - 1 ? 0x3u : 0u);          // t = ty*2 + tx
      const unsigned xbad = pw == 0 ? (x == 0 ? 0xau : 0u) : (x == Ws - 1 ? 0x5u : 0u);
      mask = 0xfu & ~(ybad | xbad);
    } else {
      pix = (b * Hs + y) * Ws + x;                            // tap (ky,kx) adds (ky-kpad)*Ws + (kx-kpad)
      for (int t = 0; t < ntaps; ++t) {
        const int iy = y + t / ks - kpad, ix = x + t % ks - kpad;
        if ((unsigned)iy < (unsigned)Hs && (unsigned)ix < (unsigned)Ws) mask |= 1u << t;
      }
    }
    amask[j] = okm ? mask : 0u;
    roff0[j] = (unsigned)pix * (unsigned)(p.C0 * ESZ);
    roff1[j] = (unsigned)pix * (unsigned)(p.C1 * ESZ);
  }
  unsigned boff[BPASS > 0 ? BPASS : 1];
#pragma unroll
  for (int j = 0; j < BPASS; ++j) boff[j] = (unsigned)((tile_n * BN + lrow + RPASS * j) * ktot + lc * EPC) * ESZ;

  const int s_begin = (int)(((int64_t)p.ksteps * split) / p.nsplit);
  const int s_end = (int)(((int64_t)p.ksteps * (split + 1)) / p.nsplit);

  typedef __attribute__((address_space(3))) void* lptr_t;
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const char*>(p.in0) - (int64_t)bshift * p.C0 * ESZ), 0, p.rec_a, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const char*>(p.C1 ? p.in1 : p.in0) - (int64_t)bshift * p.C1 * ESZ), 0, p.rec_a,
      0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const char*>(p.w) + (GEOM == ADN_GEMM_T2 ? (int64_t)phase * p.N * ktot * ESZ : 0)), 0,
      p.rec_b, 0x00020000);

  // scalar K-loop state of the NEXT step to issue: tap index and channel offset inside the tap
  const int kpt = WIDE ? Cin / BK : 1;
  int is_tap = WIDE ? s_begin / kpt : 0;
  int is_c0 = WIDE ? (s_begin - is_tap * kpt) * BK : 0;

  // One K-step of LDS-DMA = APASS + BPASS wave-instructions, issued right behind the step's barrier.  (Measured and
  // rejected in round 2: spreading the pieces between the MFMA rows of the step with sched_barrier fences -- L3 forward
  // 55 -> 67 us, D4 forward 60 -> 80 us, the others within 3 %.)
  // 1 x 1 images: of the 16 (S2) / 4 (T2 phase) taps only 4 / 1 can ever be in range (the 2 x 2 input under the single
  // output pixel, resp. the one input pixel of the phase); the K loop walks those, the host counts the K-steps the same way
  const bool onepx = p.onepx != 0;
  auto real_tap = [&](int v) -> int {
    if (!onepx) return v;
    if constexpr (GEOM == ADN_GEMM_S2) return 5 + (v & 1) + 4 * (v >> 1);       // (ky, kx) in {1, 2} x {1, 2}
    else if constexpr (GEOM == ADN_GEMM_T2) return phase;                        // ty = ph, tx = pw
    else return v;
  };
  auto issue_step = [&](int s, int buf) {
    char* adst = smem + buf * STAGE_BYTES + wave * 1024;
    char* bdst = adst + BM * 128;
    int kelem = s * BK;                        // first weight-row element of the step
    if constexpr (WIDE) {
      // whole K-step inside one tap and one source: everything but the validity bit is scalar
      const int tap = real_tap(is_tap), c0 = is_c0;
      kelem = tap * Cin + c0;
      const bool second = c0 >= p.C0;
      const int Cs = second ? p.C1 : p.C0;
      const int coff = second ? c0 - p.C0 : c0;
      int shift;
      if constexpr (GEOM == ADN_GEMM_S2) shift = (tap >> 2) * Wl + (tap & 3);
      else if constexpr (GEOM == ADN_GEMM_T2) shift = (adn_t2_dy(ph, tap >> 1) + 1) * Ws + (adn_t2_dy(pw, tap & 1) + 1);
      else shift = (tap / ks) * Ws + tap % ks;
      const int soff = (shift * Cs + coff) * ESZ;
      const unsigned lane_c = (unsigned)(lc * EPC * ESZ);
      if (!(p.skip & 1))
#pragma unroll
      for (int j = 0; j < APASS; ++j) {
        const unsigned base = (second ? roff1[j] : roff0[j]) + lane_c;
        const unsigned voff = ((amask[j] >> tap) & 1u) ? base : OOB;
        if (second)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lptr_t)(adst + j * (RPASS * 128)), 16, voff, soff, 0, 0);
        else
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (lptr_t)(adst + j * (RPASS * 128)), 16, voff, soff, 0, 0);
      }
      is_c0 += BK;
      if (is_c0 == Cin) {
        is_c0 = 0;
        ++is_tap;
      }
    } else {
      // narrow layers (Cin = 8, 16, 32; single source): several taps inside one K-step, per-lane tap
      const int k0 = s * BK + lc * EPC;
      const int tap = k0 / Cin;
      const int c = k0 - tap * Cin;
      int shift;
      if constexpr (GEOM == ADN_GEMM_S2) shift = (tap >> 2) * Wl + (tap & 3);
      else if constexpr (GEOM == ADN_GEMM_T2) shift = (adn_t2_dy(ph, tap >> 1) + 1) * Ws + (adn_t2_dy(pw, tap & 1) + 1);
      else shift = (tap / ks) * Ws + tap % ks;
      const unsigned lane_off = (unsigned)((shift * Cin + c) * ESZ);
#pragma unroll
      for (int j = 0; j < APASS; ++j) {
        const unsigned voff = (tap < ntaps && ((amask[j] >> tap) & 1u)) ? roff0[j] + lane_off : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (lptr_t)(adst + j * (RPASS * 128)), 16, voff, 0, 0, 0);
      }
    }
    const int soffb = kelem * ESZ;
    if (!(p.skip & 2))
#pragma unroll
    for (int j = 0; j < BPASS; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lptr_t)(bdst + j * (RPASS * 128)), 16, boff[j], soffb, 0, 0);
  };

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // ---- BWD epilogue operands requested up front (bf16, unsplit) ----
  // The BWD epilogue reads up to three tensors per output element (activation sign reference, the running gradient it
  // accumulates into, the raw conv output for the BatchNorm-backward sums): issued behind the K loop they are a fully
  // exposed HBM round trip per tile (the two workgroups of a CU run in phase: L2 dgrad 50 us with a store-only epilogue,
  // 85 us with this one).  Their addresses do not depend on the GEMM result, so the 16-byte chunks of this thread's
  // epilogue rows are loaded NOW and ride in registers under the K loop (<= 96 VGPRs; the kernel runs 2 waves per SIMD).
  constexpr int CPR = BN / 8;        // 8-channel column groups per row
  constexpr int RSTEP = NTHR / CPR;  // rows covered per pass
  constexpr int RPT = BM / RSTEP;    // rows per thread
  const int e_cg = tid % CPR;
  const int e_rsub = tid / CPR;
  const int e_n0 = tile_n * BN + e_cg * 8;
  constexpr bool PRE_OK = PRE && sizeof(T) == 2;
  const bool pre_on = PRE_OK && p.epi == ADN_EPI_BWD && p.nsplit == 1;
  u32x4_t pre_r[RPT], pre_o[RPT], pre_z[RPT];
  if constexpr (PRE_OK) {
    if (pre_on) {
      const bool first = e_n0 < p.seg[0].channels;
      const AdnEpiSeg& sq = first ? p.seg[0] : p.seg[1];
      const int nl0 = first ? e_n0 : e_n0 - p.seg[0].channels;
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        const int m = tile_m * BM + e_rsub + RSTEP * k;
        pre_r[k] = pre_o[k] = pre_z[k] = u32x4_t{0u, 0u, 0u, 0u};
        if (m < p.Msmall) {
          int64_t op;
          if constexpr (GEOM != ADN_GEMM_T2) {
            op = m;
          } else {
            int b, i, jx;
            decode(m, b, i, jx);
            op = ((int64_t)b * Hl + 2 * i + ph) * Wl + 2 * jx + pw;
          }
          const int64_t idx = op * sq.channels + nl0;
          pre_r[k] = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const uint16_t*>(sq.ref) + idx);
          if (sq.accumulate) pre_o[k] = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const uint16_t*>(sq.out0) + idx);
          if (sq.partials) pre_z[k] = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const uint16_t*>(sq.z) + idx);
        }
      }
    }
  }

  // ---- pipeline: STAGES-1 steps of LDS-DMA in flight; one raw barrier per K-step ----
  // At the top of step s the stages s .. s+STAGES-2 have been issued.  "s_waitcnt vmcnt(LOADS*(STAGES-2))"
  // retires this wave's DMA of stage s (vmcnt counts in issue order), the barrier then (a) makes every
  // wave's part of stage s visible and (b) proves that all waves finished reading stage s-1, whose buffer
  // the DMA of stage s+STAGES-1 may now overwrite.
#pragma unroll
  for (int d = 0; d < STAGES - 1; ++d)
    if (s_begin + d < s_end) issue_step(s_begin + d, d);

  const int frow = lane & 15;
  const int fq = lane >> 4;
  int cur = 0;
  for (int s = s_begin; s < s_end; ++s) {
    if (STAGES == 3 && s + 1 < s_end) wait_vmcnt<LOADS>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int nb = cur + STAGES - 1;
    if (nb >= STAGES) nb -= STAGES;
    if (s + STAGES - 1 < s_end) issue_step(s + STAGES - 1, nb);
    const char* Ab = smem + cur * STAGE_BYTES;
    const char* Bb = Ab + BM * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4_t af[MT], bf[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int row = wm * 64 + i * 16 + frow;
        af[i] = *reinterpret_cast<const u32x4_t*>(Ab + row * 128 + (((ks * 4 + fq) ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int row = wn * WN + j * 16 + frow;
        bf[j] = *reinterpret_cast<const u32x4_t*>(Bb + row * 128 + (((ks * 4 + fq) ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) mma_tile<T>(af[i], bf[j], acc[i][j]);
    }
    cur = cur + 1 == STAGES ? 0 : cur + 1;
  }
  __syncthreads();   // all waves done with the staging buffers before the epilogue tile reuses them

  // ---- epilogue through LDS: ctile[BM][LDC] f32 ----
  float* ct = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        ct[(wm * 64 + i * 16 + 4 * fq + r) * LDC + wn * WN + j * 16 + frow] = acc[i][j][r];
  __syncthreads();

  const int cg = e_cg;
  const int rsub = e_rsub;
  const int n0 = e_n0;

  int epi = p.epi;
  AdnEpiSeg sg;
  int nl;
  int64_t mout_total = (GEOM == ADN_GEMM_T2) ? (int64_t)p.Msmall * 4 : (int64_t)p.Msmall;
  if (p.nsplit > 1) {
    epi = ADN_EPI_RAW;
    sg = p.seg[0];
    sg.out0 = p.slab + (int64_t)split * mout_total * p.N;
    sg.channels = p.N;
    sg.partials = nullptr;
    nl = n0;
  } else if (n0 < p.seg[0].channels) {
    sg = p.seg[0];
    nl = n0;
  } else {
    sg = p.seg[1];
    nl = n0 - p.seg[0].channels;
  }
  EpiCols cols;
  epi_cols_init<T>(epi, sg, nl, cols);
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;

#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    const int row = rsub + RSTEP * k;
    const int m = tile_m * BM + row;
    if (m < p.Msmall) {
      int64_t op;
      if constexpr (GEOM != ADN_GEMM_T2) {
        op = m;
      } else {
        int b, i, jx;
        decode(m, b, i, jx);
        op = ((int64_t)b * Hl + 2 * i + ph) * Wl + 2 * jx + pw;
      }
      float v[8];
      const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(ct + row * LDC + cg * 8);
      const f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(ct + row * LDC + cg * 8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = v0[e];
        v[4 + e] = v1[e];
      }
      if constexpr (PRE_OK) {
        if (pre_on) {
          epi_bwd_pre8(sg, cols, op, nl, v, pre_r[k], pre_o[k], pre_z[k], s1, s2);
          continue;
        }
      }
      epi_vec8<T>(epi, sg, cols, op, nl, v, s1, s2);
    }
  }

  // (workgroup-uniform condition: a 128-wide tile may straddle a segment with stats and one without)
  if (p.nsplit == 1 && (p.seg[0].partials != nullptr || p.seg[1].partials != nullptr) &&
      (p.epi == ADN_EPI_Z_STATS || p.epi == ADN_EPI_BWD)) {
    // reduce over the row subsets: lanes sharing cg inside a wave, then the NW waves through LDS.
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int o = CPR; o < 64; o <<= 1) {
        s1[e] += __shfl_xor(s1[e], o, 64);
        s2[e] += __shfl_xor(s2[e], o, 64);
      }
    }
    __syncthreads();  // everyone is done reading ctile
    float* red = reinterpret_cast<float*>(smem);  // [NW waves][2][BN]
    if (lane < CPR) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[(wave * 2 + 0) * BN + cg * 8 + e] = s1[e];
        red[(wave * 2 + 1) * BN + cg * 8 + e] = s2[e];
      }
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const int st = tid / BN, c = tid % BN;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += red[(w * 2 + st) * BN + c];
      const int n = tile_n * BN + c;
      const AdnEpiSeg& sq = (n < p.seg[0].channels) ? p.seg[0] : p.seg[1];
      const int ncl = (n < p.seg[0].channels) ? n : n - p.seg[0].channels;
      const int64_t P = (int64_t)phase * p.tiles_m + tile_m;
      if (sq.partials) sq.partials[(P * 2 + st) * sq.channels + ncl] = t;
    }
  }
#endif
}


// =====================================================================================================================
// Patch-staged variant (bf16, wide channels, unsplit): the gathered operand goes through LDS as an input PATCH.
//
// Why: timing-only builds of the kernel above (ADN_IGEMM_SKIP) showed that the K loop is paced by the LDS-DMA of the
// GATHERED operand: not issuing it at all gives -27 ... -34 % on the wide layers (L2 forward 48 -> 35 us, D1 forward
// 121 -> 80 us), not issuing the weights only -7 ... -15 %.  Staging taps as K-steps fetches every input pixel 4 times
// (overlapping 4 x 4 stride-2 windows, resp. 2 x 2 windows of a transposed-conv phase).  Here a workgroup owns an
// 8 x 16 block of output pixels of ONE image and stages, per 32-channel chunk, the input pixels that block needs ONCE:
//   S2 (k4 s2 p1 conv, dgrad of the transposed conv): an 18 x 34 patch, loaded in two halves by input-row parity (a half
//      serves the two kernel rows ky of that parity = 8 taps = 4 K-steps), even / odd input columns in separate planes
//      so that the 16 output columns of a fragment read 16 CONSECUTIVE 64-byte LDS pixels for every tap;
//   T2 (one phase of the transposed conv, dgrad of the conv): a 9 x 17 patch per chunk (4 taps = 2 K-steps);
//   S1 (3 x 3 stride-1 conv of the DoubleConv nets, forward and dgrad): a 10 x 18 patch per chunk, 9 taps = 3 K-steps of
//      one kernel row (3 taps, 48 MFMAs per wave) each: every input pixel is fetched once instead of 9 times.
// A K-step = 2 taps x 32 channels (the same 32 MFMAs per wave and barrier as above); the weights of the step come as
// [2 taps][BN rows][64 B].  LDS-DMA wave-instructions per wave and step: 4 (weights) + 1.25 (patch) instead of 8; gathered
// bytes per step 4.9 KiB instead of 16 KiB.  64-byte pixels: the 16-byte chunk index is XOR-ed with ((pixel >> 2) & 1) << 1
// (on the DMA source side), which makes the ds_read_b128 fragment reads conflict free at every alignment (brute-forced).
// The next patch (half) is loaded into the other patch buffer during the steps of the current one; every step ends in
// "s_waitcnt vmcnt(0); s_barrier", so it has landed before its first use.
// TALL (64 output columns only): a 16 x 16 pixel tile, the 4 waves stacked along the pixels (64 rows x all 64 columns
// each).  With the 2 x 2 wave grid a 64-column tile leaves every wave a 64 x 32 sub-tile: 12 ds_read_b128 per 16 MFMAs, and
// the two waves of a row pair read the same A fragments -- LDS bandwidth, not the matrix pipe, bounds the step (PMC:
// SQ_WAIT_INST_LDS 4.7x the 128-column variant).  64 x 64 per wave is 8 reads per 16 MFMAs with no shared fragments.
// PAIR (S2 / T2, 8 x 8 small-grid images -- the Hs = 8 level of unet_256): a tile is two whole images side by side (8 rows x
// [8 columns of image 2t | 8 columns of image 2t + 1]); every patch plane holds the two images' 9 columns back to back (18
// columns), so the only change for the fragment reads is +1 LDS pixel for the second image's output columns.
template <int BN, int GEOM, bool TALL = false, bool PRE = true, bool PAIR = false>
__global__ __launch_bounds__(256, 2) void igemm_patch_kernel(KParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef uint16_t T;
  constexpr int TH = TALL ? 16 : 8, TW = 16;
  constexpr int BM = TH * TW, NWN = TALL ? 1 : 2, NTHR = 256;
  constexpr int WN = BN / NWN, NT = WN / 16, MT = 4;
  constexpr bool S2 = GEOM == ADN_GEMM_S2, S1 = GEOM == ADN_GEMM_S1;
  constexpr int MW = (S1 || PAIR) ? TW + 2 : TW + 1;          // patch columns per plane (S1: 3 x 3 window, 18 columns; PAIR: 9 + 9)
  constexpr int SEG_PIX = S2 ? 2 * (TH + 1) * MW : (S1 ? (TH + 2) * MW : (TH + 1) * MW);   // 306 / 180 / 153 pixels
  constexpr int SEG_STEPS = S2 ? 4 : (S1 ? 3 : 2);            // K-steps served by one segment
  constexpr int TPS = S1 ? 3 : 2;                             // taps per K-step (S1: one kernel row)
  constexpr int PPIECES = (SEG_PIX + 15) / 16;                // 1-KiB DMA pieces per segment (20 / 12 / 10)
  constexpr int PK = (PPIECES + 3) / 4;                       // pieces per wave (5 / 3 / 3)
  constexpr int PBUF = PPIECES * 1024;
  constexpr int BPT = BN / 16;                                // weight pieces per tap
  constexpr int BK_ = TPS * BPT / 4;                          // weight pieces per wave and step (4 / 2, S1: 6 / 3)
  constexpr int BBUF = TPS * BN * 64;                         // one weight stage: [TPS taps][BN][64 B]
  constexpr int LDC = BN + 4;
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Pl = smem;                       // [2][PBUF]
  char* Bl = smem + 2 * PBUF;            // [2][BBUF]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = TALL ? wave : (wave >> 1), wn = TALL ? 0 : (wave & 1);
  constexpr int NPH = (S2 || S1) ? 1 : 4;
  const int nwg = p.tiles_m * p.tiles_n * NPH;
  const int wg0 = xcd_remap(blockIdx.x, nwg);
  const int phase = wg0 % NPH;
  const int wg = wg0 / NPH;
  const int tile_n = wg % p.tiles_n;
  const int tile_m = wg / p.tiles_n;
  const int ph = phase >> 1, pw = phase & 1;
  const int Hs = p.Hs, Ws = p.Ws, Hl = 2 * Hs, Wl = 2 * Ws;
  const int Cin = p.C0 + p.C1;
  // tile -> (image, first output row, first output column) on the small grid
  int tb, oy0, ox0;
  if constexpr (PAIR) {
    tb = 2 * tile_m;                       // first image of the pair; the tile is both images in full
    oy0 = ox0 = 0;
  } else {
    const int tpr = Ws / TW, tpi = (Hs / TH) * tpr;
    tb = tile_m / tpi;
    const int trem = tile_m - tb * tpi;
    oy0 = (trem / tpr) * TH;
    ox0 = (trem % tpr) * TW;
  }
  // gathered image: S2 gathers from the large grid (2Hs x 2Ws), T2 from the small grid
  const int Hg = S2 ? Hl : Hs, Wg = S2 ? Wl : Ws;        // (S1 and T2 gather from the small grid)
  const int ymin = (S2 || S1) ? 0 : (ph == 0 ? -1 : 0), xmin = (S2 || S1) ? 0 : (pw == 0 ? -1 : 0);

  // ---- patch loader geometry (per lane, constant over the K loop) ----
  // piece pi = wave + 4k covers LDS pixels q = 16 pi .. 16 pi + 15; lane l -> pixel 16 pi + (l >> 2), physical chunk l & 3
  unsigned ppix[PK];                         // gathered pixel index (+ bshift) of this lane's pixel of piece k
  unsigned pmask = 0;                        // bit 2k: valid for row parity 0 (T2: valid), bit 2k+1: row parity 1
  const int bshift = Wg + 1;                 // descriptor base shifted back: every offset >= 0
#pragma unroll
  for (int k = 0; k < PK; ++k) {
    const int q = 16 * (wave + 4 * k) + (lane >> 2);
    int iy, ix;
    bool ok0, ok1;
    int img = 0;                             // PAIR: which image of the pair this patch column belongs to
    if constexpr (S2) {
      const int par = q / ((TH + 1) * MW), rem = q - par * ((TH + 1) * MW);
      const int hr = rem / MW;
      int m = rem - hr * MW;
      if constexpr (PAIR) {
        img = m >= 9 ? 1 : 0;
        m -= 9 * img;
      }
      iy = 2 * oy0 - 1 + 2 * hr;             // row of parity 0; parity 1 is the next row
      ix = 2 * ox0 - 1 + 2 * m + par;
      const bool okx = (unsigned)ix < (unsigned)Wg && q < SEG_PIX;
      ok0 = okx && (unsigned)iy < (unsigned)Hg;
      ok1 = okx && (unsigned)(iy + 1) < (unsigned)Hg;
    } else {
      const int hr = q / MW;
      int m = q - hr * MW;
      if constexpr (PAIR) {
        img = m >= 9 ? 1 : 0;
        m -= 9 * img;
      }
      iy = oy0 + hr + (S1 ? -1 : ymin);
      ix = ox0 + m + (S1 ? -1 : xmin);
      ok0 = ok1 = (unsigned)ix < (unsigned)Wg && (unsigned)iy < (unsigned)Hg && q < SEG_PIX;
    }
    ppix[k] = (unsigned)(((tb + img) * Hg + iy) * Wg + ix + bshift);
    pmask |= (ok0 ? 1u : 0u) << (2 * k) | (ok1 ? 2u : 0u) << (2 * k);
  }
  // ---- weight loader geometry: piece pid = wave + 4k of a step's [2][BN][64 B] tile ----
  unsigned bvo[BK_];
  const int ktot = p.wstride;
#pragma unroll
  for (int k = 0; k < BK_; ++k) {
    const int pid = wave + 4 * k;
    const int tsel = pid / BPT, row = (pid % BPT) * 16 + (lane >> 2);
    const int lc = (lane & 3) ^ (((row >> 2) & 1) << 1);
    bvo[k] = (unsigned)(((tile_n * BN + row) * ktot + tsel * Cin + lc * 8) * 2);
  }
  typedef __attribute__((address_space(3))) void* lptr_t;
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const char*>(p.in0) - (int64_t)bshift * p.C0 * 2), 0, p.rec_a, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const char*>(p.C1 ? p.in1 : p.in0) - (int64_t)bshift * p.C1 * 2), 0, p.rec_a, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const char*>(p.w) + ((S2 || S1) ? 0 : (int64_t)phase * p.N * ktot * 2)), 0, p.rec_b,
      0x00020000);

  const int nchunks = Cin >> 5;
  const int nseg = S2 ? 2 * nchunks : nchunks;
  const int nsteps = nseg * SEG_STEPS;

  // patch pieces of segment sg whose index k matches the step-in-segment ss (k = 0 and SEG_STEPS with ss = 0, ...)
  auto issue_patch = [&](int sg, int ss) {
    const int c = S2 ? sg >> 1 : sg, par = S2 ? sg & 1 : 0;
    const int c0 = c << 5;
    const bool second = c0 >= p.C0;
    const int Cs = second ? p.C1 : p.C0;
    const int coff = second ? c0 - p.C0 : c0;
    const int soff = (par * Wg * Cs + coff) * 2;
    char* dst = Pl + (sg & 1) * PBUF + wave * 1024;
#pragma unroll
    for (int k = 0; k < PK; ++k) {
      if (k % SEG_STEPS != ss) continue;
      if (wave + 4 * k >= PPIECES) continue;
      const bool ok = (pmask >> (2 * k + par)) & 1u;
      // logical 16-byte chunk of this lane: physical chunk (lane & 3) ^ swizzle of its LDS pixel q = 16 (wave + 4k) + lane / 4
      // (the wave and k terms are multiples of 16: (q >> 2) & 1 = (lane >> 4) & 1)
      const unsigned lc16 = (unsigned)(((lane & 3) ^ (((lane >> 4) & 1) << 1)) << 4);
      const unsigned vo = ok ? ppix[k] * (unsigned)(Cs * 2) + lc16 : OOB;
      if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lptr_t)(dst + k * 4096), 16, vo, soff, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (lptr_t)(dst + k * 4096), 16, vo, soff, 0, 0);
    }
  };
  // weights of step s: taps (t0, t0 + 1), channels of the step's chunk
  auto issue_b = [&](int s) {
    const int sg = s / SEG_STEPS, ss = s - sg * SEG_STEPS;
    int c, t0;
    if constexpr (S2) {
      c = sg >> 1;
      const int ky = 2 * (ss >> 1) + (sg & 1);
      t0 = ky * 4 + 2 * (ss & 1);
    } else {
      c = sg;
      t0 = TPS * ss;
    }
    const int soff = (t0 * Cin + (c << 5)) * 2;
    char* dst = Bl + (s & 1) * BBUF + wave * 1024;
#pragma unroll
    for (int k = 0; k < BK_; ++k) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lptr_t)(dst + k * 4096), 16, bvo[k], soff, 0, 0);
  };

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // ---- BWD epilogue operands requested up front (see the kernel above) ----
  constexpr int CPR = BN / 8, RSTEP = NTHR / CPR, RPT = BM / RSTEP;
  const int e_cg = tid % CPR, e_rsub = tid / CPR;
  const int e_n0 = tile_n * BN + e_cg * 8;
  auto row_op = [&](int row) -> int64_t {       // output pixel index of tile row `row` (row = oyl * 16 + oxl)
    const int oy = oy0 + (row >> 4);
    const int ox = PAIR ? (row & 7) : ox0 + (row & 15);
    const int ib = PAIR ? tb + ((row >> 3) & 1) : tb;
    if constexpr (S2 || S1) return ((int64_t)ib * Hs + oy) * Ws + ox;
    else return ((int64_t)ib * Hl + 2 * oy + ph) * Wl + 2 * ox + pw;
  };
  // (S2 = dgrad of a transposed conv never accumulates in the U-Net: its running-gradient chunk is not kept in registers
  //  -- 32 VGPRs that made the 128-column variant spill -- but read in the epilogue if a caller asks for it)
  constexpr bool KEEP_OLD = !(S2 || S1);      // T2 (dgrad of the strided conv) is the accumulating one in the U-Net
  constexpr int NPO = KEEP_OLD ? RPT : 1;
  // (PRE = false: the TALL forward instantiation -- without the 8 x 2 prefetch registers it keeps 3 waves per SIMD)
  const bool pre_on = PRE && p.epi == ADN_EPI_BWD;
  u32x4_t pre_r[RPT], pre_o[NPO], pre_z[RPT];
  if (pre_on) {
    const bool first = e_n0 < p.seg[0].channels;
    const AdnEpiSeg& sq = first ? p.seg[0] : p.seg[1];
    const int nl0 = first ? e_n0 : e_n0 - p.seg[0].channels;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const int64_t idx = row_op(e_rsub + RSTEP * k) * sq.channels + nl0;
      pre_r[k] = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const uint16_t*>(sq.ref) + idx);
      pre_z[k] = u32x4_t{0u, 0u, 0u, 0u};
      if constexpr (KEEP_OLD) {
        pre_o[k] = u32x4_t{0u, 0u, 0u, 0u};
        if (sq.accumulate) pre_o[k] = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const uint16_t*>(sq.out0) + idx);
      }
      if (sq.partials) pre_z[k] = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const uint16_t*>(sq.z) + idx);
    }
  }

  // ---- prologue: the whole first patch segment + the weights of step 0 ----
#pragma unroll
  for (int ss = 0; ss < SEG_STEPS; ++ss) issue_patch(0, ss);
  issue_b(0);

  const int frow = lane & 15, fq = lane >> 4;
  // fragment pixel base of this lane's M-tile i: (oyl = wm*4 + i) * MW + frow
  const int qb0 = wm * 4 * MW + frow + (PAIR ? (frow >> 3) : 0);   // (PAIR: the second image's columns sit one LDS pixel further)

  int s = 0;
  for (int sg = 0; sg < nseg; ++sg) {
    const char* Pb = Pl + (sg & 1) * PBUF;
#pragma unroll
    for (int ss = 0; ss < SEG_STEPS; ++ss, ++s) {
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (s + 1 < nsteps) issue_b(s + 1);
      if (sg + 1 < nseg) issue_patch(sg + 1, ss);
      const char* Bb = Bl + (s & 1) * BBUF;
#pragma unroll
      for (int e = 0; e < TPS; ++e) {               // the step's taps, 32 channels each
        int qoff;
        if constexpr (S2) qoff = ((e * (TH + 1)) + (ss >> 1)) * MW + (ss & 1);     // kx = 2 (ss & 1) + e, ky >> 1 = ss >> 1
        else if constexpr (S1) qoff = ss * MW + e;                                  // (ky, kx) = (ss, e)
        else qoff = (adn_t2_dy(ph, ss) - ymin) * MW + (adn_t2_dy(pw, e) - xmin);
        u32x4_t af[MT], bf[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int q = qb0 + i * MW + qoff;
          af[i] = *reinterpret_cast<const u32x4_t*>(Pb + q * 64 + ((fq ^ (((q >> 2) & 1) << 1)) << 4));
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int row = wn * WN + j * 16 + frow;
          bf[j] = *reinterpret_cast<const u32x4_t*>(Bb + e * (BN * 64) + row * 64 + ((fq ^ (((row >> 2) & 1) << 1)) << 4));
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) mma_tile<T>(af[i], bf[j], acc[i][j]);
      }
    }
  }
  __syncthreads();

  // ---- epilogue through LDS (as above; tile row = oyl * 16 + oxl) ----
  float* ct = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        ct[(wm * 64 + i * 16 + 4 * fq + r) * LDC + wn * WN + j * 16 + frow] = acc[i][j][r];
  __syncthreads();
  const int cg = e_cg, rsub = e_rsub, n0 = e_n0;
  const int epi = p.epi;
  AdnEpiSeg sg2;
  int nl;
  if (n0 < p.seg[0].channels) {
    sg2 = p.seg[0];
    nl = n0;
  } else {
    sg2 = p.seg[1];
    nl = n0 - p.seg[0].channels;
  }
  EpiCols cols;
  epi_cols_init<T>(epi, sg2, nl, cols);
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    const int row = rsub + RSTEP * k;
    const int64_t op = row_op(row);
    float v[8];
    const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(ct + row * LDC + cg * 8);
    const f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(ct + row * LDC + cg * 8 + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = v0[e];
      v[4 + e] = v1[e];
    }
    if (pre_on) {
      if constexpr (!KEEP_OLD) {
        u32x4_t old = {0u, 0u, 0u, 0u};
        if (sg2.accumulate) old = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const uint16_t*>(sg2.out0) + op * sg2.channels + nl);
        epi_bwd_pre8(sg2, cols, op, nl, v, pre_r[k], old, pre_z[k], s1, s2);
      } else {
        epi_bwd_pre8(sg2, cols, op, nl, v, pre_r[k], pre_o[k], pre_z[k], s1, s2);
      }
    } else {
      epi_vec8<T>(epi, sg2, cols, op, nl, v, s1, s2);
    }
  }
  if ((p.seg[0].partials != nullptr || p.seg[1].partials != nullptr) && (p.epi == ADN_EPI_Z_STATS || p.epi == ADN_EPI_BWD)) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int o = CPR; o < 64; o <<= 1) {
        s1[e] += __shfl_xor(s1[e], o, 64);
        s2[e] += __shfl_xor(s2[e], o, 64);
      }
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);  // [4 waves][2][BN]
    if (lane < CPR) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[(wave * 2 + 0) * BN + cg * 8 + e] = s1[e];
        red[(wave * 2 + 1) * BN + cg * 8 + e] = s2[e];
      }
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const int st = tid / BN, c = tid % BN;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) t += red[(w * 2 + st) * BN + c];
      const int n = tile_n * BN + c;
      const AdnEpiSeg& sq = (n < p.seg[0].channels) ? p.seg[0] : p.seg[1];
      const int ncl = (n < p.seg[0].channels) ? n : n - p.seg[0].channels;
      const int64_t P = (int64_t)phase * p.tiles_m + tile_m;
      if (sq.partials) sq.partials[(P * 2 + st) * sq.channels + ncl] = t;
    }
  }
#endif
}

#include "igemm_ring.h"

// ---- generic direct path: any channel counts, one thread per output element, f32 slab out ----
template <typename T, int GEOM>
__global__ __launch_bounds__(256) void igemm_direct_kernel(KParams p) {
  const int Hs = p.Hs, Ws = p.Ws, Hl = 2 * Hs, Wl = 2 * Ws;
  const int Cin = p.C0 + p.C1;
  const int64_t mout = (GEOM == ADN_GEMM_T2) ? (int64_t)p.Msmall * 4 : (int64_t)p.Msmall;
  const int64_t total = mout * p.N;
  const T* in0 = reinterpret_cast<const T*>(p.in0);
  const T* in1 = reinterpret_cast<const T*>(p.in1);
  const T* w = reinterpret_cast<const T*>(p.w);
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t op = e / p.N;
    const int n = (int)(e - op * p.N);
    float acc = 0.f;
    if constexpr (GEOM == ADN_GEMM_S2) {
      const int b = (int)(op / (Hs * Ws));
      const int rem = (int)(op - (int64_t)b * (Hs * Ws));
      const int oy = rem / Ws, ox = rem - oy * Ws;
      for (int tap = 0; tap < 16; ++tap) {
        const int iy = 2 * oy - 1 + (tap >> 2), ix = 2 * ox - 1 + (tap & 3);
        if ((unsigned)iy >= (unsigned)Hl || (unsigned)ix >= (unsigned)Wl) continue;
        const int64_t pix = ((int64_t)b * Hl + iy) * Wl + ix;
        const T* wr = w + (int64_t)n * p.wstride + tap * Cin;
        for (int c = 0; c < p.C0; ++c) acc += ElemTraits<T>::load(in0 + pix * p.C0 + c) * ElemTraits<T>::load(wr + c);
        for (int c = 0; c < p.C1; ++c)
          acc += ElemTraits<T>::load(in1 + pix * p.C1 + c) * ElemTraits<T>::load(wr + p.C0 + c);
      }
    } else if constexpr (GEOM == ADN_GEMM_S1) {
      const int b = (int)(op / (Hs * Ws));
      const int rem = (int)(op - (int64_t)b * (Hs * Ws));
      const int oy = rem / Ws, ox = rem - oy * Ws;
      const int ks = p.ks, kpad = p.ks >> 1;
      for (int tap = 0; tap < ks * ks; ++tap) {
        const int iy = oy + tap / ks - kpad, ix = ox + tap % ks - kpad;
        if ((unsigned)iy >= (unsigned)Hs || (unsigned)ix >= (unsigned)Ws) continue;
        const int64_t pix = ((int64_t)b * Hs + iy) * Ws + ix;
        const T* wr = w + (int64_t)n * p.wstride + tap * Cin;
        for (int c = 0; c < p.C0; ++c) acc += ElemTraits<T>::load(in0 + pix * p.C0 + c) * ElemTraits<T>::load(wr + c);
        for (int c = 0; c < p.C1; ++c)
          acc += ElemTraits<T>::load(in1 + pix * p.C1 + c) * ElemTraits<T>::load(wr + p.C0 + c);
      }
    } else {
      const int b = (int)(op / ((int64_t)Hl * Wl));
      const int rem = (int)(op - (int64_t)b * Hl * Wl);
      const int oy = rem / Wl, ox = rem - oy * Wl;
      const int ph = oy & 1, pw = ox & 1, i = oy >> 1, jx = ox >> 1;
      const int phase = ph * 2 + pw;
      for (int t = 0; t < 4; ++t) {
        const int iy = i + adn_t2_dy(ph, t >> 1), ix = jx + adn_t2_dy(pw, t & 1);
        if ((unsigned)iy >= (unsigned)Hs || (unsigned)ix >= (unsigned)Ws) continue;
        const int64_t pix = ((int64_t)b * Hs + iy) * Ws + ix;
        const T* wr = w + ((int64_t)phase * p.N + n) * p.wstride + t * Cin;
        for (int c = 0; c < p.C0; ++c) acc += ElemTraits<T>::load(in0 + pix * p.C0 + c) * ElemTraits<T>::load(wr + c);
        for (int c = 0; c < p.C1; ++c)
          acc += ElemTraits<T>::load(in1 + pix * p.C1 + c) * ElemTraits<T>::load(wr + p.C0 + c);
      }
    }
    p.slab[e] = acc;
  }
}

// ---- slab reduce + epilogue: v = sum_s slab[s][op*N+n] ----
// Column mapping: thread = channel, block = rb consecutive output pixels x 256 channels; the per-channel
// stats of a block need no cross-thread reduction and the partial row index is the row-block index.
// rb is chosen on the host so that small-M (split-K) layers still fill the chip.
template <typename T>
__global__ __launch_bounds__(256) void igemm_reduce_kernel(KParams p, int64_t mout, int rb) {
  const int64_t row0 = (int64_t)blockIdx.x * rb;
  const int64_t slab_stride = mout * p.N;
  const int n = blockIdx.y * 256 + threadIdx.x;
  if (n >= p.N) return;
  const bool in0 = n < p.seg[0].channels;
  const AdnEpiSeg& sg = in0 ? p.seg[0] : p.seg[1];
  const int nl = in0 ? n : n - p.seg[0].channels;
  float s1 = 0.f, s2 = 0.f;
  for (int r = 0; r < rb; ++r) {
    const int64_t op = row0 + r;
    if (op >= mout) break;
    const float* src = p.slab + op * p.N + n;
    float v = 0.f;
#pragma unroll 8
    for (int s = 0; s < p.nsplit; ++s) v += src[s * slab_stride];
    epi_scalar<T>(p.epi, sg, op, nl, v, s1, s2);
  }
  if (sg.partials && (p.epi == ADN_EPI_Z_STATS || p.epi == ADN_EPI_BWD)) {
    sg.partials[((int64_t)blockIdx.x * 2 + 0) * sg.channels + nl] = s1;
    sg.partials[((int64_t)blockIdx.x * 2 + 1) * sg.channels + nl] = s2;
  }
}

inline int reduce_rows(int64_t mout, int N) {
  const int64_t cb = adn_cdiv(N, 256);
  int64_t rb = mout * cb / 1024;
  if (rb > 64) rb = 64;
  if (rb < 1) rb = 1;
  return (int)rb;
}

struct Plan {
  bool mfma;
  bool wide;
  bool patch;       // patch-staged kernel (bf16, wide, unsplit, image 8 x 16 tileable)
  bool tall;        // its 16 x 16-pixel, 4 x 1-wave form (64 output columns, T2 / S1, image 16 x 16 tileable)
  bool pair;        // its two-images-per-tile form (S2 / T2, 8 x 8 small-grid images, 128 output columns, unsplit)
  bool ring;        // ring-fed persistent kernel (igemm_ring.h): 16 x 16-pixel x bn tiles, one 8-wave workgroup per CU
  int wstride;
  int rb;
  int bm;
  int bn;
  int nsplit;
  int tiles_m, tiles_n, phases;
  int kpt, ksteps;
  bool onepx;
  int64_t mout;
  int64_t partial_rows;
  int64_t slab_bytes;
};

bool make_plan(const AdnIgemmDesc* d, Plan* pl) {
  const int esz = d->dtype == ADN_BF16 ? 2 : 4;
  const int bk = 128 / esz;
  const int Cin = d->C0 + d->C1;
  const int64_t msmall = (int64_t)d->B * d->Hs * d->Ws;
  pl->phases = d->geom == ADN_GEMM_T2 ? 4 : 1;
  pl->mout = msmall * pl->phases;
  const int taps = d->geom == ADN_GEMM_S2 ? 16 : (d->geom == ADN_GEMM_T2 ? 4 : d->ks * d->ks);
  // S1 operands (adn_pack_rows / adn_pack_transpose_taps) pad their rows to the K-step; the k4 packs do not
  pl->wstride = d->geom == ADN_GEMM_S1 ? (int)(adn_cdiv((int64_t)taps * Cin, bk) * bk) : taps * Cin;
  const int epc = 16 / esz;
  // wide: every K-step lies inside one tap of one source; narrow: single source, several taps per K-step
  pl->wide = (d->C0 % bk == 0) && (d->C1 % bk == 0);
  // (S1 weights are packed with rows zero-padded to the K-step, so taps*Cin need not divide; S2/T2 packs are not)
  const bool narrow_ok = (d->C1 == 0) && (d->C0 % epc == 0) && (d->geom == ADN_GEMM_S1 || (taps * Cin) % bk == 0);
  const bool aligned = (pl->wide || narrow_ok) && (d->N % 64 == 0) && (d->seg[0].channels % 64 == 0) &&
                       (d->seg[1].channels % 64 == 0);
  pl->mfma = aligned;
  pl->patch = pl->tall = pl->pair = pl->ring = false;
  pl->onepx = false;
  if (!aligned) {
    pl->bn = 0;
    pl->nsplit = 1;
    pl->tiles_m = pl->tiles_n = 0;
    pl->kpt = pl->ksteps = 0;
    pl->rb = reduce_rows(pl->mout, d->N);
    pl->partial_rows = adn_cdiv(pl->mout, pl->rb);
    pl->slab_bytes = pl->mout * d->N * 4;
    return true;
  }
  pl->bn = (d->N % 128 == 0) ? 128 : 64;     // segments are multiples of 64: an 8-channel group never straddles
  // 256-row tiles (8 waves, 3-stage ring) when they still give every CU a workgroup; else 128-row tiles
  // (only worth it for long K loops: with K <= 1024 two 128-row workgroups per CU overlap each other's
  //  prologue/epilogue better: measured 685 vs 621 TFLOP/s on L1 forward)
  const bool fills256 = adn_cdiv(msmall, 256) * (d->N / pl->bn) * pl->phases >= 256;
  pl->bm = (fills256 && (pl->bn == 64 || taps * Cin / bk >= 32)) ? 256 : 128;
  const Tune& tn = tune();
  if (tn.bm) pl->bm = tn.bm == 256 ? 256 : 128;
  if (tn.bn) pl->bn = (tn.bn == 128 && d->N % 128 == 0) ? 128 : 64;
  // Transposed-conv geometry (forward of the up path, input gradients of the down path): 64-column tiles throughout --
  // the tall 256 x 64 patch form on the 16 x 16 ... 64 x 64 image levels (less patch halo per pixel, half the weight bytes
  // per K-step), twice the workgroups on the split-K levels.  Headline step 2.909 -> 2.860 ms and 2.932 -> 2.894 ms on two
  // boxes (ADN_IGEMM_BN_T2=128 restores 128-column tiles for an A/B; the two-image PAIR form keeps its 128 columns).
  if (d->geom == ADN_GEMM_T2 && d->dtype == ADN_BF16 && !tn.bn) pl->bn = (tn.bn_t2 == 128 && d->N % 128 == 0) ? 128 : 64;
  // exactly one 128 x 128 workgroup per CU (L3 forward of unet_256: 64 x 4 tiles) leaves every CU a single K loop with
  // nothing to overlap its LDS-DMA round trips; 64-column tiles give each CU two: 52.2 -> 47.9 us (S2 only: measured there)
  if (d->geom == ADN_GEMM_S2 && d->dtype == ADN_BF16 && pl->bn == 128 && !tn.bn) {
    const int64_t t128 = adn_cdiv(msmall, 128) * (d->N / 128);
    if (t128 >= 256 && t128 < 512) pl->bn = 64;
  }
  // 3 x 3 stride-1 convs (DoubleConv nets) with >= 128 output columns: the tall 256 x 64 tile (4 x 1 waves, the same 64 x 64
  // per wave) has less patch halo per pixel (18 x 18 for 256 pixels instead of 10 x 18 for 128) and half the weight bytes
  // per K-step: forward convs +5 ... +11 % (up3 conv1 297 -> 274 us, 1 131 TFLOP/s), input gradients 0 ... +8 %; and at the
  // 16 x 16-image level, where 128-column tiles give one workgroup per CU, 64-column tiles (not tall there) +20 ... +23 %.
  // RGBDepthNet 256^2 step 13.37 -> 13.11 ms with this rule forced everywhere (ADN_IGEMM_BN=64), so it is the rule now.
  if (d->geom == ADN_GEMM_S1 && d->ks == 3 && d->dtype == ADN_BF16 && pl->bn == 128 && !tn.bn && pl->wide &&
      d->Hs % 8 == 0 && d->Ws % 16 == 0) {
    const int64_t t128 = adn_cdiv(msmall, 128) * (d->N / 128);
    const bool tall_ok = d->Hs % 16 == 0 && tn.tall != 0 && msmall / 256 * (d->N / 64) >= 512;
    if (tall_ok || (t128 >= 256 && t128 < 512)) pl->bn = 64;
  }
  pl->tiles_m = (int)adn_cdiv(msmall, pl->bm);
  pl->tiles_n = d->N / pl->bn;
  pl->kpt = 0;
  pl->ksteps = pl->wstride / bk;
  // innermost U-Net level (1 x 1 small-grid images): 12 of 16 (S2) / 3 of 4 (T2) taps are padding for EVERY row
  pl->onepx = d->dtype == ADN_BF16 && pl->wide && d->Hs == 1 && d->Ws == 1 && d->geom != ADN_GEMM_S1 && tn.onepx != 0;
  if (pl->onepx) pl->ksteps = (d->geom == ADN_GEMM_S2 ? 4 : 1) * (Cin / bk);
  const int64_t tiles = (int64_t)pl->tiles_m * pl->tiles_n * pl->phases;
  int ns = 1;
  if (tiles < 256) {
    ns = (int)(512 / tiles);          // stay within one resident wave of workgroups (256 CUs x 2)
    const int max_by_k = pl->ksteps / 2 > 0 ? pl->ksteps / 2 : 1;
    if (ns > max_by_k) ns = max_by_k;
    // more than 16 slabs cost more in slab traffic (ns x M x N x 8 bytes written + read back) than the extra
    // workgroups return: L5 / L6 forward 23.6 / 20.8 -> 20.3 / 16.1 us, D6 dgrad 25.7 -> 19.4 us at a cap of 16
    // (bf16 only: in f32 the cap is neutral for speed, and the f32 reference fixtures of the Base+Residual net hold
    //  gradients through 4 x 4 BatchNorm layers that are sensitive to the summation order at their 5e-3 bound)
    // ... unless the layer is so small that more slabs still fit in `tinycap` MB (ADN_IGEMM_TINYCAP, default 4: the
    // 32-row GEMMs of the innermost level run 64 splits = 2 K-steps per workgroup, 17.8 -> 15.4 us incl. the reduce)
    int cap = esz == 2 ? 16 : 64;
    if (esz == 2 && tn.tinycap > 0) {
      const int64_t by_bytes = ((int64_t)tn.tinycap << 20) / (pl->mout * d->N * 4);
      if (by_bytes > cap) cap = by_bytes > 64 ? 64 : (int)by_bytes;
    }
    if (ns > cap) ns = cap;
    if (ns < 1) ns = 1;
  }
  if (tn.ns >= 1 && ns > tn.ns) ns = tn.ns;                  // tuning knob: cap on the split count
  pl->nsplit = ns;
  // patch-staged variant: bf16, every chunk of 32 channels inside one source, images tileable by 8 x 16 output pixels,
  // enough tiles that no split-K is wanted (ADN_IGEMM_PATCH=0 switches it off)
  pl->patch = d->dtype == ADN_BF16 && pl->wide && ns == 1 && (d->geom != ADN_GEMM_S1 || d->ks == 3) && d->Hs % 8 == 0 &&
              d->Ws % 16 == 0 && tn.patch != 0;
  // two 8 x 8 images per tile: replaces the split-K launch + reduce of that level by one unsplit patch launch
  // (T2 only: the S2 form -- D4 dgrad, 128 workgroups x 128 K-steps -- measured 99 us against 40 us for split-K + reduce)
  pl->pair = d->dtype == ADN_BF16 && pl->wide && d->geom == ADN_GEMM_T2 && d->Hs == 8 && d->Ws == 8 && d->B % 2 == 0 &&
             d->N % 128 == 0 && tn.pair != 0 && tn.patch != 0 &&
             msmall / 128 * (d->N / 128) * pl->phases >= 128;      // (64 workgroups with a 128-step K loop lose to split-K: L4 forward)
  if (pl->pair) {
    pl->patch = true;
    pl->bn = 128;
    pl->tiles_n = d->N / 128;
    ns = 1;
    pl->nsplit = 1;
  }
  pl->tall = !pl->pair && pl->patch && pl->bn == 64 && d->geom != ADN_GEMM_S2 && d->Hs % 16 == 0 && tn.tall != 0 &&
             msmall / 256 * pl->tiles_n * pl->phases >= 512;
  if (pl->patch) {
    pl->bm = pl->tall ? 256 : 128;
    pl->tiles_m = (int)(msmall / pl->bm);
  }
  // ring-fed persistent kernel: bf16, wide, unsplit, 16 x 16-pixel tiles, Z_STATS / BWD epilogues, a multiple of 8 K-steps
  // per tile (S2: always; T2: Cin % 128 == 0), every 32-channel chunk inside one source.  (The plan -- and with it the number
  // of partial rows -- depends on the epilogue: adn_igemm_num_partials must be asked with the epilogue of the launch.)
  pl->ring = false;
  if (d->dtype == ADN_BF16 && pl->wide && ns == 1 && !pl->pair && (d->geom == ADN_GEMM_S2 || d->geom == ADN_GEMM_T2) &&
      d->Hs % 16 == 0 && d->Ws % 16 == 0 && (d->N % 128 == 0 || (d->N == 64 && tn.ring64n != 0)) &&
      (d->geom == ADN_GEMM_S2 || Cin % 128 == 0) && tn.ring != 0 && pl->mout * d->N < (1ll << 31) &&
      (d->epi == ADN_EPI_Z_STATS || d->epi == ADN_EPI_BWD) && (tn.ring_epi == 0 || tn.ring_epi == d->epi) &&
      (tn.ring_geom < 0 || tn.ring_geom == d->geom) && (tn.ring_hs == 0 || tn.ring_hs == d->Hs)) {
    const int64_t t128 = msmall / 256 * (d->N / 128) * pl->phases;
    const int rbn = (d->N == 64 || tn.ring == 64 || (tn.ring != 128 && t128 < 192)) ? 64 : 128;    // too few 128-column tiles to fill the chip: 64
    pl->ring = true;
    pl->patch = pl->tall = false;
    pl->bm = 256;
    pl->bn = rbn;
    pl->tiles_m = (int)(msmall / 256);
    pl->tiles_n = d->N / rbn;
  }
  pl->rb = reduce_rows(pl->mout, d->N);
  if (ns > 1) {
    pl->partial_rows = adn_cdiv(pl->mout, pl->rb);
    pl->slab_bytes = (int64_t)ns * pl->mout * d->N * 4;
  } else {
    pl->partial_rows = (int64_t)pl->tiles_m * pl->phases;
    pl->slab_bytes = 0;
  }
  return true;
}

template <typename T, int BM_, int BN, int NWN, int GEOM, bool WIDE>
int launch_mfma(const KParams& kp, const Plan& pl, hipStream_t st) {
  const int stage = ((BM_ == 256 && NWN == 2) ? 3 : 2) * (BM_ + BN) * 128;
  const int epil = BM_ * (BN + 4) * 4;
  const int lds = stage > epil ? stage : epil;
  dim3 grid(pl.tiles_m * pl.tiles_n * pl.phases, 1, pl.nsplit);
  ADN_SET_LDS_ONCE(lds, &igemm_mfma_kernel<T, BM_, BN, NWN, GEOM, WIDE>);
  hipLaunchKernelGGL((igemm_mfma_kernel<T, BM_, BN, NWN, GEOM, WIDE>), grid, dim3(BM_ * NWN), lds, st, kp);
  return 0;
}

template <typename T, int GEOM, bool WIDE>
void dispatch_mfma2(const KParams& kp, const Plan& pl, hipStream_t st) {
  if (pl.bm == 256) {
    if (pl.bn == 128) launch_mfma<T, 256, 128, 2, GEOM, WIDE>(kp, pl, st);
    else launch_mfma<T, 256, 64, 1, GEOM, WIDE>(kp, pl, st);
  } else {
    if (pl.bn == 128) launch_mfma<T, 128, 128, 2, GEOM, WIDE>(kp, pl, st);
    else launch_mfma<T, 128, 64, 2, GEOM, WIDE>(kp, pl, st);
  }
}
template <typename T, int GEOM>
void dispatch_mfma(const KParams& kp, const Plan& pl, hipStream_t st) {
  if (pl.wide) dispatch_mfma2<T, GEOM, true>(kp, pl, st);
  else dispatch_mfma2<T, GEOM, false>(kp, pl, st);
}

template <int BN, int GEOM, bool TALL = false, bool PRE = true, bool PAIR = false>
void launch_patch1(const KParams& kp, const Plan& pl, hipStream_t st) {
  constexpr bool S2 = GEOM == ADN_GEMM_S2, S1 = GEOM == ADN_GEMM_S1;
  constexpr int TH = TALL ? 16 : 8;
  constexpr int MWc = PAIR ? 18 : 17;
  constexpr int ppieces = ((S2 ? 2 * (TH + 1) * MWc : (S1 ? (TH + 2) * 18 : (TH + 1) * MWc)) + 15) / 16;
  constexpr int stage = 2 * ppieces * 1024 + 2 * (S1 ? 3 : 2) * BN * 64;
  constexpr int epil = TH * 16 * (BN + 4) * 4;
  constexpr int lds = stage > epil ? stage : epil;
  ADN_SET_LDS_ONCE(lds, &igemm_patch_kernel<BN, GEOM, TALL, PRE, PAIR>);
  dim3 grid(pl.tiles_m * pl.tiles_n * pl.phases, 1, 1);
  hipLaunchKernelGGL((igemm_patch_kernel<BN, GEOM, TALL, PRE, PAIR>), grid, dim3(256), lds, st, kp);
}
inline void launch_patch(const KParams& kp, const Plan& pl, int geom, hipStream_t st) {
  if (pl.pair) {                 // 8 x 8 images, two per tile (every unet_256 layer of that level has >= 128 output columns)
    launch_patch1<128, ADN_GEMM_T2, false, true, true>(kp, pl, st);
  } else if (geom == ADN_GEMM_S2) {
    if (pl.bn == 128 && kp.epi == ADN_EPI_BWD) launch_patch1<128, ADN_GEMM_S2>(kp, pl, st);
    else if (pl.bn == 128) launch_patch1<128, ADN_GEMM_S2, false, false>(kp, pl, st);     // forward: no epilogue prefetch registers
    else launch_patch1<64, ADN_GEMM_S2>(kp, pl, st);
  } else if (geom == ADN_GEMM_T2) {
    if (pl.bn == 128) launch_patch1<128, ADN_GEMM_T2>(kp, pl, st);
    else if (pl.tall && kp.epi == ADN_EPI_BWD) launch_patch1<64, ADN_GEMM_T2, true, true>(kp, pl, st);
    else if (pl.tall) launch_patch1<64, ADN_GEMM_T2, true, false>(kp, pl, st);
    else launch_patch1<64, ADN_GEMM_T2>(kp, pl, st);
  } else {
    if (pl.bn == 128) launch_patch1<128, ADN_GEMM_S1>(kp, pl, st);
    else if (pl.tall && kp.epi == ADN_EPI_BWD) launch_patch1<64, ADN_GEMM_S1, true, true>(kp, pl, st);
    else if (pl.tall) launch_patch1<64, ADN_GEMM_S1, true, false>(kp, pl, st);
    else launch_patch1<64, ADN_GEMM_S1>(kp, pl, st);
  }
}

inline int device_cus() {
  static int n = 0;
  static std::once_flag once;
  std::call_once(once, [] {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) n = pr.multiProcessorCount;
    if (n <= 0) n = 256;
  });
  return n;
}

template <int GEOM, int BN, int SCHED>
void launch_ring1(const KParams& kp, const Plan& pl, hipStream_t st) {
  constexpr bool S2 = GEOM == ADN_GEMM_S2;
  constexpr int lds = (S2 ? 2 * 40 : 4 * 20) * 1024 + 4 * 2 * BN * 64 + 8 * BN * 4;
  ADN_SET_LDS_ONCE(lds, &igemm_ring_kernel<GEOM, BN, SCHED>);
  const int ntiles = pl.tiles_m * pl.tiles_n * pl.phases;
  const int cus = device_cus();
  const int nwgs = ntiles < cus ? ntiles : cus;
  auto magic = [](int d) -> unsigned { return (unsigned)((1ull << 32) / (unsigned)d + 1ull); };     // x / d = mulhi(x, magic), x * d < 2^32
  const int tpr = kp.Ws >> 4, tpi = (kp.Hs >> 4) * tpr;
  hipLaunchKernelGGL((igemm_ring_kernel<GEOM, BN, SCHED>), dim3(nwgs), dim3(512), lds, st, kp, ntiles, nwgs,
                     pl.tiles_n == 1 ? 0u : magic(pl.tiles_n), tpi == 1 ? 0u : magic(tpi), tpr == 1 ? 0u : magic(tpr));
}
template <int GEOM, int BN>
void launch_ring2(const KParams& kp, const Plan& pl, hipStream_t st) {
  if (tune().ring_sched) launch_ring1<GEOM, BN, 1>(kp, pl, st);
  else launch_ring1<GEOM, BN, 0>(kp, pl, st);
}
inline void launch_ring(const KParams& kp, const Plan& pl, int geom, hipStream_t st) {
  if (geom == ADN_GEMM_S2) {
    if (pl.bn == 128) launch_ring2<ADN_GEMM_S2, 128>(kp, pl, st);
    else launch_ring2<ADN_GEMM_S2, 64>(kp, pl, st);
  } else {
    if (pl.bn == 128) launch_ring2<ADN_GEMM_T2, 128>(kp, pl, st);
    else launch_ring2<ADN_GEMM_T2, 64>(kp, pl, st);
  }
}

template <typename T>
int run(const AdnIgemmDesc* d, const Plan& pl, hipStream_t st) {
  KParams kp;
  kp.in0 = d->in0;
  kp.in1 = d->in1;
  kp.w = d->w;
  kp.B = d->B;
  kp.Hs = d->Hs;
  kp.Ws = d->Ws;
  kp.C0 = d->C0;
  kp.C1 = d->C1;
  kp.N = d->N;
  kp.ks = d->geom == ADN_GEMM_S1 ? d->ks : 0;
  kp.wstride = pl.wstride;
  kp.Msmall = d->B * d->Hs * d->Ws;
  kp.kpt = pl.kpt;
  kp.ksteps = pl.ksteps;
  kp.onepx = pl.onepx ? 1 : 0;
  kp.nsplit = pl.nsplit;
  kp.tiles_m = pl.tiles_m;
  kp.tiles_n = pl.tiles_n;
  kp.epi = d->epi;
  kp.seg[0] = d->seg[0];
  kp.seg[1] = d->seg[1];
  kp.slab = reinterpret_cast<float*>(d->workspace);
  {
    auto lg2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; };
    kp.lgW = lg2(d->Ws);
    kp.lgH = lg2(d->Hs);
    if (kp.lgW < 0 || kp.lgH < 0) kp.lgW = kp.lgH = -1;
  }
  kp.rec_a = tune().noa ? 0u : 0x7ffffff0u;
  kp.rec_b = tune().nob ? 0u : 0x7ffffff0u;
  kp.skip = tune().skip;
  if (pl.mfma && pl.ring) {
    if constexpr (sizeof(T) == 2) launch_ring(kp, pl, d->geom, st);
    ADN_CHECK_LAUNCH();
  } else if (pl.mfma && pl.patch) {
    if constexpr (sizeof(T) == 2) launch_patch(kp, pl, d->geom, st);
    ADN_CHECK_LAUNCH();
  } else if (pl.mfma) {
    if (d->geom == ADN_GEMM_S2) dispatch_mfma<T, ADN_GEMM_S2>(kp, pl, st);
    else if (d->geom == ADN_GEMM_T2) dispatch_mfma<T, ADN_GEMM_T2>(kp, pl, st);
    else dispatch_mfma<T, ADN_GEMM_S1>(kp, pl, st);
    ADN_CHECK_LAUNCH();
    if (pl.nsplit > 1) {
      hipLaunchKernelGGL((igemm_reduce_kernel<T>),
                         dim3((unsigned)adn_cdiv(pl.mout, pl.rb), (unsigned)adn_cdiv(d->N, 256)), dim3(256), 0, st, kp,
                         pl.mout, pl.rb);
      ADN_CHECK_LAUNCH();
    }
  } else {
    const int64_t total = pl.mout * d->N;
    int64_t blocks = adn_cdiv(total, 256);
    if (blocks > 65536) blocks = 65536;
    if (d->geom == ADN_GEMM_S2)
      hipLaunchKernelGGL((igemm_direct_kernel<T, ADN_GEMM_S2>), dim3((unsigned)blocks), dim3(256), 0, st, kp);
    else if (d->geom == ADN_GEMM_T2)
      hipLaunchKernelGGL((igemm_direct_kernel<T, ADN_GEMM_T2>), dim3((unsigned)blocks), dim3(256), 0, st, kp);
    else
      hipLaunchKernelGGL((igemm_direct_kernel<T, ADN_GEMM_S1>), dim3((unsigned)blocks), dim3(256), 0, st, kp);
    ADN_CHECK_LAUNCH();
    kp.nsplit = 1;
    hipLaunchKernelGGL((igemm_reduce_kernel<T>),
                       dim3((unsigned)adn_cdiv(pl.mout, pl.rb), (unsigned)adn_cdiv(d->N, 256)), dim3(256), 0, st, kp,
                       pl.mout, pl.rb);
    ADN_CHECK_LAUNCH();
  }
  return ADN_OK;
}

int validate(const AdnIgemmDesc* d) {
  ADN_CHECK_ARG(d != nullptr, "adn_igemm: null descriptor");
  ADN_CHECK_ARG(d->dtype == ADN_F32 || d->dtype == ADN_BF16, "adn_igemm: bad dtype %d", d->dtype);
  ADN_CHECK_ARG(d->geom == ADN_GEMM_S2 || d->geom == ADN_GEMM_T2 || d->geom == ADN_GEMM_S1, "adn_igemm: bad geom %d",
                d->geom);
  ADN_CHECK_ARG(d->geom != ADN_GEMM_S1 || d->ks == 1 || d->ks == 3, "adn_igemm: S1 kernel side must be 1 or 3 (got %d)",
                d->ks);
  ADN_CHECK_ARG(d->B > 0 && d->Hs > 0 && d->Ws > 0, "adn_igemm: bad shape B=%d Hs=%d Ws=%d", d->B, d->Hs, d->Ws);
  ADN_CHECK_ARG(d->C0 > 0 && d->C1 >= 0 && d->N > 0, "adn_igemm: bad channels C0=%d C1=%d N=%d", d->C0, d->C1, d->N);
  ADN_CHECK_ARG(d->in0 && d->w && (d->C1 == 0 || d->in1), "adn_igemm: null operand");
  ADN_CHECK_ARG(d->epi >= ADN_EPI_RAW && d->epi <= ADN_EPI_ADD, "adn_igemm: bad epilogue %d", d->epi);
  ADN_CHECK_ARG(d->seg[0].channels + d->seg[1].channels == d->N && d->seg[0].channels > 0 && d->seg[1].channels >= 0,
                "adn_igemm: segment channels %d+%d != N=%d", d->seg[0].channels, d->seg[1].channels, d->N);
  for (int s = 0; s < 2; ++s) {
    if (d->seg[s].channels == 0) continue;
    const AdnEpiSeg& g = d->seg[s];
    if (d->epi != ADN_EPI_ACT) ADN_CHECK_ARG(g.out0, "adn_igemm: seg %d out0 is null", s);
    if (d->epi == ADN_EPI_ACT) ADN_CHECK_ARG(g.out0 || g.out1, "adn_igemm: seg %d has no output", s);
    if (d->epi == ADN_EPI_BWD) {
      ADN_CHECK_ARG(g.ref, "adn_igemm: seg %d BWD needs ref", s);
      if (g.partials) ADN_CHECK_ARG(g.z && g.mean && g.istd, "adn_igemm: seg %d BWD stats need z/mean/istd", s);
    }
  }
  // 32-bit index safety of the per-tensor element counts
  const int64_t big = (int64_t)d->B * d->Hs * d->Ws * 4;
  ADN_CHECK_ARG(big * (d->C0 + d->C1) < (1ll << 40) && big < (1ll << 31), "adn_igemm: tensor too large");
  // the MFMA loader addresses each gathered source through a 2 GiB buffer descriptor with 32-bit byte offsets
  const int64_t gpix = (int64_t)d->B * d->Hs * d->Ws * (d->geom == ADN_GEMM_S2 ? 4 : 1);
  ADN_CHECK_ARG(gpix * (d->C0 > d->C1 ? d->C0 : d->C1) * (d->dtype == ADN_BF16 ? 2 : 4) < 0x7ff00000ll,
                "adn_igemm: a gathered source exceeds 2 GiB (B=%d %dx%d C=%d/%d)", d->B, d->Hs, d->Ws, d->C0, d->C1);
  return ADN_OK;
}

}  // namespace

extern "C" int64_t adn_igemm_num_partials(const AdnIgemmDesc* d) {
  if (validate(d) != ADN_OK) return -1;
  Plan pl;
  make_plan(d, &pl);
  return pl.partial_rows;
}

extern "C" int64_t adn_igemm_workspace_bytes(const AdnIgemmDesc* d) {
  if (validate(d) != ADN_OK) return -1;
  Plan pl;
  make_plan(d, &pl);
  return pl.slab_bytes;
}

extern "C" int adn_igemm(const AdnIgemmDesc* d, void* stream) {
  int rc = validate(d);
  if (rc != ADN_OK) return rc;
  Plan pl;
  make_plan(d, &pl);
  ADN_CHECK_ARG(pl.slab_bytes == 0 || (d->workspace && d->workspace_bytes >= pl.slab_bytes),
                "adn_igemm: workspace too small (%lld < %lld)", (long long)d->workspace_bytes,
                (long long)pl.slab_bytes);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d->dtype == ADN_BF16) return run<uint16_t>(d, pl, st);
  return run<float>(d, pl, st);
}
