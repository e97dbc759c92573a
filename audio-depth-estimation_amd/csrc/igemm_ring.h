// Ring-fed persistent implicit GEMM (round 3): bf16, wide channels, unsplit, S2 (k4 s2 p1 conv / dgrad of the transposed
// conv) and T2 (one phase of the transposed conv / dgrad of the conv) geometries.  Included by igemm.hip inside its
// anonymous namespace (KParams, xcd_remap, epilogue.h are in scope).
//
// Why a second structure beside igemm_patch_kernel: that kernel is "wait vmcnt(0) -> barrier -> issue the NEXT step's
// LDS-DMA -> 32 MFMAs" with two independent 4-wave workgroups per CU hiding each other's round trips; every K-step waits for
// data requested one step earlier, the epilogue tile (68 KB of LDS) caps the CU at two workgroups, and a layer is one or two
// rounds of workgroups that all run their prologue and epilogue at the same time.  Here:
//   * ONE 8-wave workgroup per CU (two waves per SIMD), PERSISTENT over its tiles (grid = min(tiles, CUs)); a tile is
//     16 x 16 output pixels of one image x BN (128 | 64) output channels, every wave owns 64 pixels x BN/2 channels;
//   * all of the LDS is a staging RING that never drains: weights [4 slots][2 taps][BN][64 B] requested THREE K-steps ahead,
//     input patches (S2: 2 buffers of 2 x 17 x 17 pixels, T2: 4 buffers of 17 x 17 pixels, 32 channels = 64 B per pixel)
//     requested one / two segments ahead, counted s_waitcnt vmcnt(N) (never 0 in steady state), one raw s_barrier per
//     K-step; the ring runs ACROSS tile boundaries, so the next tile's first steps land under the current tile's epilogue;
//   * at the top of step s everything step s+1 needs has landed, so the fragments of step s+1's first tap are read
//     during the last tap of step s, across the barrier (explicit double-buffered fragment registers);
//   * every LDS buffer index is a compile-time constant (the K loop is unrolled over a super-step of 8 K-steps = one
//     period of the weight ring and the patch buffers), so a fragment read is "per-lane base register + immediate":
//     no address arithmetic in the loop;
//   * the MFMA operands are SWAPPED (A = weights, B = pixels): a lane then holds 4 consecutive CHANNELS of one pixel
//     per accumulator, i.e. 8 contiguous bytes of the NHWC output -- the epilogue runs straight from the registers
//     (no LDS tile, no barrier for the data), BatchNorm column sums by DPP row rotations.
// Tile order: t = (tile_m * phases + phase) * tiles_n + tile_n; workgroup w (XCD-contiguous numbering) owns tiles
// w, w + G, w + 2G ...: at any time the 32 CUs of an XCD work on 32 consecutive tiles (shared patches and weights in L2).
#pragma once

#include <type_traits>

template <int N>
__device__ __forceinline__ void ring_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// v + (v rotated right by SH lanes inside its 16-lane DPP row)
template <int SH>
__device__ __forceinline__ float ring_row_ror_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + SH, 0xf, 0xf, false));
}
__device__ __forceinline__ float ring_row_sum(float v) {
  v = ring_row_ror_add<8>(v);
  v = ring_row_ror_add<4>(v);
  v = ring_row_ror_add<2>(v);
  return ring_row_ror_add<1>(v);
}

// Reduce-scatter of per-lane values over the 16 lanes of a DPP row, four results per call.  Step A pairs lane L with 15 - L
// (row_mirror: bit 3 differs), lanes 0-7 keep the sums of the `lo` values, lanes 8-15 those of the `hi` values (bank masks
// 0x3 / 0xc: a disabled lane keeps its destination); step B pairs L with its mirror inside the half row (bit 2 differs,
// bank masks 0x5 / 0xa).  One v_add_f32_dpp per kept value and step instead of v_mov_b32_dpp + add on every value: a row
// total of V values per lane costs V + V/2 + 2 * V/4 instructions instead of 6 V.  (s_nop 1: a VALU result may not be read
// by a DPP operand in the next two issue slots; the hazard recogniser does not look inside inline assembly.)
__device__ __forceinline__ void ring_rs_mirror4(const float* lo, const float* hi, float* out) {
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %4, %4 row_mirror row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %1, %5, %5 row_mirror row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %2, %6, %6 row_mirror row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %3, %7, %7 row_mirror row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %0, %8, %8 row_mirror row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %1, %9, %9 row_mirror row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %2, %10, %10 row_mirror row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %3, %11, %11 row_mirror row_mask:0xf bank_mask:0xc"
      : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3])
      : "v"(lo[0]), "v"(lo[1]), "v"(lo[2]), "v"(lo[3]), "v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3]));
}
__device__ __forceinline__ void ring_rs_half_mirror4(const float* lo, const float* hi, float* out) {
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %4, %4 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %1, %5, %5 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %2, %6, %6 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %3, %7, %7 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %0, %8, %8 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %1, %9, %9 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %2, %10, %10 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %3, %11, %11 row_half_mirror row_mask:0xf bank_mask:0xa"
      : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3])
      : "v"(lo[0]), "v"(lo[1]), "v"(lo[2]), "v"(lo[3]), "v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3]));
}
// all-reduce over the four lanes of a quad (every lane ends with the quad total), four values per call
__device__ __forceinline__ void ring_quad_sum4(float* v) {
  float t[4];
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
      : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3])
      : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %6, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %7, %7 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
      : "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]));
}

#ifdef ADN_RING_STAMPS
// diagnostic build only (tools/ring_diag.py): s_memtime around the parts of a K-step, sums written to the workspace
#define RING_STAMP(var)                                                                   \
  do {                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");          \
    __builtin_amdgcn_sched_barrier(0);                                                    \
  } while (0)
#else
#define RING_STAMP(var) do { } while (0)
#endif

constexpr unsigned RING_OOB = 0x80000000u;      // voffset that fails the buffer range check: the LDS-DMA writes zeros

struct RingTile {
  int tile_n, phase, tb, oy0, ox0, tile_m;
};

template <int GEOM, int BN, int SCHED>
__global__ __launch_bounds__(512, 2) void igemm_ring_kernel(KParams p, int ntiles, int nwgs, unsigned mg_tn, unsigned mg_tpi,
                                                             unsigned mg_tpr) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr bool S2 = GEOM == ADN_GEMM_S2;
  constexpr int MW = 17, PLANE = 17 * 17;
  constexpr int SEG_PIX = S2 ? 2 * PLANE : PLANE;
  constexpr int SEG_STEPS = S2 ? 4 : 2;
  constexpr int PPIECES = S2 ? 40 : 20;             // 1-KiB LDS-DMA pieces per patch segment (S2: 37 used, padded to 8 waves)
  constexpr int PKW = S2 ? 5 : 3;                   // pieces per wave (T2: waves 4..7 repeat pieces 16..19 of waves 0..3)
  constexpr int PBUF = PPIECES * 1024;
  constexpr int NPBUF = S2 ? 2 : 4;
  constexpr int PAHEAD = S2 ? 1 : 2;                // segments the patch requests run ahead of the compute
  constexpr int NT = BN / 32;                       // 16-channel MFMA tiles per wave (two waves along the channels)
  constexpr int NH = NT / 2;                        // 8-channel (16-byte) output groups per lane
  constexpr int WPW = BN / 64;                      // weight pieces per wave and K-step
  constexpr int WBUF = 2 * BN * 64;                 // one ring slot: [2 taps][BN rows][64 B]
  constexpr int RING = 4, WAHEAD = 3;
  constexpr int W_OFF = NPBUF * PBUF, R_OFF = W_OFF + RING * WBUF;
  constexpr int NPH = S2 ? 1 : 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 15, fq = lane >> 4;
  const int Hs = p.Hs, Ws = p.Ws, Hl = 2 * Hs, Wl = 2 * Ws;
  const int Cin = p.C0 + p.C1;
  const int Hg = S2 ? Hl : Hs, Wg = S2 ? Wl : Ws;
  const int ktot = p.wstride;
  const int wgx = xcd_remap(blockIdx.x, nwgs);
  const int my_tiles = (ntiles - wgx + nwgs - 1) / nwgs;
  const int nchunks = Cin >> 5;
  const int nseg = S2 ? 2 * nchunks : nchunks;
  const int nsuper = nseg * SEG_STEPS / 8;          // super-steps (8 K-steps) per tile
  const int tpr = Ws >> 4, tpi = (Hs >> 4) * tpr;   // 16 x 16 tiles per image row / per image
  if (my_tiles <= 0) return;

  // tile id -> (column tile, phase, image, pixel tile): divisions by launch constants as multiply-high with host-made
  // reciprocals (floor(2^32 / d) + 1: exact for x * d < 2^32, tile ids are < 2^20) -- the three request cursors and the
  // epilogue decode a tile each, a hardware-less integer division is ~25 instructions
  auto fdiv = [](int x, unsigned magic, int d) -> int { return d == 1 ? x : (int)__umulhi((unsigned)x, magic); };
  auto decode_tile = [&](int k, RingTile& t) {
    const int id = wgx + k * nwgs;
    const int r = fdiv(id, mg_tn, p.tiles_n);
    t.tile_n = id - r * p.tiles_n;
    t.phase = r % NPH;
    t.tile_m = r / NPH;
    t.tb = fdiv(t.tile_m, mg_tpi, tpi);
    const int rem = t.tile_m - t.tb * tpi;
    const int ry = fdiv(rem, mg_tpr, tpr);
    t.oy0 = ry << 4;
    t.ox0 = (rem - ry * tpr) << 4;
  };

  typedef __attribute__((address_space(3))) void* lptr_t;
  const int bshift = Wg + 1;
  const char* gbase0 = reinterpret_cast<const char*>(p.in0) - (int64_t)bshift * p.C0 * 2;
  const char* gbase1 = reinterpret_cast<const char*>(p.C1 ? p.in1 : p.in0) - (int64_t)bshift * p.C1 * 2;
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.rec_b, 0x00020000);

  // ---- per-lane constants (patch / weight loader offsets, fragment base addresses) ----
  // ~40 registers that are dead during the epilogue: they are RE-COMPUTED after every epilogue from an opaque copy of the
  // lane id (so that the compiler cannot keep the first computation alive instead), which is what lets the BWD epilogue
  // keep its operand loads in flight without spilling
  auto piece_of = [&](int k) -> int { return (!S2 && k == 2) ? 16 + (wave & 3) : wave + 8 * k; };
  unsigned prel[PKW];          // pixel offset of this lane's pixel relative to the tile's patch origin
  unsigned phm[PKW];           // hr | m << 8 | column parity << 16 | (pixel beyond the segment) << 24
  unsigned lc16;               // source-side chunk swizzle of the patch pieces: (q >> 2) & 1 = (lane >> 4) & 1
  unsigned bvo[WPW];           // weight loader: piece pid = wave + 8k of a step's [2][BN][64 B] tile
  unsigned pa[4][4], wa[NT];   // fragment base addresses (see below)
  auto lane_consts = [&]() {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int l_frow = ln & 15, l_fq = ln >> 4;
#pragma unroll
    for (int k = 0; k < PKW; ++k) {
      const int q = 16 * piece_of(k) + (ln >> 2);
      int cpar = 0, rem = q;
      if constexpr (S2) {
        cpar = q >= PLANE ? 1 : 0;
        rem = q - cpar * PLANE;
      }
      const int hr = rem / MW, m = rem - hr * MW;
      const bool beyond = q >= SEG_PIX;
      phm[k] = (unsigned)hr | ((unsigned)m << 8) | ((unsigned)cpar << 16) | (beyond ? 1u << 24 : 0u);
      prel[k] = S2 ? (unsigned)(2 * hr * Wg + 2 * m + cpar) : (unsigned)(hr * Wg + m);
    }
    lc16 = (unsigned)(((ln & 3) ^ (((ln >> 4) & 1) << 1)) << 4);
    // LDS row L of a tap holds output channel chan(L): inside a wave's BN/2 rows, MFMA tile j (L >> 4) row rho (L & 15) is
    // channel (j >> 1) * 32 + (rho >> 2) * 8 + (j & 1) * 4 + (rho & 3), so that a lane's accumulators of tiles 2h, 2h + 1 are
    // the 8 CONSECUTIVE channels h * 32 + fq * 8 .. + 7 of its pixel: 16-byte epilogue accesses, 64 contiguous bytes per
    // pixel and wave-instruction
#pragma unroll
    for (int k = 0; k < WPW; ++k) {
      const int pid = wave + 8 * k;
      const int tsel = pid / (BN / 16), row = (pid % (BN / 16)) * 16 + (ln >> 2);
      const int l = row % (BN / 2), j = l >> 4, rho = l & 15;
      const int chan = (row - l) + (j >> 1) * 32 + (rho >> 2) * 8 + (j & 1) * 4 + (rho & 3);
      const int lc = (ln & 3) ^ (((row >> 2) & 1) << 1);
      bvo[k] = (unsigned)((chan * ktot + tsel * Cin + lc * 8) * 2);
    }
    // pixel fragment of pixel-row i at tap offset qoff: LDS pixel q = q0[i] + qoff, chunk fq ^ (((q >> 2) & 1) << 1); the
    // swizzle bit only depends on qoff & 7 (adding a multiple of 8 pixels keeps bit 2), and the tap offsets of both
    // geometries have qoff & 7 in {0, 1, 2, 3}: four base registers per pixel-row, everything else is an immediate
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q0 = (wm * 4 + i) * MW + l_frow;
#pragma unroll
      for (int c = 0; c < 4; ++c) pa[i][c] = (unsigned)(q0 * 64 + ((l_fq ^ ((((q0 + c) >> 2) & 1) << 1)) << 4));
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int row = wn * (BN / 2) + j * 16 + l_frow;
      wa[j] = (unsigned)(W_OFF + row * 64 + ((l_fq ^ (((row >> 2) & 1) << 1)) << 4));
    }
  };
  lane_consts();

  // ---- request cursors (wave-uniform scalars); a dead cursor keeps issuing out-of-range requests (zeros into slots nobody
  // reads any more) so that the request count per K-step never changes ----
  int w_k = 0, w_seg = 0, w_ss = 0, w_base = 0;
  bool w_live = true;
  auto w_enter = [&]() {
    RingTile t;
    decode_tile(w_k, t);
    w_base = ((t.phase * p.N + t.tile_n * BN) * ktot) * 2;
  };
  w_enter();
  auto issue_w = [&](auto SLOT) {                   // straight-line: the weights of the cursor's K-step into ring slot SLOT
    constexpr int slot = decltype(SLOT)::value;
    int c, t0;
    if constexpr (S2) {
      c = w_seg >> 1;
      t0 = (2 * (w_ss >> 1) + (w_seg & 1)) * 4 + 2 * (w_ss & 1);
    } else {
      c = w_seg;
      t0 = 2 * w_ss;
    }
    const int soff = w_base + (t0 * Cin + (c << 5)) * 2;
    char* dst = smem + W_OFF + slot * WBUF + wave * 1024;
#pragma unroll
    for (int k = 0; k < WPW; ++k)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lptr_t)(dst + k * 8192), 16, w_live ? bvo[k] : RING_OOB, soff, 0, 0);
  };
  auto w_next = [&]() {
    if (++w_ss == SEG_STEPS) {
      w_ss = 0;
      if (++w_seg == nseg) {
        w_seg = 0;
        if (++w_k >= my_tiles) w_live = false;
        else w_enter();
      }
    }
  };
  int p_k = 0, p_seg = 0;
  unsigned p_origin = 0;       // pixel index (+ bshift) of the patch origin
  unsigned p_mask = 0;         // bit 2k: pixel valid for row parity 0 (T2: valid), bit 2k+1: row parity 1
  auto p_enter = [&]() {
    RingTile t;
    decode_tile(p_k, t);
    int y0, x0;                // gathered-image coordinates of patch pixel (hr = 0, m = 0, parity 0)
    if constexpr (S2) {
      y0 = 2 * t.oy0 - 1;
      x0 = 2 * t.ox0 - 1;
    } else {
      y0 = t.oy0 + ((t.phase >> 1) == 0 ? -1 : 0);
      x0 = t.ox0 + ((t.phase & 1) == 0 ? -1 : 0);
    }
    p_origin = (unsigned)((t.tb * Hg + y0) * Wg + x0 + bshift);
    p_mask = 0;
#pragma unroll
    for (int k = 0; k < PKW; ++k) {
      const int hr = phm[k] & 0xff, m = (phm[k] >> 8) & 0xff, cpar = (phm[k] >> 16) & 1;
      const bool inseg = (phm[k] >> 24) == 0;
      const int iy = S2 ? y0 + 2 * hr : y0 + hr;
      const int ix = S2 ? x0 + 2 * m + cpar : x0 + m;
      const bool okx = inseg && (unsigned)ix < (unsigned)Wg;
      const bool ok0 = okx && (unsigned)iy < (unsigned)Hg;
      const bool ok1 = S2 ? (okx && (unsigned)(iy + 1) < (unsigned)Hg) : ok0;
      p_mask |= (ok0 ? 1u : 0u) << (2 * k) | (ok1 ? 2u : 0u) << (2 * k);
    }
  };
  p_enter();
  // straight-line: pieces K0 .. K1-1 of the cursor's segment into patch buffer PB
  auto issue_p = [&](auto PB, auto K0, auto K1) {
    constexpr int pb = decltype(PB)::value, k0 = decltype(K0)::value, k1 = decltype(K1)::value;
    const int c = S2 ? p_seg >> 1 : p_seg, par = S2 ? p_seg & 1 : 0;
    const int c0 = c << 5;
    const bool second = c0 >= p.C0;
    const int Cs = second ? p.C1 : p.C0;
    const int coff = second ? c0 - p.C0 : c0;
    const int soff = (par * Wg * Cs + coff) * 2;
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(second ? gbase1 : gbase0), 0, p.rec_a, 0x00020000);
#pragma unroll
    for (int k = k0; k < k1; ++k) {
      char* dst = smem + pb * PBUF + piece_of(k) * 1024;
      const bool ok = (p_mask >> (2 * k + par)) & 1u;
      const unsigned vo = ok ? (p_origin + prel[k]) * (unsigned)(Cs * 2) + lc16 : RING_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)dst, 16, vo, soff, 0, 0);
    }
  };
  auto p_next = [&]() {
    if (++p_seg == nseg) {
      p_seg = 0;
      if (++p_k >= my_tiles) p_mask = 0;           // dead cursor: every further request is out of range
      else p_enter();
    }
  };

  // tap offset (LDS pixels) of tap e of step-in-segment ss
  auto qoff_of = [](int ss, int e) constexpr -> int {
    return S2 ? ((e * 17) + (ss >> 1)) * MW + (ss & 1) : (ss == 0 ? 1 : 0) * MW + (e == 0 ? 1 : 0);
  };

  f32x4_t acc[4][NT];
  u32x4_t pfA[4], wfA[NT], pfB[4], wfB[NT];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  };
  // fragments of (super-step position u, tap e) -> registers
  auto load_frags = [&](auto U, auto E, u32x4_t* pf, u32x4_t* wf) {
    constexpr int u = decltype(U)::value & 7, e = decltype(E)::value;
    constexpr int ss = u % SEG_STEPS, pb = (u / SEG_STEPS) % NPBUF, slot = u & 3;
    constexpr int qo = qoff_of(ss, e);
    constexpr int pimm = pb * PBUF + qo * 64;
    constexpr int wimm = slot * WBUF + e * (BN * 64);
#pragma unroll
    for (int i = 0; i < 4; ++i) pf[i] = *reinterpret_cast<const u32x4_t*>(smem + pa[i][qo & 7] + pimm);
#pragma unroll
    for (int j = 0; j < NT; ++j) wf[j] = *reinterpret_cast<const u32x4_t*>(smem + wa[j] + wimm);
  };
  auto mma = [&](const u32x4_t* pf, const u32x4_t* wf) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(&wf[j]),
                                                            *reinterpret_cast<const bf16x8_t*>(&pf[i]), acc[i][j], 0, 0, 0);
  };

  // ---- epilogue straight from the accumulators ----
  // acc[i][2h + b][r] = pixel (tile row wm*4 + i, column frow), channel n0 + 32 h + 8 fq + 4 b + r
  float* red = reinterpret_cast<float*>(smem + R_OFF);      // [8 waves][2][BN/2]
#ifdef ADN_RING_STAMPS
  unsigned long long sum_ea = 0, sum_eb = 0, sum_ec = 0, sum_ed = 0;
#endif
  // The data-dependent options (bias, accumulate, BatchNorm sums) are compile-time flags of the body: tested per element
  // they became a branch around every single load, and the loads of an absent bias made the Z_STATS path wait vmcnt(0)
  // -- i.e. for the whole staging ring -- before its first store.
  auto epilogue_body = [&](int k, auto BWD_, auto FLAG_A, auto FLAG_S, auto FLAG_Z) {
    constexpr bool BWD = decltype(BWD_)::value;
    constexpr bool FA = decltype(FLAG_A)::value;       // Z_STATS: the conv has a bias;  BWD: accumulate into the output
    constexpr bool STATS = decltype(FLAG_S)::value;    // this wave's segment takes BatchNorm sums
    [[maybe_unused]] unsigned long long x0, x1, x2, x3, x4;
    RING_STAMP(x0);
    RingTile t;
    decode_tile(k, t);
    const int n0 = t.tile_n * BN + wn * (BN / 2);             // first channel of this wave (wave-uniform)
    const bool first = n0 < p.seg[0].channels;
    const AdnEpiSeg& sg = first ? p.seg[0] : p.seg[1];
    const int nl0 = (first ? n0 : n0 - p.seg[0].channels) + 8 * fq;
    const int ph = t.phase >> 1, pw = t.phase & 1;
    unsigned opix[4];            // element offsets (validate() bounds every tensor of the launch by 2^31 elements)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int oy = t.oy0 + wm * 4 + i, ox = t.ox0 + frow;
      opix[i] = (unsigned)(S2 ? (t.tb * Hs + oy) * Ws + ox : (t.tb * Hl + 2 * oy + ph) * Wl + 2 * ox + pw) * (unsigned)sg.channels + (unsigned)nl0;
    }
    const bool any_stats = p.seg[0].partials != nullptr || p.seg[1].partials != nullptr;     // workgroup-uniform
    uint16_t* out = reinterpret_cast<uint16_t*>(sg.out0);
    float s1[NH][8], s2[NH][8];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int e = 0; e < 8; ++e) s1[h][e] = s2[h][e] = 0.f;
    auto pack8 = [&](int i, int h) -> u32x4_t {
      u32x4_t pk;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        pk[2 * b] = (uint32_t)f32_to_bf16_bits(acc[i][2 * h + b][0]) | ((uint32_t)f32_to_bf16_bits(acc[i][2 * h + b][1]) << 16);
        pk[2 * b + 1] = (uint32_t)f32_to_bf16_bits(acc[i][2 * h + b][2]) | ((uint32_t)f32_to_bf16_bits(acc[i][2 * h + b][3]) << 16);
      }
      return pk;
    };
    if constexpr (!BWD) {
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        float bias[8];
        if constexpr (FA) {
          const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(sg.bias + nl0 + 32 * h);
          const f32x4_t b1 = *reinterpret_cast<const f32x4_t*>(sg.bias + nl0 + 32 * h + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            bias[e] = b0[e];
            bias[4 + e] = b1[e];
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float v = acc[i][2 * h + (e >> 2)][e & 3];
            if constexpr (FA) {
              v += bias[e];
              acc[i][2 * h + (e >> 2)][e & 3] = v;
            }
            if constexpr (STATS) {
              s1[h][e] += v;
              s2[h][e] += v * v;
            }
          }
          *reinterpret_cast<u32x4_t*>(out + opix[i] + 32 * h) = pack8(i, h);
        }
      }
    } else {   // BWD: one pixel-row at a time, the next row's operands requested before this one is worked on; the outputs
               // are stored after the LAST operand load (vmcnt retires in issue order: a store between two loads would
               // put its write latency in front of the next load's data)
      const uint16_t* ref = reinterpret_cast<const uint16_t*>(sg.ref);
      const uint16_t* zz = reinterpret_cast<const uint16_t*>(sg.z);
      const float slope = sg.slope;
      // A segment with BatchNorm sums reads its pre-activation z anyway; when the caller also passes the forward's
      // scale / shift, the activation's sign is that of fma(z, scale, shift) -- the very value the forward's apply kernel
      // rounded to `ref` -- and `ref` is not read at all (a third of the operand traffic of an input-gradient launch that
      // is HBM-bound at the wide levels).
      constexpr bool MZ = STATS && decltype(FLAG_Z)::value;
      u32x4_t rr[2][NH], oo[2][NH], zr[2][NH];
      [[maybe_unused]] float msc[NH][8], msh[NH][8];
      if constexpr (MZ) {
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          const f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(sg.scale + nl0 + 32 * h), a1 = *reinterpret_cast<const f32x4_t*>(sg.scale + nl0 + 32 * h + 4);
          const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(sg.shift + nl0 + 32 * h), b1 = *reinterpret_cast<const f32x4_t*>(sg.shift + nl0 + 32 * h + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            msc[h][e] = a0[e];
            msc[h][4 + e] = a1[e];
            msh[h][e] = b0[e];
            msh[h][4 + e] = b1[e];
          }
        }
      }
      auto request = [&](int i, u32x4_t* r_, u32x4_t* o_, u32x4_t* z_) {
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          const unsigned idx = opix[i] + 32 * h;
          if constexpr (!MZ) r_[h] = *reinterpret_cast<const u32x4_t*>(ref + idx);
          if constexpr (FA) o_[h] = *reinterpret_cast<const u32x4_t*>(out + idx);
          if constexpr (STATS) z_[h] = *reinterpret_cast<const u32x4_t*>(zz + idx);
        }
      };
      request(0, rr[0], oo[0], zr[0]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i + 1 < 4) request(i + 1, rr[(i + 1) & 1], oo[(i + 1) & 1], zr[(i + 1) & 1]);
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const uint32_t sh = (e & 1) ? 0u : 16u;
            [[maybe_unused]] float zf = 0.f;
            if constexpr (STATS) zf = __uint_as_float((zr[i & 1][h][e >> 1] << sh) & 0xffff0000u);
            float rf;
            if constexpr (MZ) rf = __builtin_fmaf(zf, msc[h][e], msh[h][e]);
            else rf = __uint_as_float((rr[i & 1][h][e >> 1] << sh) & 0xffff0000u);
            float g = acc[i][2 * h + (e >> 2)][e & 3] * (rf > 0.f ? 1.0f : slope);
            if constexpr (FA) g += __uint_as_float((oo[i & 1][h][e >> 1] << sh) & 0xffff0000u);
            if constexpr (STATS) {       // s2 = sum g * z here; (sum g z - mean sum g) * istd once per channel below
              s1[h][e] += g;
              s2[h][e] += g * zf;
            }
            acc[i][2 * h + (e >> 2)][e & 3] = g;
          }
      }
#pragma unroll
      for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4_t*>(out + opix[i] + 32 * h) = pack8(i, h);
    }
    RING_STAMP(x1);
    // column sums over the 16 pixels of a DPP row as a reduce-scatter (see ring_rs_mirror4): the lanes of a row end with
    // the totals of DIFFERENT values -- lanes 0-7 the sums (s1), lanes 8-15 the second statistic (s2); inside each half,
    // lanes 0-3 the first and lanes 4-7 the second half of the channels -- then over the four pixel-row waves that share
    // these channels through LDS
    if (any_stats) {
      constexpr int HV = NH * 8;           // values per statistic and lane
      constexpr int QV = HV / 2;           // values a lane keeps
      float kept[QV];
      if constexpr (STATS) {
        float half[HV];
#pragma unroll
        for (int k = 0; k < HV; k += 4) ring_rs_mirror4(&s1[0][0] + k, &s2[0][0] + k, half + k);
#pragma unroll
        for (int k = 0; k < QV; k += 4) ring_rs_half_mirror4(half + k, half + QV + k, kept + k);
#pragma unroll
        for (int k = 0; k < QV; k += 4) ring_quad_sum4(kept + k);
      } else {
#pragma unroll
        for (int k = 0; k < QV; ++k) kept[k] = 0.f;
      }
      if ((frow & 3) == 0) {
        const int st = frow >> 3, q = (frow >> 2) & 1;
        // NH = 2: the kept values are channels 32 q + 8 fq + 0..7 of this wave; NH = 1: channels 8 fq + 4 q + 0..3
        float* r = red + (wave * 2 + st) * (BN / 2) + (NH == 2 ? 32 * q + 8 * fq : 8 * fq + 4 * q);
#pragma unroll
        for (int k = 0; k < QV; k += 4) *reinterpret_cast<f32x4_t*>(r + k) = f32x4_t{kept[k], kept[k + 1], kept[k + 2], kept[k + 3]};
      }
      RING_STAMP(x2);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (not __syncthreads(): its vmcnt(0) would drain the ring)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      RING_STAMP(x3);
      if (tid < 2 * BN) {
        const int st = tid / BN, c = tid % BN;
        const int cw = c / (BN / 2), cc = c % (BN / 2);
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) tot += red[((w * 2 + cw) * 2 + st) * (BN / 2) + cc];
        const int n = t.tile_n * BN + c;
        const AdnEpiSeg& sq = (n < p.seg[0].channels) ? p.seg[0] : p.seg[1];
        const int ncl = (n < p.seg[0].channels) ? n : n - p.seg[0].channels;
        const int64_t P = (int64_t)t.phase * p.tiles_m + t.tile_m;
        if constexpr (BWD) {       // second statistic of the backward epilogue: (sum g z - mean sum g) * istd
          if (st == 1 && sq.partials) {
            float tg = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) tg += red[((w * 2 + cw) * 2 + 0) * (BN / 2) + cc];
            tot = (tot - sq.mean[ncl] * tg) * sq.istd[ncl];
          }
        }
        if (sq.partials) sq.partials[(P * 2 + st) * sq.channels + ncl] = tot;
      }
#ifdef ADN_RING_STAMPS
      RING_STAMP(x4);
      sum_ea += x1 - x0; sum_eb += x2 - x1; sum_ec += x3 - x2; sum_ed += x4 - x3;
#endif
    }
  };
  auto epilogue = [&](int k) {
    using T_ = std::true_type;
    using F_ = std::false_type;
    // (wave-uniform dispatch: which segment this wave's channels belong to only depends on the tile and wn)
    RingTile t;
    decode_tile(k, t);
    const int n0 = t.tile_n * BN + wn * (BN / 2);
    const AdnEpiSeg& sg = n0 < p.seg[0].channels ? p.seg[0] : p.seg[1];
    const bool stats = sg.partials != nullptr;
    if (p.epi == ADN_EPI_Z_STATS) {
      const bool bias = sg.bias != nullptr;
      if (bias) {
        if (stats) epilogue_body(k, F_{}, T_{}, T_{}, F_{});
        else epilogue_body(k, F_{}, T_{}, F_{}, F_{});
      } else {
        if (stats) epilogue_body(k, F_{}, F_{}, T_{}, F_{});
        else epilogue_body(k, F_{}, F_{}, F_{}, F_{});
      }
    } else {
      const bool accu = sg.accumulate != 0;
      const bool zmask = stats && sg.scale != nullptr && sg.shift != nullptr;      // activation mask from z (no `ref` read)
      if (accu) {
        if (zmask) epilogue_body(k, T_{}, T_{}, T_{}, T_{});
        else if (stats) epilogue_body(k, T_{}, T_{}, T_{}, F_{});
        else epilogue_body(k, T_{}, T_{}, F_{}, F_{});
      } else {
        if (zmask) epilogue_body(k, T_{}, F_{}, T_{}, T_{});
        else if (stats) epilogue_body(k, T_{}, F_{}, T_{}, F_{});
        else epilogue_body(k, T_{}, F_{}, F_{}, F_{});
      }
    }
  };

  // ---- prologue: weights of steps 0 .. WAHEAD-1, the first PAHEAD patch segments in full ----
  issue_w(std::integral_constant<int, 0>{});
  w_next();
  issue_w(std::integral_constant<int, 1>{});
  w_next();
  issue_w(std::integral_constant<int, 2>{});
  w_next();
  issue_p(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, PKW>{});
  p_next();
  if constexpr (PAHEAD == 2) {
    issue_p(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, PKW>{});
    p_next();
  }
  ring_wait<0>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  load_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, pfA, wfA);
  zero_acc();

#ifdef ADN_RING_STAMPS
  unsigned long long sum_wait = 0, sum_issue = 0, sum_tap0 = 0, sum_tap1 = 0, sum_epi = 0, n_steps = 0, t_begin, t_end;
  RING_STAMP(t_begin);
  const unsigned long long r_begin = __builtin_amdgcn_s_memrealtime();      // 100 MHz: shader clock = cycles / ticks * 100 MHz
#endif
  bool after_epi = false;       // the 4 * NH output stores of the epilogue are younger than the previous request group
  // one K-step at super-step position U
  auto step = [&](auto U) {
    constexpr int u = decltype(U)::value;
    constexpr int ss = u % SEG_STEPS;
    // patch pieces requested in this step / in the previous one (S2: 3, 2, 0, 0 over a segment's steps; T2: 2, 1)
    constexpr int np = S2 ? (ss == 0 ? 3 : (ss == 1 ? 2 : 0)) : (ss == 0 ? 2 : 1);
    constexpr int np_prev = S2 ? (ss == 1 ? 3 : (ss == 2 ? 2 : 0)) : (ss == 1 ? 2 : 1);
    [[maybe_unused]] unsigned long long st0, st1, st2, st3, st4;
    RING_STAMP(st0);
    __builtin_amdgcn_sched_barrier(0);
    // everything but the previous step's request group has landed (vmcnt counts in issue order)
    if constexpr (u == 0) {
      if (after_epi) ring_wait<WPW + np_prev + 4 * NH>();
      else ring_wait<WPW + np_prev>();
    } else {
      ring_wait<WPW + np_prev>();
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    RING_STAMP(st1);
    // ---- straight-line block: this step's requests (weights of step +3 into the slot just released, patch pieces PAHEAD
    // segments ahead) are issued BETWEEN the MFMAs: a request costs the issuing wave ~100 cycles, which the other wave of
    // the SIMD fills with its own MFMAs -- issued in a burst behind the barrier (all eight waves at once) they were 30 %
    // of the step with every matrix pipe idle (tools/ring_diag.py) ----
    // region 0: the weight requests, the reads of tap 1's fragments (set B), tap 0's MFMAs (set A, read one region ago)
    issue_w(std::integral_constant<int, (u + WAHEAD) & 3>{});
    RING_STAMP(st2);
    load_frags(U, std::integral_constant<int, 1>{}, pfB, wfB);
    mma(pfA, wfA);
    if constexpr (SCHED == 1) {
      constexpr int reads = 4 + NT, mf = 4 * NT;
#pragma unroll
      for (int g = 0; g < reads; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, mf / reads, 0);           // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                    // DS read
        if (g % 3 == 1 && g / 3 < WPW) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read (an LDS-DMA request)
      }
    }
    RING_STAMP(st3);
    __builtin_amdgcn_sched_barrier(0);
    // region 1: the patch requests, the reads of the NEXT step's tap 0 (set A), tap 1's MFMAs (set B)
    if constexpr (np > 0) {
      constexpr int pb = ((u / SEG_STEPS) + PAHEAD) % NPBUF;
      constexpr int k0 = S2 ? (ss == 0 ? 0 : 3) : (ss == 0 ? 0 : 2);
      issue_p(std::integral_constant<int, pb>{}, std::integral_constant<int, k0>{}, std::integral_constant<int, k0 + np>{});
    }
    load_frags(std::integral_constant<int, u + 1>{}, std::integral_constant<int, 0>{}, pfA, wfA);
    mma(pfB, wfB);
    if constexpr (SCHED == 1) {
      constexpr int reads = 4 + NT, mf = 4 * NT;
#pragma unroll
      for (int g = 0; g < reads; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, mf / reads, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (g % 2 == 1 && g / 2 < np) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    }
    RING_STAMP(st4);
    __builtin_amdgcn_sched_barrier(0);
    // ---- cursor advance (branches; outside the block above) ----
    w_next();
    if constexpr ((S2 && ss == 1) || (!S2 && ss == 1)) p_next();
#ifdef ADN_RING_STAMPS
    sum_wait += st1 - st0;
    sum_issue += st2 - st1;
    sum_tap0 += st3 - st2;
    sum_tap1 += st4 - st3;
    ++n_steps;
#endif
  };

  for (int k = 0; k < my_tiles; ++k) {
    for (int sb = 0; sb < nsuper; ++sb) {
      step(std::integral_constant<int, 0>{});
      after_epi = false;
      step(std::integral_constant<int, 1>{});
      step(std::integral_constant<int, 2>{});
      step(std::integral_constant<int, 3>{});
      step(std::integral_constant<int, 4>{});
      step(std::integral_constant<int, 5>{});
      step(std::integral_constant<int, 6>{});
      step(std::integral_constant<int, 7>{});
    }
#ifdef ADN_RING_STAMPS
    unsigned long long e0, e1;
    RING_STAMP(e0);
#endif
    epilogue(k);
#ifdef ADN_RING_STAMPS
    RING_STAMP(e1);
    sum_epi += e1 - e0;
#endif
    zero_acc();
    lane_consts();
    // (re-read instead of keeping step 7's prefetch alive across the epilogue: 32 registers the epilogue needs)
    load_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, pfA, wfA);
    after_epi = true;
  }
  ring_wait<0>();
#ifdef ADN_RING_STAMPS
  RING_STAMP(t_end);
  if (lane == 0 && p.slab != nullptr) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(p.slab) + ((size_t)wgx * 8 + wave) * 16;
    o[0] = sum_wait; o[1] = sum_issue; o[2] = sum_tap0; o[3] = sum_tap1; o[4] = sum_epi; o[5] = n_steps; o[6] = t_end - t_begin; o[7] = my_tiles;
    o[8] = __builtin_amdgcn_s_memrealtime() - r_begin;
    o[9] = sum_ea; o[10] = sum_eb; o[11] = sum_ec; o[12] = sum_ed;
  }
#endif
#endif
}
