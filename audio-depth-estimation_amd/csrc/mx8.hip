// Block-scaled fp8 (OCP MX: e4m3 elements, one E8M0 scale per 32 channels) path of the 3 x 3 stride-1 convolutions of the
// DoubleConv nets (BASELINE config 5: RGBDepthNet at 512 x 512): forward and input-gradient GEMMs on
// v_mfma_scale_f32_16x16x128_f8f6f4, outputs / epilogues in bf16 exactly as the bf16 patch kernel of igemm.hip.
//
//   adn_mx8_quantize   bf16 [rows][C]  ->  e4m3 [rows][C] + E8M0 [rows][C/32]           (activations, gradients)
//   adn_mx8_pack       f32 master [X][9][Y] -> e4m3 [rows][10][K] + E8M0 [rows][K/64][5][4]   (weights; 10th tap = 0)
//   adn_conv3x3_mx8    out[m][n] = sum_{tap, c} in[b, y+dy, x+dx, c] * w[n][tap][c]
//
// Operand map of the instruction (probed with exact integer data on the hardware, tools/probes/mx_*.hip -- it is NOT
// "32 consecutive k per lane"): lane l = (row/col l & 15, group g = l >> 4) holds 32 bytes; bytes 0..15 are k =
// 16 g + j of the first 64, bytes 16..31 are k = 64 + 16 g + j.  The 32-element scale blocks are therefore
//   block 0 = low halves of groups 0,1   block 1 = low halves of groups 2,3
//   block 2 = high halves of groups 0,1  block 3 = high halves of groups 2,3
// and the scale of block b is the scale byte supplied by lane group b (same row/col).
// That map fits the patch kernel's LDS image exactly: a K-step is 2 taps x 64 channels (64-byte fp8 pixels, four 16-byte
// chunks); lane group g reads chunk g of tap t0's pixel (low half) and chunk g of tap t0+1's pixel (high half): the two
// ds_read_b128 of the bf16 kernel's two MFMAs feed ONE MFMA of four times the K.  Blocks = (t0, ch 0..31), (t0, ch 32..63),
// (t0+1, ch 0..31), (t0+1, ch 32..63); lane group g supplies the scale of (tap t0 + (g >> 1), half g & 1).
// 9 taps = 4.5 tap pairs: the packed weights carry a 10th all-zero tap (the activation side re-reads tap 8's pixel).
//
// Staging (per workgroup = 8 x 16 output pixels of one image, 4 waves as 2 x 2, 64 x BN/2 each), per 64-channel chunk:
//   10 x 18 pixel patch (11.25 KiB, LDS-DMA 16 B/lane, chunk index XOR ((pixel >> 2) & 1) << 1 as in igemm.hip),
//   its 2 scale bytes per pixel (LDS-DMA buffer_load_ushort: the hardware writes one zero-extended DWORD per lane, probed
//   with tools/probes/lds_dma_u16.hip -- 64 pixels per instruction, LDS image [pixel][4 B]),
//   per step [2 taps][BN][64 B] weights (16 B/lane) and [BN][4] weight-scale bytes (4 B/lane).
#include <math.h>

#include "epilogue.h"
#include "mx8_quant.h"

namespace {

typedef __attribute__((ext_vector_type(8))) int v8i_t;

__device__ __forceinline__ int xcd_remap8(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// bf16 [rows][C] -> e4m3 + scales.  One thread per 8 channels (16-byte load, 8-byte store), 4 threads per block.
__global__ __launch_bounds__(256) void mx8_quant_kernel(const uint16_t* __restrict__ src, int64_t chunks, uint2* __restrict__ dst,
                                                        uint8_t* __restrict__ sc) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < chunks; i += stride) {   // chunks % 4 == 0: quads stay whole
    const u32x4_t raw = *reinterpret_cast<const u32x4_t*>(src + i * 8);
    float f[8];
    Chunk<uint16_t>::unpack(raw, f);
    int byte;
    const uint2 o = mx_quant8(f, byte);
    dst[i] = o;
    if ((threadIdx.x & 3) == 0) sc[i >> 2] = (uint8_t)byte;
  }
}

// weights: one thread per (row, tap 0..9, 32-block of the contraction channels)
//   transpose == 0: row = x (output channel), contraction = y:  v = master[row][tap][k]
//                   (threads run along the blocks of a row: each reads 128 contiguous bytes, neighbours adjacent)
//   transpose == 1: row = y (input channel), contraction = x, taps flipped:  v = master[k][8 - tap][row]
//                   (threads run along the rows: for every k the 64 lanes read 64 consecutive floats)
__global__ __launch_bounds__(256) void mx8_pack_kernel(const float* __restrict__ master, int X, int Y, int transpose,
                                                       uint8_t* __restrict__ w8, uint8_t* __restrict__ wsc) {
  const int rows = transpose ? Y : X, Kc = transpose ? X : Y;
  const int nblk = Kc >> 5;
  const int64_t total = (int64_t)rows * 10 * nblk;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  int blk, tap, row;
  if (transpose) {
    row = (int)(i % rows);
    blk = (int)((i / rows) % nblk);
    tap = (int)(i / ((int64_t)rows * nblk));
  } else {
    blk = (int)(i % nblk);
    tap = (int)((i / nblk) % 10);
    row = (int)(i / ((int64_t)nblk * 10));
  }
  float v[32];
  float am = 0.f;
  if (tap < 9) {
    if (transpose) {
      const float* src = master + ((int64_t)blk * 32 * 9 + (8 - tap)) * Y + row;
#pragma unroll
      for (int j = 0; j < 32; ++j) v[j] = src[(int64_t)j * 9 * Y];
    } else {
      const f32x4_t* src = reinterpret_cast<const f32x4_t*>(master + ((int64_t)row * 9 + tap) * Y + blk * 32);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const f32x4_t t = src[j];
        v[4 * j] = t[0];
        v[4 * j + 1] = t[1];
        v[4 * j + 2] = t[2];
        v[4 * j + 3] = t[3];
      }
    }
#pragma unroll
    for (int j = 0; j < 32; ++j) am = fmaxf(am, fabsf(v[j]));
  } else {
#pragma unroll
    for (int j = 0; j < 32; ++j) v[j] = 0.f;
  }
  const int byte = tap < 9 ? mx_scale_byte(am) : 127;
  u32x4_t* o = reinterpret_cast<u32x4_t*>(w8 + ((int64_t)row * 10 + tap) * Kc + blk * 32);
  u32x4_t q[2];
#pragma unroll
  for (int j = 0; j < 8; ++j)
    q[j >> 2][j & 3] = cvt4_e4m3(mx_descale(v[4 * j], byte), mx_descale(v[4 * j + 1], byte), mx_descale(v[4 * j + 2], byte),
                                 mx_descale(v[4 * j + 3], byte));
  o[0] = q[0];
  o[1] = q[1];
  // scale layout [row][chunk = blk >> 1][pair = tap >> 1][(tap & 1) * 2 + (blk & 1)]: the 4 bytes of a K-step are one dword
  wsc[(((int64_t)row * (nblk >> 1) + (blk >> 1)) * 5 + (tap >> 1)) * 4 + (tap & 1) * 2 + (blk & 1)] = (uint8_t)byte;
}

struct MxParams {
  const void* in0;
  const void* sc0;
  const void* in1;
  const void* sc1;
  const void* w;
  const void* wsc;
  int B, H, W, C0, C1, N;
  int tiles_m, tiles_n;
  int epi;
  AdnEpiSeg seg[2];
};

__device__ __forceinline__ void wait_all() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }

// TALL (64 output columns): 16 x 16-pixel tile, the 4 waves stacked along the pixels (64 x 64 each) instead of 2 x 2 waves
// of 64 x 32 -- 8 instead of 12 fragment reads per 16 MFMAs and no fragment read twice (see igemm_patch_kernel).
template <int BN, bool TALL = false>
__global__ __launch_bounds__(256, 2) void conv3x3_mx8_kernel(MxParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef uint16_t T;                                       // output / epilogue element type: bf16
  constexpr int TH = TALL ? 16 : 8, TW = 16;
  constexpr int BM = TH * TW, NWN = TALL ? 1 : 2, NTHR = 256;
  constexpr int WN = BN / NWN, NT = WN / 16, MT = 4;
  constexpr int MW = TW + 2, SEG_PIX = (TH + 2) * MW;       // 18 x 10 = 180 (tall: 18 x 18 = 324) patch pixels of 64 bytes
  constexpr int SEG_STEPS = 5;                              // tap pairs (0,1) (2,3) (4,5) (6,7) (8, zero tap)
  constexpr int PPIECES = (SEG_PIX + 15) / 16;              // 12 (21) one-KiB pieces
  constexpr int PK = (PPIECES + 3) / 4;                     // 3 (6) per wave
  constexpr int PBUF = PPIECES * 1024;
  constexpr int SPIECES = (SEG_PIX + 63) / 64;              // scale-patch DMA instructions of 64 pixels: 3 (6)
  constexpr int SK = (SPIECES + 3) / 4;                     // per wave: 1 (2)
  constexpr int ASBUF = SPIECES * 256;                      // one dword per pixel (2 scale bytes, zero-extended)
  constexpr int BPT = BN / 16;                              // weight pieces per tap
  constexpr int BK_ = 2 * BPT / 4;                          // weight pieces per wave and step
  constexpr int BBUF = 2 * BN * 64;
  constexpr int BSBUF = BN * 4;
  constexpr int LDC = BN + 4;
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Pl = smem;                          // [2][PBUF]
  char* Bl = Pl + 2 * PBUF;                 // [2][BBUF]
  char* ASl = Bl + 2 * BBUF;                // [2][ASBUF]
  char* BSl = ASl + 2 * ASBUF;              // [2][BSBUF]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = TALL ? wave : (wave >> 1), wn = TALL ? 0 : (wave & 1);
  const int nwg = p.tiles_m * p.tiles_n;
  const int wg = xcd_remap8(blockIdx.x, nwg);
  const int tile_n = wg % p.tiles_n;
  const int tile_m = wg / p.tiles_n;
  const int H = p.H, W = p.W;
  const int Cin = p.C0 + p.C1;
  const int tpr = W / TW, tpi = (H / TH) * tpr;
  const int tb = tile_m / tpi, trem = tile_m - tb * tpi;
  const int oy0 = (trem / tpr) * TH, ox0 = (trem % tpr) * TW;

  // ---- patch loader geometry: piece wave + 4k = LDS pixels 16 pi .. 16 pi + 15, lane -> pixel +(lane >> 2), chunk lane & 3
  unsigned ppix[PK];
  unsigned pmask = 0;
  const int bshift = W + 1;
#pragma unroll
  for (int k = 0; k < PK; ++k) {
    const int q = 16 * (wave + 4 * k) + (lane >> 2);
    const int hr = q / MW, m = q - hr * MW;
    const int iy = oy0 + hr - 1, ix = ox0 + m - 1;
    const bool ok = (unsigned)ix < (unsigned)W && (unsigned)iy < (unsigned)H && q < SEG_PIX;
    ppix[k] = (unsigned)((tb * H + iy) * W + ix + bshift);
    pmask |= (ok ? 1u : 0u) << k;
  }
  // scale patch: DMA instruction wave + 4k covers LDS pixels 64 (wave + 4k) .. + 63, lane -> its pixel (2 bytes)
  unsigned spix[SK];
  unsigned smask = 0;
#pragma unroll
  for (int k = 0; k < SK; ++k) {
    const int q = 64 * (wave + 4 * k) + lane;
    const int hr = q / MW, m = q - hr * MW;
    const int iy = oy0 + hr - 1, ix = ox0 + m - 1;
    const bool ok = (unsigned)ix < (unsigned)W && (unsigned)iy < (unsigned)H && q < SEG_PIX;
    spix[k] = (unsigned)((tb * H + iy) * W + ix + bshift);
    smask |= (ok ? 1u : 0u) << k;
  }
  unsigned bvo[BK_];
  const int ktot = 10 * Cin;
#pragma unroll
  for (int k = 0; k < BK_; ++k) {
    const int pid = wave + 4 * k;
    const int tsel = pid / BPT, row = (pid % BPT) * 16 + (lane >> 2);
    const int lc = (lane & 3) ^ (((row >> 2) & 1) << 1);
    bvo[k] = (unsigned)((tile_n * BN + row) * ktot + tsel * Cin + lc * 16);
  }
  const int nchunks = Cin >> 6;
  const int nsteps = nchunks * SEG_STEPS;
  // weight scales: [N][nchunks][5][4] bytes = one dword per (row, step); waves 0 .. BN/64-1, lane -> row 64 wave + lane
  const unsigned bsvo = (unsigned)((tile_n * BN + 64 * wave + lane) * nsteps * 4);

  typedef __attribute__((address_space(3))) void* lptr_t;
  constexpr unsigned REC = 0x7ffffff0u;
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const char*>(p.in0) - (int64_t)bshift * p.C0), 0, REC, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const char*>(p.C1 ? p.in1 : p.in0) - (int64_t)bshift * p.C1), 0, REC, 0x00020000);
  const __amdgpu_buffer_rsrc_t rq0 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const char*>(p.sc0) - (int64_t)bshift * (p.C0 >> 5)), 0, REC, 0x00020000);
  const __amdgpu_buffer_rsrc_t rq1 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const char*>(p.C1 ? p.sc1 : p.sc0) - (int64_t)bshift * (p.C1 >> 5)), 0, REC, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, REC, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsq = __builtin_amdgcn_make_buffer_rsrc((void*)p.wsc, 0, REC, 0x00020000);

  // patch pieces of chunk sg: piece k goes out at step-in-segment k (k < 3); the scale bytes at step 3
  auto issue_patch = [&](int sg, int ss) {
    const int c0 = sg << 6;
    const bool second = c0 >= p.C0;
    const int Cs = second ? p.C1 : p.C0;
    const int coff = second ? c0 - p.C0 : c0;
    char* dst = Pl + (sg & 1) * PBUF + wave * 1024;
#pragma unroll
    for (int k = 0; k < PK; ++k) {
      if (k % SEG_STEPS != ss) continue;
      if (wave + 4 * k >= PPIECES) continue;
      const bool ok = (pmask >> k) & 1u;
      const unsigned lc16 = (unsigned)(((lane & 3) ^ (((lane >> 4) & 1) << 1)) << 4);
      const unsigned vo = ok ? ppix[k] * (unsigned)Cs + lc16 : OOB;
      if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lptr_t)(dst + k * 4096), 16, vo, coff, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (lptr_t)(dst + k * 4096), 16, vo, coff, 0, 0);
    }
    if (ss == 3) {
#pragma unroll
      for (int k = 0; k < SK; ++k) {
        if (wave + 4 * k >= SPIECES) continue;
        const unsigned vo = ((smask >> k) & 1u) ? spix[k] * (unsigned)(Cs >> 5) : OOB;
        char* sd = ASl + (sg & 1) * ASBUF + (wave + 4 * k) * 256;
        if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rq1, (lptr_t)sd, 2, vo, coff >> 5, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rq0, (lptr_t)sd, 2, vo, coff >> 5, 0, 0);
      }
    }
  };
  auto issue_b = [&](int s) {
    const int sg = s / SEG_STEPS, ss = s - sg * SEG_STEPS;
    const int soff = 2 * ss * Cin + (sg << 6);
    char* dst = Bl + (s & 1) * BBUF + wave * 1024;
#pragma unroll
    for (int k = 0; k < BK_; ++k) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lptr_t)(dst + k * 4096), 16, bvo[k], soff, 0, 0);
    if (wave < BN / 64)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsq, (lptr_t)(BSl + (s & 1) * BSBUF + wave * 256), 4, bsvo, s * 4, 0, 0);
  };

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: the whole first patch + scales, weights of step 0 ----
#pragma unroll
  for (int ss = 0; ss < SEG_STEPS; ++ss) issue_patch(0, ss);
  issue_b(0);

  const int frow = lane & 15, fq = lane >> 4;
  const int qb0 = wm * 4 * MW + frow;
  const int nseg = nchunks;
  int s = 0;
  for (int sg = 0; sg < nseg; ++sg) {
    const char* Pb = Pl + (sg & 1) * PBUF;
    const char* ASb = ASl + (sg & 1) * ASBUF;
#pragma unroll
    for (int ss = 0; ss < SEG_STEPS; ++ss, ++s) {
      wait_all();                                    // own DMA landed, own LDS reads of the previous step returned
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (s + 1 < nsteps) issue_b(s + 1);
      if (sg + 1 < nseg) issue_patch(sg + 1, ss);
      const char* Bb = Bl + (s & 1) * BBUF;
      const char* BSb = BSl + (s & 1) * BSBUF;
      // taps of the pair: t0 = 2 ss, t1 = 2 ss + 1 (tap 9 does not exist: zero weights, re-read tap 8's pixel)
      const int t0 = 2 * ss, t1 = ss == 4 ? 8 : 2 * ss + 1;
      const int qo0 = (t0 / 3) * MW + (t0 % 3), qo1 = (t1 / 3) * MW + (t1 % 3);
      // (the fragment addresses are recomputed per step from an opaque copy of the lane's base pixel: left to itself the
      //  compiler hoists the 5 x 12 swizzled addresses of the unrolled steps out of the chunk loop and spills ~1 KB/lane)
      int qv = qb0;
      asm volatile("" : "+v"(qv));
      const int qos = (fq >> 1) ? qo1 : qo0;           // the tap whose scale this lane group supplies
      v8i_t af[MT], bf[NT];
      int as[MT], bs[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int qa = qv + i * MW + qo0, qb = qv + i * MW + qo1;
        const u32x4_t lo = *reinterpret_cast<const u32x4_t*>(Pb + qa * 64 + ((fq ^ (((qa >> 2) & 1) << 1)) << 4));
        const u32x4_t hi = *reinterpret_cast<const u32x4_t*>(Pb + qb * 64 + ((fq ^ (((qb >> 2) & 1) << 1)) << 4));
        af[i] = v8i_t{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
        as[i] = *reinterpret_cast<const uint8_t*>(ASb + (qv + i * MW + qos) * 4 + (fq & 1));
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int row = wn * WN + j * 16 + frow;
        const int sw = (fq ^ (((row >> 2) & 1) << 1)) << 4;
        const u32x4_t lo = *reinterpret_cast<const u32x4_t*>(Bb + row * 64 + sw);
        const u32x4_t hi = *reinterpret_cast<const u32x4_t*>(Bb + BN * 64 + row * 64 + sw);
        bf[j] = v8i_t{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
        bs[j] = *reinterpret_cast<const uint8_t*>(BSb + row * 4 + fq);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af[i], bf[j], acc[i][j], 0, 0, 0, as[i], 0, bs[j]);
      // keep the step self-contained: the MFMA intrinsic has no side effects, so the IR sink pass otherwise moves the
      // MFMAs of all five steps behind the last step's loads (each step is its own basic block) and every fragment of the
      // chunk is spilled to scratch (752 B/lane).  The empty asm "uses" the accumulators here.  (The matrix pipe still
      // overlaps the next step's DMA issue and LDS reads in hardware, and the CU's second workgroup fills the gaps.)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) asm volatile("" : "+v"(acc[i][j]));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();

  // ---- epilogue through LDS (as igemm_patch_kernel; tile row = oyl * 16 + oxl) ----
  float* ct = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        ct[(wm * 64 + i * 16 + 4 * fq + r) * LDC + wn * WN + j * 16 + frow] = acc[i][j][r];
  __syncthreads();
  constexpr int CPR = BN / 8, RSTEP = NTHR / CPR, RPT = BM / RSTEP;
  const int cg = tid % CPR, rsub = tid / CPR;
  const int n0 = tile_n * BN + cg * 8;
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
    const int64_t op = ((int64_t)tb * H + oy0 + (row >> 4)) * W + ox0 + (row & 15);
    float v[8];
    const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(ct + row * LDC + cg * 8);
    const f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(ct + row * LDC + cg * 8 + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = v0[e];
      v[4 + e] = v1[e];
    }
    epi_vec8<T>(epi, sg2, cols, op, nl, v, s1, s2);
  }
  if ((p.seg[0].partials != nullptr || p.seg[1].partials != nullptr) && (epi == ADN_EPI_Z_STATS || epi == ADN_EPI_BWD)) {
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
      if (sq.partials) sq.partials[((int64_t)tile_m * 2 + st) * sq.channels + ncl] = t;
    }
  }
#endif
}

template <int BN, bool TALL = false>
void launch_mx8(const MxParams& kp, hipStream_t st) {
  constexpr int TH = TALL ? 16 : 8;
  constexpr int segpix = (TH + 2) * 18;
  constexpr int stage = 2 * ((segpix + 15) / 16) * 1024 + 2 * (2 * BN * 64) + 2 * ((segpix + 63) / 64) * 256 + 2 * BN * 4;
  constexpr int epil = TH * 16 * (BN + 4) * 4;
  constexpr int lds = stage > epil ? stage : epil;
  ADN_SET_LDS_ONCE(lds, &conv3x3_mx8_kernel<BN, TALL>);
  hipLaunchKernelGGL((conv3x3_mx8_kernel<BN, TALL>), dim3(kp.tiles_m * kp.tiles_n), dim3(256), lds, st, kp);
}

// Output columns per workgroup (ADN_MX8_BN=128 restores the round-2 rule "128 whenever N % 128 == 0" for an A/B; 64 forces
// the 64-column forms).  As for the bf16 kernel (igemm.hip make_plan) the tall 256 x 64 tile beats 128 x 128 on the wide
// layers -- here from 256 output columns on: d3 conv2 92.8 -> 88.5 us (1 747 TFLOP/s), d2 conv2 100.9 -> 99.5; at 128
// columns it is neutral (forward) to 5 % slower (input gradient), so those keep 128 -- and 64 columns give two workgroups
// per CU where 128 give one (d4 conv2 32.8 -> 31.5 us).
int mx8_bn(const AdnMx8ConvDesc* d) {
  static const int forced = getenv("ADN_MX8_BN") ? atoi(getenv("ADN_MX8_BN")) : 0;
  if (d->N % 128 != 0) return 64;
  if (forced == 128) return 128;
  const int64_t pix = (int64_t)d->B * d->H * d->W;
  const bool tall_ok = d->N >= 256 && d->H % 16 == 0 && pix / 256 * (d->N / 64) >= 512;
  const int64_t t128 = pix / 128 * (d->N / 128);
  if (forced == 64) return 64;
  return (tall_ok || (t128 >= 256 && t128 < 512)) ? 64 : 128;
}
// rows per workgroup: 256 (tall form) for 64-column tiles whose images tile by 16 x 16 and still fill the chip
int mx8_rows(const AdnMx8ConvDesc* d) {
  const bool tall = mx8_bn(d) == 64 && d->H % 16 == 0 && (int64_t)d->B * d->H * d->W / 256 * (d->N / 64) >= 512;
  return tall ? 256 : 128;
}

int validate(const AdnMx8ConvDesc* d) {
  ADN_CHECK_ARG(d != nullptr, "adn_conv3x3_mx8: null descriptor");
  ADN_CHECK_ARG(d->B > 0 && d->H > 0 && d->W > 0 && d->H % 8 == 0 && d->W % 16 == 0,
                "adn_conv3x3_mx8: the image must tile by 8 x 16 output pixels (B=%d H=%d W=%d)", d->B, d->H, d->W);
  ADN_CHECK_ARG(d->C0 > 0 && d->C0 % 64 == 0 && d->C1 >= 0 && d->C1 % 64 == 0 && d->N > 0 && d->N % 64 == 0,
                "adn_conv3x3_mx8: channels must be multiples of 64 (C0=%d C1=%d N=%d)", d->C0, d->C1, d->N);
  ADN_CHECK_ARG(d->in0 && d->sc0 && d->w && d->wsc && (d->C1 == 0 || (d->in1 && d->sc1)), "adn_conv3x3_mx8: null operand");
  ADN_CHECK_ARG(d->epi == ADN_EPI_Z_STATS || d->epi == ADN_EPI_ACT || d->epi == ADN_EPI_BWD || d->epi == ADN_EPI_ADD,
                "adn_conv3x3_mx8: epilogue %d not supported", d->epi);
  ADN_CHECK_ARG(d->seg[0].channels + d->seg[1].channels == d->N && d->seg[0].channels > 0 && d->seg[0].channels % 64 == 0 &&
                    d->seg[1].channels >= 0 && d->seg[1].channels % 64 == 0,
                "adn_conv3x3_mx8: segment channels %d+%d != N=%d", d->seg[0].channels, d->seg[1].channels, d->N);
  for (int s = 0; s < 2; ++s) {
    if (d->seg[s].channels == 0) continue;
    const AdnEpiSeg& g = d->seg[s];
    if (d->epi != ADN_EPI_ACT) ADN_CHECK_ARG(g.out0, "adn_conv3x3_mx8: seg %d out0 is null", s);
    if (d->epi == ADN_EPI_ACT) ADN_CHECK_ARG(g.out0 || g.out1, "adn_conv3x3_mx8: seg %d has no output", s);
    if (d->epi == ADN_EPI_BWD) {
      ADN_CHECK_ARG(g.ref, "adn_conv3x3_mx8: seg %d BWD needs ref", s);
      if (g.partials) ADN_CHECK_ARG(g.z && g.mean && g.istd, "adn_conv3x3_mx8: seg %d BWD stats need z/mean/istd", s);
    }
  }
  const int64_t pix = (int64_t)d->B * d->H * d->W;
  ADN_CHECK_ARG(pix * (d->C0 > d->C1 ? d->C0 : d->C1) < 0x7ff00000ll && pix < (1ll << 29),
                "adn_conv3x3_mx8: a gathered source exceeds 2 GiB (B=%d %dx%d C=%d/%d)", d->B, d->H, d->W, d->C0, d->C1);
  return ADN_OK;
}

}  // namespace

extern "C" int adn_mx8_quantize(const void* src, int64_t rows, int32_t C, void* dst, void* scales, void* stream) {
  ADN_CHECK_ARG(src && dst && scales && rows > 0 && C > 0 && C % 32 == 0 && (rows * C) % 128 == 0,
                "adn_mx8_quantize: rows=%lld C=%d (C %% 32, rows*C %% 128 must be 0)", (long long)rows, C);
  const int64_t chunks = rows * C / 8;
  int64_t blocks = adn_cdiv(chunks, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(mx8_quant_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const uint16_t*>(src), chunks, reinterpret_cast<uint2*>(dst),
                     reinterpret_cast<uint8_t*>(scales));
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_mx8_pack(const float* master, int32_t X, int32_t Y, int32_t transpose, void* w8, void* wsc, void* stream) {
  ADN_CHECK_ARG(master && w8 && wsc && X > 0 && Y > 0, "adn_mx8_pack: null operand / bad shape");
  const int Kc = transpose ? X : Y, rows = transpose ? Y : X;
  ADN_CHECK_ARG(Kc % 64 == 0, "adn_mx8_pack: contraction channels %d must be a multiple of 64", Kc);
  const int64_t total = (int64_t)rows * 10 * (Kc / 32);
  hipLaunchKernelGGL(mx8_pack_kernel, dim3((unsigned)adn_cdiv(total, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     master, X, Y, transpose, reinterpret_cast<uint8_t*>(w8), reinterpret_cast<uint8_t*>(wsc));
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int64_t adn_conv3x3_mx8_num_partials(const AdnMx8ConvDesc* d) {
  if (validate(d) != ADN_OK) return -1;
  return (int64_t)d->B * d->H * d->W / mx8_rows(d);
}

extern "C" int adn_conv3x3_mx8(const AdnMx8ConvDesc* d, void* stream) {
  const int rc = validate(d);
  if (rc != ADN_OK) return rc;
  MxParams kp;
  kp.in0 = d->in0;
  kp.sc0 = d->sc0;
  kp.in1 = d->in1;
  kp.sc1 = d->sc1;
  kp.w = d->w;
  kp.wsc = d->wsc;
  kp.B = d->B;
  kp.H = d->H;
  kp.W = d->W;
  kp.C0 = d->C0;
  kp.C1 = d->C1;
  kp.N = d->N;
  kp.epi = d->epi;
  kp.seg[0] = d->seg[0];
  kp.seg[1] = d->seg[1];
  const int bn = mx8_bn(d);
  const int rows = mx8_rows(d);
  kp.tiles_m = (int)((int64_t)d->B * d->H * d->W / rows);
  kp.tiles_n = d->N / bn;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (bn == 128) launch_mx8<128>(kp, st);
  else if (rows == 256) launch_mx8<64, true>(kp, st);
  else launch_mx8<64>(kp, st);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}
