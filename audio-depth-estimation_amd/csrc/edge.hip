// Thin outermost layers of the U-Net generator on the bf16 path (gfx950): the first Conv2d(k4,s2,p1) with 2 input
// channels and the last ConvTranspose2d(k4,s2,p1) with 1 output channel (/root/reference/models/unetbaseline_model.py
// :187-198).  With K = 32 (16 taps x 2 channels) resp. N = 1 these layers do 0.5 % of the step's FLOPs but stream the
// two widest activations of the network (B x 128 x 128 x 64 each way), so every kernel here is HBM-bound: one MFMA per
// 16 pixels, everything else is about moving whole 64-byte / 128-byte pieces of NHWC rows once.
//
//   l0_fwd        out[b,oy,ox,:64]  = leaky / relu ( sum_{ky,kx,ci} x[b,ci,2oy-1+ky,2ox-1+kx] * w[o][ky][kx][ci] )
//   d0_dgrad      G[b,i,j,c]        = mask * sum_{ky,kx} dz[b,2i-1+ky,2j-1+kx] * w[c][ky][kx]      c in skip(64) ++ up(64)
//                 + BatchNorm-backward statistics of the up half (sum g, sum g*xhat)
//   thin_wgrad    dW[c][tap*CT+ct]  = sum_{b,i,j} plain[b,i,j,c] * thin[b,ct,2i-1+ky,2j-1+kx]
//                 (CT = 1: weight gradient of the last transposed conv, plain = its 128 input channels, thin = dz;
//                  CT = 2: weight gradient of the first conv, plain = its output gradient, thin = the network input)
//
// The thin operand is read as planar f32 ([B][CT][2Hs][2Ws]: the network input itself, resp. the gradient of the
// 1-channel output) and rounded to bf16 in registers, which is what the padded-channel MFMA path stores.
//
// Transposed-output trick (l0_fwd, d0_dgrad): the MFMA computes out^T = W * window^T, so a lane ends up with one
// PIXEL and 4 consecutive accumulator rows = 4 channels per tile; the weight rows are permuted such that the four tiles
// of a 64-channel segment give lane (pixel n, group q) channels q*8..q*8+7 and 32+q*8..32+q*8+7: two 16-byte pieces,
// the four lanes of a pixel together write 64 contiguous bytes per store instruction.
#include "epilogue.h"

namespace {

typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(1))) void* gptr_t;

__device__ __forceinline__ uint32_t pk_bf16(float lo, float hi) {
  return (uint32_t)f32_to_bf16_bits(lo) | ((uint32_t)f32_to_bf16_bits(hi) << 16);
}
__device__ __forceinline__ bf16x8_t as_frag(const u32x4_t& c) { return *reinterpret_cast<const bf16x8_t*>(&c); }

// tile t (0..3 of a 64-channel segment), MFMA row a (0..15) -> channel inside the segment
__device__ __forceinline__ int edge_ch(int t, int a) { return (t >> 1) * 32 + (a >> 2) * 8 + (t & 1) * 4 + (a & 3); }

// Columns 2j-1 .. 2j+2 of one row of a planar f32 image: four independent dword loads (clamped addresses, zero by
// select: no branch, so the loads of all rows of an iteration are in flight together; neighbouring lanes share the
// cache lines).  rowp must point at a readable row even when row_ok is false.
__device__ __forceinline__ void window4(const float* rowp, bool row_ok, int j, int Wl, float* w4) {
  const int c = 2 * j - 1;
  const float v0 = rowp[c < 0 ? 0 : c];
  const float v1 = rowp[c + 1];
  const float v2 = rowp[c + 2];
  const float v3 = rowp[c + 3 < Wl ? c + 3 : Wl - 1];
  w4[0] = (row_ok && c >= 0) ? v0 : 0.f;
  w4[1] = row_ok ? v1 : 0.f;
  w4[2] = row_ok ? v2 : 0.f;
  w4[3] = (row_ok && c + 3 < Wl) ? v3 : 0.f;
}

__device__ __forceinline__ void store_bf16x8(uint16_t* p, const float* f) {
  *reinterpret_cast<u32x4_t*>(p) = Chunk<uint16_t>::pack(f);
}
__device__ __forceinline__ void load_bf16x8(const uint16_t* p, float* f) {
  Chunk<uint16_t>::unpack(*reinterpret_cast<const u32x4_t*>(p), f);
}

// ---------------------------------------------------------------------------------------------------------------------
// First conv, forward.  x [B][2][2Hs][2Ws] f32, w [64][16][2] f32 (the parameter's channels_last memory), outputs
// [B][Hs][Ws][64] bf16: leaky (operand of the next conv) and relu (skip operand of the last transposed conv).
__global__ __launch_bounds__(256) void l0_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, int B,
                                                     int Hs, int Ws, float slope, uint16_t* out_leaky,
                                                     uint16_t* out_relu) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 15, q = lane >> 4;
  const int Hl = 2 * Hs, Wl = 2 * Ws;
  bf16x8_t wf[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float* wp = w + edge_ch(t, n) * 32 + q * 8;          // k = ky*8 + kx*2 + ci, this lane: ky = q
    const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(wp), hi = *reinterpret_cast<const f32x4_t*>(wp + 4);
    const u32x4_t c = {pk_bf16(lo[0], lo[1]), pk_bf16(lo[2], lo[3]), pk_bf16(hi[0], hi[1]), pk_bf16(hi[2], hi[3])};
    wf[t] = as_frag(c);
  }
  // A wave's work unit is a run of 4 pixel groups of one output row: the row decode (32-bit divisions) and the row
  // pointers are per unit, the inner loop over the 16-pixel groups only advances by constants (the kernel is VALU-issue bound, not HBM bound, when
  // every group pays its own index arithmetic: 300 instructions per 16 pixels).
  const int gpr = Ws >> 4;
  const int gu = (gpr & 3) == 0 ? 4 : 1;                     // groups per work unit (a run of 64 pixels of one row)
  const unsigned upr = (unsigned)(gpr / gu);
  const unsigned units = (unsigned)B * Hs * upr;
  for (unsigned u = blockIdx.x * 4 + wave; u < units; u += gridDim.x * 4) {
    const unsigned bi = u / upr;
    const int jg0 = (int)(u - bi * upr) * gu;
    const int b = (int)(bi / (unsigned)Hs);
    const int oy = (int)(bi - (unsigned)b * Hs);
   for (int jg = jg0; jg < jg0 + gu; ++jg) {
    const int ox = jg * 16 + n;
    const int iy = 2 * oy - 1 + q;
    const bool rok = (unsigned)iy < (unsigned)Hl;
    const float* r0 = x + (((int64_t)b * 2) * Hl + (rok ? iy : 0)) * Wl;
    const float* r1 = r0 + (int64_t)Hl * Wl;
    float a0[4], a1[4];
    window4(r0, rok, ox, Wl, a0);
    window4(r1, rok, ox, Wl, a1);
    u32x4_t bc;
#pragma unroll
    for (int kx = 0; kx < 4; ++kx) bc[kx] = pk_bf16(a0[kx], a1[kx]);
    const bf16x8_t bfr = as_frag(bc);
    f32x4_t acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t], bfr, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    const int64_t pix = ((int64_t)b * Hs + oy) * Ws + ox;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float v[8], o[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = acc[2 * h][e];
        v[4 + e] = acc[2 * h + 1][e];
      }
      const int64_t idx = pix * 64 + h * 32 + q * 8;
      if (out_leaky) {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = v[e] > 0.f ? v[e] : v[e] * slope;
        store_bf16x8(out_leaky + idx, o);
      }
      if (out_relu) {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = fmaxf(v[e], 0.f);
        store_bf16x8(out_relu + idx, o);
      }
    }
   }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Last transposed conv, input gradient.  dz [B][2Hs][2Ws] f32, w [128][16] f32 ([Cin][kh][kw][1]); segment 0 = skip
// half, segment 1 = up half (ReLU mask by `ref`, optional BatchNorm-backward statistics of the segment).
struct D0Params {
  const float* dz;
  const float* w;
  int B, Hs, Ws;
  AdnEpiSeg seg[2];
};

__global__ __launch_bounds__(256) void d0_dgrad_kernel(D0Params p) {
  __shared__ float red[4][2][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 15, q = lane >> 4;
  const int Hs = p.Hs, Ws = p.Ws, Hl = 2 * Hs, Wl = 2 * Ws;
  bf16x8_t wf[2][4];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      // k = tap = 8q + j; k >= 16 (q >= 2) is padding: loaded from a valid address and zeroed by select (no branch:
      // the 8 fragment loads of a wave's prologue stay in flight together)
      const float* wp = p.w + (s * 64 + edge_ch(t, n)) * 16 + (q & 1) * 8;
      const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(wp), hi = *reinterpret_cast<const f32x4_t*>(wp + 4);
      u32x4_t c = {pk_bf16(lo[0], lo[1]), pk_bf16(lo[2], lo[3]), pk_bf16(hi[0], hi[1]), pk_bf16(hi[2], hi[3])};
      if (q >= 2) c = u32x4_t{0u, 0u, 0u, 0u};
      wf[s][t] = as_frag(c);
    }
  // BatchNorm-backward statistics of segment 1 (the up half: the skip half of the outermost level has no BatchNorm):
  // this lane's 16 channels (h*32 + q*8 + e), accumulated over its pixels
  const bool stats = p.seg[1].partials != nullptr;
  float mean[16], istd[16], s1[16], s2[16];
  {
    const float* mp = stats ? p.seg[1].mean : p.w;      // any readable 64 floats when there are no statistics
    const float* ip = stats ? p.seg[1].istd : p.w;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int v4 = 0; v4 < 2; ++v4) {
        const f32x4_t m4 = *reinterpret_cast<const f32x4_t*>(mp + h * 32 + q * 8 + v4 * 4);
        const f32x4_t i4 = *reinterpret_cast<const f32x4_t*>(ip + h * 32 + q * 8 + v4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          mean[h * 8 + v4 * 4 + e] = m4[e];
          istd[h * 8 + v4 * 4 + e] = i4[e];
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) s1[e] = s2[e] = 0.f;
  }
  const int gpr = Ws >> 4;
  const int gu = (gpr & 3) == 0 ? 4 : 1;                     // work unit = 4 pixel groups of one row (see l0_fwd)
  const unsigned upr = (unsigned)(gpr / gu);
  const unsigned units = (unsigned)p.B * Hs * upr;
  for (unsigned u = blockIdx.x * 4 + wave; u < units; u += gridDim.x * 4) {
    const unsigned bi = u / upr;
    const int jg0 = (int)(u - bi * upr) * gu;
    const int b = (int)(bi / (unsigned)Hs);
    const int i = (int)(bi - (unsigned)b * Hs);
   for (int jg = jg0; jg < jg0 + gu; ++jg) {
    const int j = jg * 16 + n;
    // B fragment: k = 8q + e, e = rr*4 + kx -> tap (ky = 2q + rr, kx): rows 2i-1+2q+rr
    u32x4_t bc = {0u, 0u, 0u, 0u};
    {
      const int iy0 = 2 * i - 1 + 2 * q;
      float w0[4], w1[4];
      const bool ok0 = q < 2 && (unsigned)iy0 < (unsigned)Hl;
      const bool ok1 = q < 2 && (unsigned)(iy0 + 1) < (unsigned)Hl;
      const float* base = p.dz + (int64_t)b * Hl * Wl;
      window4(base + (int64_t)(ok0 ? iy0 : 0) * Wl, ok0, j, Wl, w0);
      window4(base + (int64_t)(ok1 ? iy0 + 1 : 0) * Wl, ok1, j, Wl, w1);
      bc[0] = pk_bf16(w0[0], w0[1]);
      bc[1] = pk_bf16(w0[2], w0[3]);
      bc[2] = pk_bf16(w1[0], w1[1]);
      bc[3] = pk_bf16(w1[2], w1[3]);
    }
    const bf16x8_t bfr = as_frag(bc);
    const int64_t pix = ((int64_t)b * Hs + i) * Ws + j;
    // every epilogue operand of this pixel group is requested before the first use (6 x 16 bytes per lane in flight)
    u32x4_t rraw[2][2], zraw[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        rraw[s][h] = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const uint16_t*>(p.seg[s].ref) + pix * 64 + h * 32 +
                                                       q * 8);
    {
      const uint16_t* zp = reinterpret_cast<const uint16_t*>(stats ? p.seg[1].z : p.seg[1].ref);
#pragma unroll
      for (int h = 0; h < 2; ++h) zraw[h] = *reinterpret_cast<const u32x4_t*>(zp + pix * 64 + h * 32 + q * 8);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const AdnEpiSeg& sg = p.seg[s];
      f32x4_t acc[4];
#pragma unroll
      for (int t = 0; t < 4; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s][t], bfr, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int64_t idx = pix * 64 + h * 32 + q * 8;
        float r[8], gq[8];
        Chunk<uint16_t>::unpack(rraw[s][h], r);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          gq[e] = acc[2 * h][e];
          gq[4 + e] = acc[2 * h + 1][e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) gq[e] *= (r[e] > 0.f ? 1.0f : sg.slope);
        store_bf16x8(reinterpret_cast<uint16_t*>(sg.out0) + idx, gq);
        if (s == 1) {
          float z[8];
          Chunk<uint16_t>::unpack(zraw[h], z);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            s1[h * 8 + e] += gq[e];
            s2[h * 8 + e] += gq[e] * ((z[e] - mean[h * 8 + e]) * istd[h * 8 + e]);
          }
        }
      }
    }
   }
  }
  // statistics: sum over the 16 pixels of a lane row, then over the 4 waves, one partial row per workgroup
  if (stats) {                                   // kernel-uniform
#pragma unroll
    for (int e = 0; e < 16; ++e) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        s1[e] += __shfl_xor(s1[e], o, 64);
        s2[e] += __shfl_xor(s2[e], o, 64);
      }
    }
    if (n == 0) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int c = (e >> 3) * 32 + q * 8 + (e & 7);
        red[wave][0][c] = s1[e];
        red[wave][1][c] = s2[e];
      }
    }
    __syncthreads();
    if (threadIdx.x < 128) {
      const int st = threadIdx.x >> 6, c = threadIdx.x & 63;
      const float t = (red[0][st][c] + red[1][st][c]) + (red[2][st][c] + red[3][st][c]);
      p.seg[1].partials[((int64_t)blockIdx.x * 2 + st) * 64 + c] = t;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient against a thin gathered operand.  One K-step = 32 consecutive pixels of one row of the plain
// (low-resolution, NHWC bf16) tensor(s); the contraction runs over pixels, so the plain tile goes through LDS
// (LDS-DMA, wave-private double buffer: no workgroup barrier in the loop) and is read back transposed with
// ds_read_b64_tr_b16; the thin operand's (tap, ct) x pixel fragment is gathered straight from the planar f32 image.
struct TWParams {
  const float* thin;
  const uint16_t* plain0;
  const uint16_t* plain1;
  int B, Hs, Ws;
  float* slab;       // [blocks][16*CT][C0+C1]
};

template <int CT, int NT0, int NT1>
__global__ __launch_bounds__(256) void thin_wgrad_kernel(TWParams p) {
  constexpr int NT = NT0 + NT1;
  constexpr int RB0 = NT0 * 32, RB1 = NT1 * 32;          // tile row bytes of the two plain sources
  constexpr int SUB0 = 32 * RB0, SUB1 = 32 * RB1;
  constexpr int BUF = SUB0 + SUB1;
  constexpr int CTOT = NT * 16, ROWS = 16 * CT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, q = lane >> 4;
  const int Hs = p.Hs, Ws = p.Ws, Hl = 2 * Hs, Wl = 2 * Ws;
  char* wbase = smem + wave * (2 * BUF);
  const unsigned spr = (unsigned)Ws >> 5;                    // K-steps per image row
  const unsigned steps = (unsigned)p.B * Hs * spr;          // (32-bit index arithmetic: no 64-bit divisions)
  const unsigned stride = gridDim.x * 4;

  auto dma = [&](unsigned s, int buf) {
    const int64_t pix0 = (int64_t)s * 32;                  // rows are contiguous: step s starts at pixel 32*s
    char* dst = wbase + buf * BUF;
    if constexpr (NT0 > 0) {
      constexpr int LPR = RB0 / 16, RPI = 64 / LPR, NI = 32 / RPI;
      const char* src = reinterpret_cast<const char*>(p.plain0) + (pix0 + lane / LPR) * RB0 + (lane % LPR) * 16;
#pragma unroll
      for (int k = 0; k < NI; ++k)
        __builtin_amdgcn_global_load_lds((gptr_t)(src + (int64_t)k * RPI * RB0), (lptr_t)(dst + k * 1024), 16, 0, 0);
    }
    if constexpr (NT1 > 0) {
      constexpr int LPR = RB1 / 16, RPI = 64 / LPR, NI = 32 / RPI;
      const char* src = reinterpret_cast<const char*>(p.plain1) + (pix0 + lane / LPR) * RB1 + (lane % LPR) * 16;
#pragma unroll
      for (int k = 0; k < NI; ++k)
        __builtin_amdgcn_global_load_lds((gptr_t)(src + (int64_t)k * RPI * RB1), (lptr_t)(dst + SUB0 + k * 1024), 16, 0,
                                         0);
    }
  };
  // thin fragment of M-tile m: lane (a = n, q) holds row 16m + a = (tap, ct), pixels j0 + 8q + jj (jj = 0..7)
  auto load_thin = [&](unsigned s, float (*av)[8]) {
    const unsigned bi = s / spr;
    const int js = (int)(s - bi * spr);
    const int b = (int)(bi / (unsigned)Hs);
    const int i = (int)(bi - (unsigned)b * Hs);
#pragma unroll
    for (int m = 0; m < CT; ++m) {
      const int row = 16 * m + n;
      const int tap = row / CT, ct = row % CT;
      const int iy = 2 * i - 1 + (tap >> 2);
      const bool rok = (unsigned)iy < (unsigned)Hl;
      const float* rp = p.thin + (((int64_t)b * CT + ct) * Hl + (rok ? iy : 0)) * Wl;
      const int c0 = 2 * (js * 32 + 8 * q) - 1 + (tap & 3);
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int col = c0 + 2 * jj;
        const bool ok = rok && (unsigned)col < (unsigned)Wl;
        av[m][jj] = ok ? rp[ok ? col : 0] : 0.f;
      }
    }
  };

  f32x4_t acc[CT][NT];
#pragma unroll
  for (int m = 0; m < CT; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[m][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  unsigned s = blockIdx.x * 4 + wave;
  float nxt[CT][8];
  if (s < steps) {
    dma(s, 0);
    load_thin(s, nxt);
  }
  int buf = 0;
  const int r = n >> 2, pc = n & 3;
  for (; s < steps; s += stride) {
    // this wave's DMA of step s (and the thin loads) landed; the transposed reads of the previous step are complete
    // before the DMA below may overwrite their buffer
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    bf16x8_t af[CT];
#pragma unroll
    for (int m = 0; m < CT; ++m) {
      u32x4_t c;
#pragma unroll
      for (int k = 0; k < 4; ++k) c[k] = pk_bf16(nxt[m][2 * k], nxt[m][2 * k + 1]);
      af[m] = as_frag(c);
    }
    if (s + stride < steps) {
      dma(s + stride, buf ^ 1);
      load_thin(s + stride, nxt);
    }
    const char* tb = wbase + buf * BUF;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const char* sub = t < NT0 ? tb : tb + SUB0;
      const int rb = t < NT0 ? RB0 : RB1;
      const int tt = t < NT0 ? t : t - NT0;
      const char* a0 = sub + (8 * q + r) * rb + tt * 32 + pc * 8;
      const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(a0));
      const s16x4_t hi =
          __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(a0 + 4 * rb));
      s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      const bf16x8_t bfr = *reinterpret_cast<bf16x8_t*>(&v);
#pragma unroll
      for (int m = 0; m < CT; ++m) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bfr, acc[m][t], 0, 0, 0);
    }
    buf ^= 1;
  }
  // sum the 4 waves through LDS (the staging buffers are dead: every wave waits for its last DMA above), one slab
  // tile [ROWS][CTOT] per workgroup
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);              // [4][ROWS][CTOT]
#pragma unroll
  for (int m = 0; m < CT; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[(wave * ROWS + 16 * m + 4 * q + e) * CTOT + t * 16 + n] = acc[m][t][e];
  __syncthreads();
  float* out = p.slab + (int64_t)blockIdx.x * ROWS * CTOT;
  for (int e = threadIdx.x; e < ROWS * CTOT; e += 256)
    out[e] = (red[e] + red[ROWS * CTOT + e]) + (red[2 * ROWS * CTOT + e] + red[3 * ROWS * CTOT + e]);
}

// dW[c][row] = sum over the workgroup slabs in a fixed order (bit-reproducible): a workgroup owns 16 consecutive
// outputs, its 16 thread groups each sum every 16th slab, LDS combines the groups.
__global__ __launch_bounds__(256) void thin_wgrad_sum_kernel(const float* slab, int nblk, int rows, int ctot, float* dw) {
  __shared__ float part[16][17];
  const int o = threadIdx.x & 15, sg = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + o;                          // e = row * ctot + c
  const int64_t n = (int64_t)rows * ctot;
  float s0 = 0.f, s1 = 0.f;
  if (e < n) {
    int k = sg;
    for (; k + 16 < nblk; k += 32) {
      s0 += slab[(int64_t)k * n + e];
      s1 += slab[(int64_t)(k + 16) * n + e];
    }
    if (k < nblk) s0 += slab[(int64_t)k * n + e];
  }
  part[sg][o] = s0 + s1;
  __syncthreads();
  if (threadIdx.x < 16 && e < n) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += part[g][o];
    const int row = e / ctot, c = e % ctot;
    dw[c * rows + row] = t;
  }
}

inline int edge_blocks(int64_t units, int cap) {       // 4 waves per workgroup, >= 2 units per wave when there is enough work
  int64_t b = adn_cdiv(units, 8);
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}
constexpr int kPixBlocks = 2048;    // l0_fwd: 8 workgroups per CU
constexpr int kDgradBlocks = 768;   // d0_dgrad: 3 workgroups per CU (143 VGPRs), ~11 pixel groups per wave at B = 32
constexpr int kWgradBlocks = 512;   // thin_wgrad: 2 per CU (64 KiB of LDS each), >= 8 K-steps per wave at B = 32

}  // namespace

extern "C" int adn_l0_forward(const float* x, const float* w, int32_t B, int32_t Hs, int32_t Ws, int32_t cin,
                              int32_t cout, float slope, void* out_leaky, void* out_relu, void* stream) {
  ADN_CHECK_ARG(x && w && (out_leaky || out_relu), "adn_l0_forward: null operand");
  ADN_CHECK_ARG(cin == 2 && cout == 64, "adn_l0_forward: built for 2 -> 64 channels (got %d -> %d)", cin, cout);
  ADN_CHECK_ARG(B > 0 && Hs > 0 && Ws > 0 && Ws % 16 == 0, "adn_l0_forward: bad shape B=%d Hs=%d Ws=%d (Ws %% 16)", B, Hs,
                Ws);
  ADN_CHECK_ARG((int64_t)B * Hs * Ws * 4 * 64 < (1ll << 40) && (int64_t)B * Hs * Ws * 4 < (1ll << 31),
                "adn_l0_forward: tensor too large");
  const int64_t units = (int64_t)B * Hs * (Ws / 16) / ((Ws / 16) % 4 == 0 ? 4 : 1);
  hipLaunchKernelGGL(l0_fwd_kernel, dim3(edge_blocks(units * 2, kPixBlocks)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, w,
                     B, Hs, Ws, slope, reinterpret_cast<uint16_t*>(out_leaky), reinterpret_cast<uint16_t*>(out_relu));
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int64_t adn_d0_dgrad_num_partials(int32_t B, int32_t Hs, int32_t Ws) {
  if (B <= 0 || Hs <= 0 || Ws <= 0 || Ws % 16) return -1;
  return edge_blocks((int64_t)B * Hs * (Ws / 16), kDgradBlocks);
}

extern "C" int adn_d0_dgrad(const float* dz, const float* w, int32_t B, int32_t Hs, int32_t Ws, const AdnEpiSeg* seg0,
                            const AdnEpiSeg* seg1, void* stream) {
  ADN_CHECK_ARG(dz && w && seg0 && seg1, "adn_d0_dgrad: null operand");
  ADN_CHECK_ARG(B > 0 && Hs > 0 && Ws > 0 && Ws % 16 == 0, "adn_d0_dgrad: bad shape B=%d Hs=%d Ws=%d (Ws %% 16)", B, Hs, Ws);
  ADN_CHECK_ARG(seg0->channels == 64 && seg1->channels == 64, "adn_d0_dgrad: built for 64 + 64 input channels (got %d + %d)",
                seg0->channels, seg1->channels);
  ADN_CHECK_ARG((int64_t)B * Hs * Ws * 4 < (1ll << 31), "adn_d0_dgrad: tensor too large");
  D0Params p;
  p.dz = dz;
  p.w = w;
  p.B = B;
  p.Hs = Hs;
  p.Ws = Ws;
  p.seg[0] = *seg0;
  p.seg[1] = *seg1;
  for (int s = 0; s < 2; ++s) {
    ADN_CHECK_ARG(p.seg[s].out0 && p.seg[s].ref, "adn_d0_dgrad: seg %d needs out0 and ref", s);
    ADN_CHECK_ARG(!p.seg[s].partials || (p.seg[s].z && p.seg[s].mean && p.seg[s].istd),
                  "adn_d0_dgrad: seg %d statistics need z / mean / istd", s);
    ADN_CHECK_ARG(!p.seg[s].accumulate, "adn_d0_dgrad: accumulate is not supported");
  }
  ADN_CHECK_ARG(!p.seg[0].partials, "adn_d0_dgrad: statistics are produced for segment 1 only");
  hipLaunchKernelGGL(d0_dgrad_kernel, dim3(edge_blocks((int64_t)B * Hs * (Ws / 16), kDgradBlocks)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), p);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int64_t adn_thin_wgrad_workspace_bytes(int32_t B, int32_t Hs, int32_t Ws, int32_t ct, int32_t c0, int32_t c1) {
  if (B <= 0 || Hs <= 0 || Ws <= 0 || Ws % 32 || (ct != 1 && ct != 2)) return -1;
  return (int64_t)edge_blocks((int64_t)B * Hs * (Ws / 32), kWgradBlocks) * 16 * ct * (c0 + c1) * 4;
}

extern "C" int adn_thin_wgrad(const float* thin, int32_t ct, const void* plain0, int32_t c0, const void* plain1,
                              int32_t c1, int32_t B, int32_t Hs, int32_t Ws, float* dw, void* workspace,
                              int64_t workspace_bytes, void* stream) {
  ADN_CHECK_ARG(thin && plain0 && dw && workspace && (c1 == 0 || plain1), "adn_thin_wgrad: null operand");
  ADN_CHECK_ARG(B > 0 && Hs > 0 && Ws > 0 && Ws % 32 == 0, "adn_thin_wgrad: bad shape B=%d Hs=%d Ws=%d (Ws %% 32)", B, Hs,
                Ws);
  ADN_CHECK_ARG((ct == 1 && c0 == 64 && c1 == 64) || (ct == 2 && c0 == 64 && c1 == 0),
                "adn_thin_wgrad: built for (ct 1, 64 + 64 channels) and (ct 2, 64 channels), got ct %d, %d + %d", ct, c0, c1);
  ADN_CHECK_ARG((int64_t)B * Hs * Ws * 4 < (1ll << 31), "adn_thin_wgrad: tensor too large");
  const int64_t need = adn_thin_wgrad_workspace_bytes(B, Hs, Ws, ct, c0, c1);
  ADN_CHECK_ARG(workspace_bytes >= need, "adn_thin_wgrad: workspace too small (%lld < %lld)", (long long)workspace_bytes,
                (long long)need);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  TWParams p;
  p.thin = thin;
  p.plain0 = reinterpret_cast<const uint16_t*>(plain0);
  p.plain1 = reinterpret_cast<const uint16_t*>(plain1);
  p.B = B;
  p.Hs = Hs;
  p.Ws = Ws;
  p.slab = reinterpret_cast<float*>(workspace);
  const int nblk = edge_blocks((int64_t)B * Hs * (Ws / 32), kWgradBlocks);
  const int rows = 16 * ct, ctot = c0 + c1;
  // LDS: 4 waves x 2 staging buffers x 32 pixels x (c0 + c1) bf16, reused as the [4][rows][ctot] f32 reduction tile
  const int stage = 4 * 2 * 32 * ctot * 2, redb = 4 * rows * ctot * 4;
  const int lds = stage > redb ? stage : redb;
  if (ct == 1) {
    ADN_SET_LDS_ONCE(lds, &thin_wgrad_kernel<1, 4, 4>);
    hipLaunchKernelGGL((thin_wgrad_kernel<1, 4, 4>), dim3(nblk), dim3(256), lds, st, p);
  } else {
    ADN_SET_LDS_ONCE(lds, &thin_wgrad_kernel<2, 4, 0>);
    hipLaunchKernelGGL((thin_wgrad_kernel<2, 4, 0>), dim3(nblk), dim3(256), lds, st, p);
  }
  ADN_CHECK_LAUNCH();
  hipLaunchKernelGGL(thin_wgrad_sum_kernel, dim3((unsigned)adn_cdiv(rows * ctot, 16)), dim3(256), 0, st, p.slab, nblk,
                     rows, ctot, dw);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}
