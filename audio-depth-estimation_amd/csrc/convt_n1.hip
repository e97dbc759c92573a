// Outermost ConvTranspose2d(k4,s2,p1) with a single output channel (the depth map), forward.
//
// With Cout = 1 the 4-phase implicit GEMM degenerates to N = 1.  Instead: every input pixel m first
// produces its 16 kernel-tap products P[m][kh*4+kw] = sum_c in[m][c] * W[c][kh][kw] -- a pointwise GEMM
// [M x Cin] x [Cin x 16] that fits one MFMA N-tile with the weights resident in LDS and the activations
// streamed straight from HBM into the A fragments (no LDS staging: each element is used once) -- and
// a second pass adds, per output pixel, the four (input pixel, tap) products that land on it (col2im),
// plus bias and the final ReLU / Sigmoid.  Both passes are HBM-bound: 2*Cin bytes per input pixel in,
// 64 + 64 bytes of P out/in, 16 bytes of output.
#include "epilogue.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void convt_n1_partial_kernel(const T* in0, int C0, const T* in1, int C1,
                                                               const float* w, int64_t M, float* P) {
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int KS = 4 * EPC;                       // channels per MFMA step group (32 bf16 / 16 f32)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32x4_t* wl = reinterpret_cast<u32x4_t*>(smem);   // [nk][64 lanes]
  const int Cin = C0 + C1;
  const int nk = Cin / KS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  // B operand: lane (tap = fr, k-group fq) of step s holds W[c = s*KS + fq*EPC + j][tap]
  for (int e = tid; e < nk * 64; e += 256) {
    const int s = e >> 6, l = e & 63;
    const int tap = l & 15, q = l >> 4;
    float f[EPC];
#pragma unroll
    for (int j = 0; j < EPC; ++j) f[j] = w[(int64_t)(s * KS + q * EPC + j) * 16 + tap];
    wl[e] = Chunk<T>::pack(f);
  }
  __syncthreads();
  const int64_t groups = (M + 15) >> 4;
  const int nk0 = C0 / KS;
  for (int64_t g = (int64_t)blockIdx.x * 4 + wave; g < groups; g += (int64_t)gridDim.x * 4) {
    const int64_t m = g * 16 + fr;
    const bool ok = m < M;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < nk; ++s) {
      u32x4_t a = {0u, 0u, 0u, 0u};
      if (ok) {
        if (s < nk0) a = *reinterpret_cast<const u32x4_t*>(in0 + m * C0 + s * KS + fq * EPC);
        else a = *reinterpret_cast<const u32x4_t*>(in1 + m * C1 + (s - nk0) * KS + fq * EPC);
      }
      const u32x4_t b = wl[s * 64 + lane];
      mma_tile<T>(a, b, acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t mm = g * 16 + 4 * fq + r;
      if (mm < M) P[mm * 16 + fr] = acc[r];
    }
  }
}

__global__ __launch_bounds__(256) void convt_n1_gather_kernel(const float* P, int B, int Hs, int Ws,
                                                              const float* bias, int final_act, float* out) {
  const int Hl = 2 * Hs, Wl = 2 * Ws;
  const int64_t n = (int64_t)B * Hl * Wl;
  const float bv = bias ? bias[0] : 0.f;
  // (32-bit index arithmetic: the host checks n < 2^31; a 64-bit division is a several-hundred-instruction routine)
  for (unsigned e = blockIdx.x * 256 + threadIdx.x; e < (unsigned)n; e += gridDim.x * 256) {
    const unsigned rowi = e / (unsigned)Wl;
    const int ox = (int)(e - rowi * (unsigned)Wl);
    const int b = (int)(rowi / (unsigned)Hl);
    const int oy = (int)(rowi - (unsigned)b * Hl);
    const int ph = oy & 1, pw = ox & 1, i = oy >> 1, j = ox >> 1;
    float v = bv;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int iy = i + adn_t2_dy(ph, t >> 1), ix = j + adn_t2_dy(pw, t & 1);
      if ((unsigned)iy < (unsigned)Hs && (unsigned)ix < (unsigned)Ws)
        v += P[(((int64_t)b * Hs + iy) * Ws + ix) * 16 + adn_t2_kh(ph, t >> 1) * 4 + adn_t2_kh(pw, t & 1)];
    }
    out[e] = adn_final_act(v, final_act);
  }
}

}  // namespace

extern "C" int64_t adn_convt_n1_workspace_bytes(int32_t B, int32_t Hs, int32_t Ws) {
  if (B <= 0 || Hs <= 0 || Ws <= 0) return -1;
  return (int64_t)B * Hs * Ws * 16 * 4;
}

extern "C" int adn_convt_n1_forward(int32_t dtype, int32_t B, int32_t Hs, int32_t Ws, const void* in0, int32_t C0,
                                    const void* in1, int32_t C1, const float* w, const float* bias,
                                    int32_t final_act, float* out, void* workspace, int64_t workspace_bytes,
                                    void* stream) {
  ADN_CHECK_ARG(dtype == ADN_F32 || dtype == ADN_BF16, "adn_convt_n1_forward: bad dtype %d", dtype);
  ADN_CHECK_ARG(B > 0 && Hs > 0 && Ws > 0 && C0 > 0 && C1 >= 0, "adn_convt_n1_forward: bad shape");
  ADN_CHECK_ARG(in0 && (C1 == 0 || in1) && w && out && workspace, "adn_convt_n1_forward: null operand");
  const int ks = dtype == ADN_BF16 ? 32 : 16;
  ADN_CHECK_ARG(C0 % ks == 0 && C1 % ks == 0, "adn_convt_n1_forward: channels must be multiples of %d (got %d+%d)",
                ks, C0, C1);
  const int64_t M = (int64_t)B * Hs * Ws;
  ADN_CHECK_ARG(workspace_bytes >= M * 64, "adn_convt_n1_forward: workspace too small");
  ADN_CHECK_ARG(M * 4 < (1ll << 31), "adn_convt_n1_forward: tensor too large");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  float* P = reinterpret_cast<float*>(workspace);
  const int nk = (C0 + C1) / ks;
  const int lds = nk * 64 * 16;
  int64_t blocks = adn_cdiv(adn_cdiv(M, 16), 4);
  if (blocks > 2048) blocks = 2048;
  if (dtype == ADN_BF16)
    hipLaunchKernelGGL((convt_n1_partial_kernel<uint16_t>), dim3((unsigned)blocks), dim3(256), lds, st,
                       reinterpret_cast<const uint16_t*>(in0), C0, reinterpret_cast<const uint16_t*>(in1), C1, w, M,
                       P);
  else
    hipLaunchKernelGGL((convt_n1_partial_kernel<float>), dim3((unsigned)blocks), dim3(256), lds, st,
                       reinterpret_cast<const float*>(in0), C0, reinterpret_cast<const float*>(in1), C1, w, M, P);
  ADN_CHECK_LAUNCH();
  int64_t gb = adn_cdiv(M * 4, 256);
  if (gb > 4096) gb = 4096;
  hipLaunchKernelGGL(convt_n1_gather_kernel, dim3((unsigned)gb), dim3(256), 0, st, P, B, Hs, Ws, bias, final_act, out);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}
