// MX-fp8 quantisation helpers shared by mx8.hip and the fused BatchNorm epilogues of elementwise.hip (gfx950 only).
#pragma once
#include "adn_common.h"

// ---- MX quantisation of one 32-element block -------------------------------------------------------------------------
// E8M0 byte of a block with absolute maximum amax: the smallest power of two >= amax / 448, so that no element
// saturates (amax = m 2^e: 2^(e-8) when m <= 1.75, else 2^(e-7); the floor rule of the OCP MX v1.0 text, 2^(e-8) always,
// clips the elements with m > 1.75 by up to 12.5 %).
__device__ __forceinline__ int mx_scale_byte(float amax) {
  if (!(amax > 0.f)) return 0;
  const uint32_t b = __float_as_uint(amax);
  int e = (int)((b >> 23) & 0xff) - 8 + ((b & 0x7fffffu) > 0x600000u ? 1 : 0);   // denormal f32 -> 2^-127
  return e < 0 ? 0 : (e > 254 ? 254 : e);
}
// 2^(127 - byte) as a float factor applied in two exact halves (the full factor can exceed the f32 range)
__device__ __forceinline__ float mx_descale(float v, int byte) {
  const int k = 127 - byte;                                   // -127 .. 127
  const int k1 = k / 2, k2 = k - k1;
  return v * __uint_as_float((unsigned)(127 + k1) << 23) * __uint_as_float((unsigned)(127 + k2) << 23);
}
// round-to-nearest-even e4m3fn with saturation at +-448 (the hardware cast; clamped first: |v| < 512 can exceed 448)
__device__ __forceinline__ uint32_t cvt4_e4m3(float a, float b, float c, float d) {
  a = fminf(fmaxf(a, -448.f), 448.f);
  b = fminf(fmaxf(b, -448.f), 448.f);
  c = fminf(fmaxf(c, -448.f), 448.f);
  d = fminf(fmaxf(d, -448.f), 448.f);
  int r = 0;
  r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, r, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
  return (uint32_t)r;
}

// One thread = 8 consecutive channels (values already rounded to what the bf16 tensor holds), 4 consecutive lanes = one
// 32-channel block: returns the 8 e4m3 bytes and the block's E8M0 byte (identical on the 4 lanes).
__device__ __forceinline__ uint2 mx_quant8(float* f, int& byte) {
  float am = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) am = fmaxf(am, fabsf(f[e]));
  am = fmaxf(am, __shfl_xor(am, 1, 64));
  am = fmaxf(am, __shfl_xor(am, 2, 64));
  byte = mx_scale_byte(am);
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = mx_descale(f[e], byte);
  uint2 o;
  o.x = cvt4_e4m3(f[0], f[1], f[2], f[3]);
  o.y = cvt4_e4m3(f[4], f[5], f[6], f[7]);
  return o;
}
