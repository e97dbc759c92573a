// Shared device/host helpers for libadn (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/adn.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

void adn_set_error(const char* fmt, ...);

#define ADN_CHECK_ARG(cond, ...)          \
  do {                                    \
    if (!(cond)) {                        \
      adn_set_error(__VA_ARGS__);         \
      return ADN_ERR_ARG;                 \
    }                                     \
  } while (0)

#define ADN_CHECK_LAUNCH()                                                   \
  do {                                                                       \
    hipError_t e_ = hipGetLastError();                                       \
    if (e_ != hipSuccess) {                                                  \
      adn_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,           \
                    hipGetErrorString(e_));                                  \
      return ADN_ERR_LAUNCH;                                                 \
    }                                                                        \
  } while (0)

// ---- element conversion ------------------------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t h) {
  return __uint_as_float(((uint32_t)h) << 16);
}
// round-to-nearest-even via the hardware cast (keeps NaN a NaN, MI355X_MICROARCH.md).
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
  __bf16 b = (__bf16)f;
  return *reinterpret_cast<uint16_t*>(&b);
}

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
  static constexpr int kDtype = ADN_F32;
  static constexpr int kPerChunk = 4;  // elements per 16-byte chunk
  __device__ static __forceinline__ float load(const float* p) { return *p; }
  __device__ static __forceinline__ void store(float* p, float v) { *p = v; }
};
template <> struct ElemTraits<uint16_t> {  // bf16 stored as raw bits
  static constexpr int kDtype = ADN_BF16;
  static constexpr int kPerChunk = 8;
  __device__ static __forceinline__ float load(const uint16_t* p) { return bf16_bits_to_f32(*p); }
  __device__ static __forceinline__ void store(uint16_t* p, float v) { *p = f32_to_bf16_bits(v); }
};

// 16-byte chunk <-> floats
template <typename T> struct Chunk;
template <> struct Chunk<float> {
  static constexpr int N = 4;
  __device__ static __forceinline__ void unpack(const u32x4_t& c, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(c[i]);
  }
  __device__ static __forceinline__ u32x4_t pack(const float* f) {
    u32x4_t c;
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = __float_as_uint(f[i]);
    return c;
  }
};
template <> struct Chunk<uint16_t> {
  static constexpr int N = 8;
  __device__ static __forceinline__ void unpack(const u32x4_t& c, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = __uint_as_float(c[i] << 16);
      f[2 * i + 1] = __uint_as_float(c[i] & 0xffff0000u);
    }
  }
  __device__ static __forceinline__ u32x4_t pack(const float* f) {
    u32x4_t c;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      c[i] = (uint32_t)f32_to_bf16_bits(f[2 * i]) | ((uint32_t)f32_to_bf16_bits(f[2 * i + 1]) << 16);
    return c;
  }
};

// ---- transposed-conv / conv-dgrad phase tables (k4 s2 p1) ------------------------------------
// Output row oy = 2*i + ph gets input rows i + DY(ph,t) through kernel row KH(ph,t), t in {0,1}:
//   oy = 2*iy - 1 + kh  =>  ph=0: (iy=i, kh=1), (iy=i-1, kh=3);  ph=1: (iy=i+1, kh=0), (iy=i, kh=2)
__host__ __device__ __forceinline__ int adn_t2_kh(int ph, int t) { return ph == 0 ? (t == 0 ? 1 : 3) : (t == 0 ? 0 : 2); }
__host__ __device__ __forceinline__ int adn_t2_dy(int ph, int t) { return ph == 0 ? (t == 0 ? 0 : -1) : (t == 0 ? 1 : 0); }

// ---- wave reductions ---------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- one 16x16 MFMA tile step on 16-byte operand chunks (bf16: K=32 in one op; f32: 4 exact K=4 ops) ----
template <typename T>
__device__ __forceinline__ void mma_tile(const u32x4_t& a, const u32x4_t& b, f32x4_t& acc) {
  if constexpr (sizeof(T) == 2) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(&a),
                                                  *reinterpret_cast<const bf16x8_t*>(&b), acc, 0, 0, 0);
  } else {
    // exact f32: lane group q=(lane>>4) holds k = 4q..4q+3 of this 16-wide slice for A and B alike;
    // MFMA step s contracts element s of every lane group (a permutation of k, same on both sides).
#pragma unroll
    for (int s = 0; s < 4; ++s)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[s]), __uint_as_float(b[s]), acc, 0, 0, 0);
  }
}

static inline int64_t adn_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
// Raise a kernel's dynamic-LDS limit exactly once per instantiation.  Launch functions run concurrently on the
// caller's threads (backward is driven from autograd threads): a plain static flag would be a data race.
#include <mutex>
#define ADN_SET_LDS_ONCE(bytes, ...)                                                                       \
  do {                                                                                                     \
    static std::once_flag adn_once_;                                                                       \
    std::call_once(adn_once_, [&] {                                                                        \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(__VA_ARGS__),                                \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (bytes));                      \
    });                                                                                                    \
  } while (0)


