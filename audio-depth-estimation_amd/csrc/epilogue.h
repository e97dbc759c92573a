// Epilogue of the implicit-GEMM family, shared by the fused MFMA path (8 channels per thread,
// 16-byte accesses) and the split-K / generic reduce path (1 element per call).
#pragma once
#include "adn_common.h"

__device__ __forceinline__ float adn_final_act(float y, int kind) {
  return kind == 1 ? 1.0f / (1.0f + __expf(-y)) : fmaxf(y, 0.0f);
}

// ---- scalar form ---------------------------------------------------------------------------------
// v: GEMM result for output pixel `op`, channel `nl` of segment `sg`.  s1/s2 accumulate stats.
template <typename T>
__device__ __forceinline__ void epi_scalar(int epi, const AdnEpiSeg& sg, int64_t op, int nl, float v,
                                           float& s1, float& s2) {
  const int64_t idx = op * sg.channels + nl;
  if (epi == ADN_EPI_RAW) {
    reinterpret_cast<float*>(sg.out0)[idx] = v;
  } else if (epi == ADN_EPI_Z_STATS) {
    if (sg.bias) v += sg.bias[nl];
    ElemTraits<T>::store(reinterpret_cast<T*>(sg.out0) + idx, v);
    s1 += v;
    s2 += v * v;
  } else if (epi == ADN_EPI_ACT) {
    float y = v;
    if (sg.scale) y = y * sg.scale[nl];
    if (sg.shift) y += sg.shift[nl];
    if (sg.bias) y += sg.bias[nl];
    if (sg.out0) ElemTraits<T>::store(reinterpret_cast<T*>(sg.out0) + idx, y > 0.f ? y : y * sg.slope);
    if (sg.out1) ElemTraits<T>::store(reinterpret_cast<T*>(sg.out1) + idx, fmaxf(y, 0.f));
  } else if (epi == ADN_EPI_BWD) {
    const float r = ElemTraits<T>::load(reinterpret_cast<const T*>(sg.ref) + idx);
    float g = v * (r > 0.f ? 1.0f : sg.slope);
    if (sg.accumulate) g += ElemTraits<T>::load(reinterpret_cast<const T*>(sg.out0) + idx);
    ElemTraits<T>::store(reinterpret_cast<T*>(sg.out0) + idx, g);
    if (sg.partials) {
      const float z = ElemTraits<T>::load(reinterpret_cast<const T*>(sg.z) + idx);
      s1 += g;
      s2 += g * ((z - sg.mean[nl]) * sg.istd[nl]);
    }
  } else if (epi == ADN_EPI_ADD) {
    float g = v;
    if (sg.bias) g += sg.bias[nl];
    if (sg.scale) g *= sg.final_act ? sg.scale[0] : sg.scale[nl];
    if (sg.ref) g += ElemTraits<T>::load(reinterpret_cast<const T*>(sg.ref) + idx);
    if (sg.accumulate) g += ElemTraits<T>::load(reinterpret_cast<const T*>(sg.out0) + idx);
    ElemTraits<T>::store(reinterpret_cast<T*>(sg.out0) + idx, g);
  } else {  // ADN_EPI_FINAL
    float y = v;
    if (sg.bias) y += sg.bias[nl];
    reinterpret_cast<float*>(sg.out0)[idx] = adn_final_act(y, sg.final_act);
  }
}

// ---- 8-channel vector form -------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void load8(const void* base, int64_t idx, float* f) {
  if constexpr (sizeof(T) == 2) {
    u32x4_t c = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const uint16_t*>(base) + idx);
    Chunk<uint16_t>::unpack(c, f);
  } else {
    const u32x4_t* p = reinterpret_cast<const u32x4_t*>(reinterpret_cast<const float*>(base) + idx);
    u32x4_t c0 = p[0], c1 = p[1];
    Chunk<float>::unpack(c0, f);
    Chunk<float>::unpack(c1, f + 4);
  }
}
template <typename T>
__device__ __forceinline__ void store8(void* base, int64_t idx, const float* f) {
  if constexpr (sizeof(T) == 2) {
    *reinterpret_cast<u32x4_t*>(reinterpret_cast<uint16_t*>(base) + idx) = Chunk<uint16_t>::pack(f);
  } else {
    u32x4_t* p = reinterpret_cast<u32x4_t*>(reinterpret_cast<float*>(base) + idx);
    p[0] = Chunk<float>::pack(f);
    p[1] = Chunk<float>::pack(f + 4);
  }
}

// Per-thread constants of an 8-channel column group (hoisted out of the row loop).
struct EpiCols {
  float a[8], b[8];  // ACT: scale, shift(+bias); BWD: mean, istd; FINAL: bias in b
};

template <typename T>
__device__ __forceinline__ void epi_cols_init(int epi, const AdnEpiSeg& sg, int nl, EpiCols& c) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    c.a[e] = 1.0f;
    c.b[e] = 0.0f;
  }
  if (epi == ADN_EPI_ACT) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (sg.scale) c.a[e] = sg.scale[nl + e];
      if (sg.shift) c.b[e] = sg.shift[nl + e];
      if (sg.bias) c.b[e] += sg.bias[nl + e];
    }
  } else if (epi == ADN_EPI_BWD && sg.partials) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      c.a[e] = sg.mean[nl + e];
      c.b[e] = sg.istd[nl + e];
    }
  } else if ((epi == ADN_EPI_FINAL || epi == ADN_EPI_Z_STATS) && sg.bias) {
#pragma unroll
    for (int e = 0; e < 8; ++e) c.b[e] = sg.bias[nl + e];
  } else if (epi == ADN_EPI_ADD) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (sg.scale) c.a[e] = sg.final_act ? sg.scale[0] : sg.scale[nl + e];
      if (sg.bias) c.b[e] = sg.bias[nl + e];
    }
  }
}

template <typename T>
__device__ __forceinline__ void epi_vec8(int epi, const AdnEpiSeg& sg, const EpiCols& c, int64_t op, int nl,
                                         const float* v, float* s1, float* s2) {
  const int64_t idx = op * sg.channels + nl;
  if (epi == ADN_EPI_RAW) {
    store8<float>(sg.out0, idx, v);
  } else if (epi == ADN_EPI_Z_STATS) {
    float zb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) zb[e] = v[e] + c.b[e];       // c.b = bias (0 when the conv has none)
    store8<T>(sg.out0, idx, zb);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s1[e] += zb[e];
      s2[e] += zb[e] * zb[e];
    }
  } else if (epi == ADN_EPI_ACT) {
    float y[8], o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) y[e] = v[e] * c.a[e] + c.b[e];
    if (sg.out0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = y[e] > 0.f ? y[e] : y[e] * sg.slope;
      store8<T>(sg.out0, idx, o);
    }
    if (sg.out1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = fmaxf(y[e], 0.f);
      store8<T>(sg.out1, idx, o);
    }
  } else if (epi == ADN_EPI_BWD) {
    float r[8], g[8];
    load8<T>(sg.ref, idx, r);
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = v[e] * (r[e] > 0.f ? 1.0f : sg.slope);
    if (sg.accumulate) {
      float old[8];
      load8<T>(sg.out0, idx, old);
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] += old[e];
    }
    store8<T>(sg.out0, idx, g);
    if (sg.partials) {
      float z[8];
      load8<T>(sg.z, idx, z);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s1[e] += g[e];
        s2[e] += g[e] * ((z[e] - c.a[e]) * c.b[e]);
      }
    }
  } else if (epi == ADN_EPI_ADD) {
    float g[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = (v[e] + c.b[e]) * c.a[e];       // a = 1, b = 0 when not given
    if (sg.ref) {
      float r[8];
      load8<T>(sg.ref, idx, r);
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] += r[e];
    }
    if (sg.accumulate) {
      float old[8];
      load8<T>(sg.out0, idx, old);
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] += old[e];
    }
    store8<T>(sg.out0, idx, g);
  } else {  // FINAL
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = adn_final_act(v[e] + c.b[e], sg.final_act);
    store8<float>(sg.out0, idx, o);
  }
}

// BWD epilogue (bf16) on operand chunks that were loaded ahead of the K loop: same arithmetic as the ADN_EPI_BWD branch
// of epi_vec8.
__device__ __forceinline__ void epi_bwd_pre8(const AdnEpiSeg& sg, const EpiCols& c, int64_t op, int nl, const float* v,
                                             const u32x4_t& ref_raw, const u32x4_t& old_raw, const u32x4_t& z_raw,
                                             float* s1, float* s2) {
  const int64_t idx = op * sg.channels + nl;
  float r[8], g[8];
  Chunk<uint16_t>::unpack(ref_raw, r);
#pragma unroll
  for (int e = 0; e < 8; ++e) g[e] = v[e] * (r[e] > 0.f ? 1.0f : sg.slope);
  if (sg.accumulate) {
    float old[8];
    Chunk<uint16_t>::unpack(old_raw, old);
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] += old[e];
  }
  store8<uint16_t>(sg.out0, idx, g);
  if (sg.partials) {
    float z[8];
    Chunk<uint16_t>::unpack(z_raw, z);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s1[e] += g[e];
      s2[e] += g[e] * ((z[e] - c.a[e]) * c.b[e]);
    }
  }
}
