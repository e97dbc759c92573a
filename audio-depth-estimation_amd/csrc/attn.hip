// Binaural cross-attention (BinauralCrossAttention.forward, binaural_attention_model.py:106-153) as a
// flash-style streaming softmax: the N x N score matrix (up to 1 GiB per sample and direction in the reference)
// is never materialised.  Tokens are the H*W pixels of an NHWC tensor, so q/k/v/o rows are channel-contiguous
// token rows: q,k [B2][N][dqk], v,o [B2][N][dv] with arbitrary row strides (they are slices of one fused
// projection buffer).  Entry b attends from queries of batch b to keys/values of batch (b + kv_shift) % B2: with
// [left; right] stacked along the batch this is "left attends right" and "right attends left" in ONE launch.
//
//   S = scale * Q K^T,  P = softmax_keys(S),  O = P V,  lse = logsumexp_keys(S)
//   backward (recompute P from lse):  D = rowsum(dO * O),  dS = P * (dO V^T - D) * scale,
//                                     dQ = dS K,  dK = dS^T Q,  dV = P^T dO
//
// This file holds the generic kernels (any dqk <= 64, dv <= 512, f32 or bf16 storage, f32 arithmetic):
// one wave per query (forward, dQ) or per key (dK/dV), 64 keys / queries per inner step, probabilities
// exchanged through LDS.  The bf16 MFMA kernels for the full-width head dims live in attn_mfma.hip.
#include "adn_common.h"
#include "epilogue.h"

namespace {

constexpr int kMaxDv = 1024;       // value channels (16 per lane)
constexpr int kMaxDqk = 128;       // query / key channels (2 per lane)
constexpr int kWaves = 4;

struct AttnParams {
  const void* q; const void* k; const void* v;
  void* o; float* lse;
  const void* dout; const float* dsum;        // backward: dO and D = rowsum(dO * O)
  void* dq; void* dk; void* dv;
  int B2, N, dqk, dvv, kv_shift;
  int ld_q, ld_k, ld_v, ld_o, ld_do, ld_dq, ld_dk, ld_dv;
  float scale;
};

template <typename T>
__device__ __forceinline__ float ldg(const void* base, int64_t idx) {
  return ElemTraits<T>::load(reinterpret_cast<const T*>(base) + idx);
}

// ---- forward: one wave per query ---------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_generic(AttnParams p) {
  __shared__ float qs[kWaves][kMaxDqk];
  __shared__ float ps[kWaves][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.y;
  const int i = blockIdx.x * kWaves + wave;
  const int kb = (b + p.kv_shift) % p.B2;
  const bool live = i < p.N;
  if (live)
    for (int d = lane; d < p.dqk; d += 64) qs[wave][d] = ldg<T>(p.q, ((int64_t)b * p.N + i) * p.ld_q + d);
  __syncthreads();
  float m = -INFINITY, l = 0.f;
  float o[kMaxDv / 64];
#pragma unroll
  for (int u = 0; u < kMaxDv / 64; ++u) o[u] = 0.f;
  const int nu = (p.dvv + 63) / 64;
  for (int j0 = 0; j0 < p.N; j0 += 64) {
    const int j = j0 + lane;
    float s = -INFINITY;
    if (live && j < p.N) {
      const int64_t kr = ((int64_t)kb * p.N + j) * p.ld_k;
      float a = 0.f;
      for (int d = 0; d < p.dqk; ++d) a += qs[wave][d] * ldg<T>(p.k, kr + d);
      s = a * p.scale;
    }
    const float mn = fmaxf(m, wave_max(s));
    const float alpha = __expf(m - mn);
    const float pj = (live && j < p.N) ? __expf(s - mn) : 0.f;
    l = l * alpha + wave_sum(pj);
    m = mn;
    ps[wave][lane] = pj;
    __syncthreads();
    if (live) {
      const int jn = p.N - j0 < 64 ? p.N - j0 : 64;
      for (int u = 0; u < nu; ++u) {
        const int c = lane + 64 * u;
        if (c < p.dvv) {
          float acc = o[u] * alpha;
          const int64_t vb = ((int64_t)kb * p.N + j0) * p.ld_v + c;
          for (int jj = 0; jj < jn; ++jj) acc += ps[wave][jj] * ldg<T>(p.v, vb + (int64_t)jj * p.ld_v);
          o[u] = acc;
        }
      }
    }
    __syncthreads();
  }
  if (live) {
    const float inv = 1.f / l;
    for (int u = 0; u < nu; ++u) {
      const int c = lane + 64 * u;
      if (c < p.dvv)
        ElemTraits<T>::store(reinterpret_cast<T*>(p.o) + ((int64_t)b * p.N + i) * p.ld_o + c, o[u] * inv);
    }
    if (lane == 0) p.lse[(int64_t)b * p.N + i] = m + __logf(l);
  }
}

// ---- D = rowsum(dO * O) --------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void attn_rowdot(const void* a, int lda, const void* bb, int ldb, int64_t rows, int C,
                                                   float* out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int64_t r = (int64_t)blockIdx.x * kWaves + wave; r < rows; r += (int64_t)gridDim.x * kWaves) {
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += ldg<T>(a, r * lda + c) * ldg<T>(bb, r * ldb + c);
    s = wave_sum(s);
    if (lane == 0) out[r] = s;
  }
}

// ---- backward, dQ: one wave per query --------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dq_generic(AttnParams p) {
  __shared__ float qs[kWaves][kMaxDqk];
  __shared__ float dos[kWaves][kMaxDv];
  __shared__ float dss[kWaves][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.y;
  const int i = blockIdx.x * kWaves + wave;
  const int kb = (b + p.kv_shift) % p.B2;
  const bool live = i < p.N;
  float lse = 0.f, Di = 0.f;
  if (live) {
    const int64_t row = (int64_t)b * p.N + i;
    for (int d = lane; d < p.dqk; d += 64) qs[wave][d] = ldg<T>(p.q, row * p.ld_q + d);
    for (int c = lane; c < p.dvv; c += 64) dos[wave][c] = ldg<T>(p.dout, row * p.ld_do + c);
    lse = p.lse[row];
    Di = p.dsum[row];
  }
  __syncthreads();
  float dq[kMaxDqk / 64] = {0.f, 0.f};
  for (int j0 = 0; j0 < p.N; j0 += 64) {
    const int j = j0 + lane;
    float ds = 0.f;
    if (live && j < p.N) {
      const int64_t kr = ((int64_t)kb * p.N + j) * p.ld_k;
      float a = 0.f;
      for (int d = 0; d < p.dqk; ++d) a += qs[wave][d] * ldg<T>(p.k, kr + d);
      const float pj = __expf(a * p.scale - lse);
      const int64_t vr = ((int64_t)kb * p.N + j) * p.ld_v;
      float dp = 0.f;
      for (int c = 0; c < p.dvv; ++c) dp += dos[wave][c] * ldg<T>(p.v, vr + c);
      ds = pj * (dp - Di) * p.scale;
    }
    dss[wave][lane] = ds;
    __syncthreads();
    if (live) {
      const int jn = p.N - j0 < 64 ? p.N - j0 : 64;
#pragma unroll
      for (int u = 0; u < kMaxDqk / 64; ++u) {
        const int d = lane + 64 * u;
        if (d < p.dqk) {
          const int64_t kbse = ((int64_t)kb * p.N + j0) * p.ld_k + d;
          float acc = dq[u];
          for (int jj = 0; jj < jn; ++jj) acc += dss[wave][jj] * ldg<T>(p.k, kbse + (int64_t)jj * p.ld_k);
          dq[u] = acc;
        }
      }
    }
    __syncthreads();
  }
  if (live)
#pragma unroll
    for (int u = 0; u < kMaxDqk / 64; ++u)
      if (lane + 64 * u < p.dqk)
        ElemTraits<T>::store(reinterpret_cast<T*>(p.dq) + ((int64_t)b * p.N + i) * p.ld_dq + lane + 64 * u, dq[u]);
}

// ---- backward, dK / dV: one wave per key -------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dkv_generic(AttnParams p) {
  __shared__ float ks[kWaves][kMaxDqk];
  __shared__ float vs[kWaves][kMaxDv];
  __shared__ float pss[kWaves][64];
  __shared__ float dss[kWaves][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int kb = blockIdx.y;                                    // batch entry of the keys
  const int qb = (kb - p.kv_shift % p.B2 + p.B2) % p.B2;        // the queries that attend to it
  const int j = blockIdx.x * kWaves + wave;
  const bool live = j < p.N;
  if (live) {
    const int64_t row = (int64_t)kb * p.N + j;
    for (int d = lane; d < p.dqk; d += 64) ks[wave][d] = ldg<T>(p.k, row * p.ld_k + d);
    for (int c = lane; c < p.dvv; c += 64) vs[wave][c] = ldg<T>(p.v, row * p.ld_v + c);
  }
  __syncthreads();
  float dk[kMaxDqk / 64] = {0.f, 0.f};
  float dvacc[kMaxDv / 64];
#pragma unroll
  for (int u = 0; u < kMaxDv / 64; ++u) dvacc[u] = 0.f;
  const int nu = (p.dvv + 63) / 64;
  for (int i0 = 0; i0 < p.N; i0 += 64) {
    const int i = i0 + lane;
    float pj = 0.f, ds = 0.f;
    if (live && i < p.N) {
      const int64_t row = (int64_t)qb * p.N + i;
      float a = 0.f;
      for (int d = 0; d < p.dqk; ++d) a += ks[wave][d] * ldg<T>(p.q, row * p.ld_q + d);
      pj = __expf(a * p.scale - p.lse[row]);
      float dp = 0.f;
      for (int c = 0; c < p.dvv; ++c) dp += vs[wave][c] * ldg<T>(p.dout, row * p.ld_do + c);
      ds = pj * (dp - p.dsum[row]) * p.scale;
    }
    pss[wave][lane] = pj;
    dss[wave][lane] = ds;
    __syncthreads();
    if (live) {
      const int in = p.N - i0 < 64 ? p.N - i0 : 64;
      const int64_t rb = (int64_t)qb * p.N + i0;
      for (int u = 0; u < nu; ++u) {
        const int c = lane + 64 * u;
        if (c < p.dvv) {
          float acc = dvacc[u];
          for (int ii = 0; ii < in; ++ii) acc += pss[wave][ii] * ldg<T>(p.dout, (rb + ii) * p.ld_do + c);
          dvacc[u] = acc;
        }
      }
#pragma unroll
      for (int u = 0; u < kMaxDqk / 64; ++u) {
        const int d = lane + 64 * u;
        if (d < p.dqk) {
          float acc = dk[u];
          for (int ii = 0; ii < in; ++ii) acc += dss[wave][ii] * ldg<T>(p.q, (rb + ii) * p.ld_q + d);
          dk[u] = acc;
        }
      }
    }
    __syncthreads();
  }
  if (live) {
    const int64_t row = (int64_t)kb * p.N + j;
    for (int u = 0; u < nu; ++u) {
      const int c = lane + 64 * u;
      if (c < p.dvv) ElemTraits<T>::store(reinterpret_cast<T*>(p.dv) + row * p.ld_dv + c, dvacc[u]);
    }
#pragma unroll
    for (int u = 0; u < kMaxDqk / 64; ++u)
      if (lane + 64 * u < p.dqk) ElemTraits<T>::store(reinterpret_cast<T*>(p.dk) + row * p.ld_dk + lane + 64 * u, dk[u]);
  }
}

// ---- per-channel sums of a [rows][ld] tensor (bias gradients of the 1x1 projections) -----------------------
// 8 channels (16/32 bytes) per thread, 256 / (C/8) rows per iteration, cross-row-group reduction through LDS.
template <typename T>
__global__ __launch_bounds__(256) void channel_sum_partial(const void* x, int64_t rows, int C, int ld, float* partials) {
  __shared__ float red[256 * 8];
  const int64_t rpb = (rows + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * rpb;
  const int64_t r1 = r0 + rpb < rows ? r0 + rpb : rows;
  float* po = partials + (int64_t)blockIdx.x * C;
  if ((C & 7) == 0 && (ld & 7) == 0) {
    // blockIdx.y selects a 2048-column chunk; inside it 8 columns per thread
    const int c0 = blockIdx.y * 2048;
    const int cw = C - c0 < 2048 ? C - c0 : 2048;
    const int ncg = cw >> 3;
    const int rpi = 256 / ncg;
    const int r = threadIdx.x / ncg, cg = threadIdx.x - r * ncg;
    float s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = 0.f;
    if (r < rpi) {
      for (int64_t row = r0 + r; row < r1; row += rpi) {
        float v[8];
        load8<T>(x, row * ld + c0 + cg * 8, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] += v[k];
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) red[threadIdx.x * 8 + k] = s[k];
    __syncthreads();
    for (int t = threadIdx.x; t < cw; t += 256) {
      const int g = t >> 3, k = t & 7;
      float a = 0.f;
      for (int rr = 0; rr < rpi; ++rr) a += red[(rr * ncg + g) * 8 + k];
      po[c0 + t] = a;
    }
    return;
  }
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c < C) {
    float s = 0.f;
    for (int64_t r = r0; r < r1; ++r) s += ldg<T>(x, r * ld + c);
    po[c] = s;
  }
}

__global__ __launch_bounds__(256) void channel_sum_final(const float* partials, int P, int C, float* out) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (c >= C) return;
  double s = 0.0;
  for (int r = lane; r < P; r += 64) s += (double)partials[(int64_t)r * C + c];
  s = wave_sum_d(s);
  if (lane == 0) out[c] = (float)s;
}

// ---- residual gate x + gamma * proj(att): backward pieces ----------------------------------------------------
// t = dgrad of the out projection for the UNSCALED upstream gradient G.  partial sums of t * att (-> dgamma),
// then t <- gamma * t in place (the gradient that enters the attention backward).
template <typename T>
__global__ __launch_bounds__(256) void gate_bwd_kernel(void* t, const void* att, int64_t n, const float* gamma,
                                                       double* partial) {
  const float g = gamma[0];
  double s = 0.0;
  T* tp = reinterpret_cast<T*>(t);
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const float tv = ElemTraits<T>::load(tp + e);
    s += (double)(tv * ldg<T>(att, e));
    ElemTraits<T>::store(tp + e, tv * g);
  }
  __shared__ double sm[4];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

// dgamma = sum(partials) + sum_c bias[c] * gsum[c];  dbias = gamma * gsum;  dw *= gamma
__global__ __launch_bounds__(256) void gate_bwd_finish_kernel(const double* partial, int P, const float* gsum,
                                                              const float* bias, int C, const float* gamma,
                                                              float* dgamma, float* dbias, float* dw, int64_t nw) {
  const float g = gamma[0];
  if (blockIdx.x == 0) {
    __shared__ double sm[4];
    double s = 0.0;
    for (int r = threadIdx.x; r < P; r += 256) s += partial[r];
    if (bias)
      for (int c = threadIdx.x; c < C; c += 256) s += (double)bias[c] * (double)gsum[c];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) dgamma[0] = (float)(sm[0] + sm[1] + sm[2] + sm[3]);
    if (dbias)
      for (int c = threadIdx.x; c < C; c += 256) dbias[c] = g * gsum[c];
  }
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nw; e += (int64_t)gridDim.x * 256) dw[e] *= g;
}

int avalidate(const AdnAttnDesc* d, bool bwd) {
  ADN_CHECK_ARG(d != nullptr, "adn_attn: null descriptor");
  ADN_CHECK_ARG(d->dtype == ADN_F32 || d->dtype == ADN_BF16, "adn_attn: bad dtype %d", d->dtype);
  ADN_CHECK_ARG(d->B2 > 0 && d->N > 0 && d->dqk > 0 && d->dv > 0, "adn_attn: bad shape");
  ADN_CHECK_ARG(d->dqk <= kMaxDqk && d->dv <= kMaxDv, "adn_attn: head dims %d/%d exceed %d/%d", d->dqk, d->dv, kMaxDqk,
                kMaxDv);
  ADN_CHECK_ARG(d->kv_shift >= 0 && d->kv_shift < d->B2, "adn_attn: bad kv_shift %d", d->kv_shift);
  ADN_CHECK_ARG(d->q && d->k && d->v && d->o && d->lse, "adn_attn: null operand");
  ADN_CHECK_ARG(d->ld_q >= d->dqk && d->ld_k >= d->dqk && d->ld_v >= d->dv && d->ld_o >= d->dv, "adn_attn: bad strides");
  if (bwd) {
    ADN_CHECK_ARG(d->dout && d->dq && d->dk && d->dvp, "adn_attn_bwd: null gradient operand");
    ADN_CHECK_ARG(d->ld_do >= d->dv && d->ld_dq >= d->dqk && d->ld_dk >= d->dqk && d->ld_dv >= d->dv,
                  "adn_attn_bwd: bad gradient strides");
    ADN_CHECK_ARG(d->workspace && d->workspace_bytes >= (int64_t)d->B2 * d->N * 4, "adn_attn_bwd: workspace too small");
  }
  return ADN_OK;
}

AttnParams to_params(const AdnAttnDesc* d) {
  AttnParams p;
  p.q = d->q; p.k = d->k; p.v = d->v; p.o = d->o; p.lse = d->lse;
  p.dout = d->dout; p.dsum = reinterpret_cast<const float*>(d->workspace);
  p.dq = d->dq; p.dk = d->dk; p.dv = d->dvp;
  p.B2 = d->B2; p.N = d->N; p.dqk = d->dqk; p.dvv = d->dv; p.kv_shift = d->kv_shift;
  p.ld_q = d->ld_q; p.ld_k = d->ld_k; p.ld_v = d->ld_v; p.ld_o = d->ld_o;
  p.ld_do = d->ld_do; p.ld_dq = d->ld_dq; p.ld_dk = d->ld_dk; p.ld_dv = d->ld_dv;
  p.scale = d->scale;
  return p;
}

}  // namespace

int adn_attn_mfma_fwd(const AdnAttnDesc* d, hipStream_t st);   // attn_mfma.hip: returns 1 if it took the launch
int adn_attn_mfma_bwd(const AdnAttnDesc* d, hipStream_t st);

extern "C" int adn_attn_fwd(const AdnAttnDesc* d, void* stream) {
  int rc = avalidate(d, false);
  if (rc != ADN_OK) return rc;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (adn_attn_mfma_fwd(d, st) == 1) {
    ADN_CHECK_LAUNCH();
    return ADN_OK;
  }
  const AttnParams p = to_params(d);
  const dim3 grid((unsigned)adn_cdiv(d->N, kWaves), d->B2);
  if (d->dtype == ADN_BF16) hipLaunchKernelGGL((attn_fwd_generic<uint16_t>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((attn_fwd_generic<float>), grid, dim3(256), 0, st, p);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int64_t adn_attn_bwd_workspace_bytes(const AdnAttnDesc* d) {
  if (!d || d->B2 <= 0 || d->N <= 0) return -1;
  return (int64_t)d->B2 * d->N * 4;
}

extern "C" int adn_attn_bwd(const AdnAttnDesc* d, void* stream) {
  int rc = avalidate(d, true);
  if (rc != ADN_OK) return rc;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t rows = (int64_t)d->B2 * d->N;
  unsigned rb = (unsigned)adn_cdiv(rows, kWaves);
  if (rb > 4096) rb = 4096;
  float* dsum = reinterpret_cast<float*>(d->workspace);
  if (d->dtype == ADN_BF16)
    hipLaunchKernelGGL((attn_rowdot<uint16_t>), dim3(rb), dim3(256), 0, st, d->dout, d->ld_do, d->o, d->ld_o, rows, d->dv, dsum);
  else
    hipLaunchKernelGGL((attn_rowdot<float>), dim3(rb), dim3(256), 0, st, d->dout, d->ld_do, d->o, d->ld_o, rows, d->dv, dsum);
  ADN_CHECK_LAUNCH();
  if (adn_attn_mfma_bwd(d, st) == 1) {
    ADN_CHECK_LAUNCH();
    return ADN_OK;
  }
  const AttnParams p = to_params(d);
  const dim3 grid((unsigned)adn_cdiv(d->N, kWaves), d->B2);
  if (d->dtype == ADN_BF16) {
    hipLaunchKernelGGL((attn_bwd_dq_generic<uint16_t>), grid, dim3(256), 0, st, p);
    hipLaunchKernelGGL((attn_bwd_dkv_generic<uint16_t>), grid, dim3(256), 0, st, p);
  } else {
    hipLaunchKernelGGL((attn_bwd_dq_generic<float>), grid, dim3(256), 0, st, p);
    hipLaunchKernelGGL((attn_bwd_dkv_generic<float>), grid, dim3(256), 0, st, p);
  }
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int64_t adn_channel_sum_workspace_bytes(int64_t rows, int32_t C) {
  if (rows <= 0 || C <= 0) return -1;
  int64_t P = adn_cdiv(rows, 256);
  if (P > 1024) P = 1024;
  return P * C * 4;
}

extern "C" int adn_channel_sum(const void* x, int64_t rows, int32_t C, int32_t ld, int32_t dtype, float* out,
                               void* workspace, int64_t workspace_bytes, void* stream) {
  ADN_CHECK_ARG(x && out && rows > 0 && C > 0 && ld >= C, "adn_channel_sum: bad arguments");
  ADN_CHECK_ARG(dtype == ADN_F32 || dtype == ADN_BF16, "adn_channel_sum: bad dtype %d", dtype);
  const int64_t need = adn_channel_sum_workspace_bytes(rows, C);
  ADN_CHECK_ARG(workspace && workspace_bytes >= need, "adn_channel_sum: workspace too small");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int P = (int)(need / (4 * (int64_t)C));
  float* part = reinterpret_cast<float*>(workspace);
  const bool vec = (C & 7) == 0 && (ld & 7) == 0;
  const dim3 grid(P, (unsigned)adn_cdiv(C, vec ? 2048 : 256));
  if (dtype == ADN_BF16) hipLaunchKernelGGL((channel_sum_partial<uint16_t>), grid, dim3(256), 0, st, x, rows, C, ld, part);
  else hipLaunchKernelGGL((channel_sum_partial<float>), grid, dim3(256), 0, st, x, rows, C, ld, part);
  ADN_CHECK_LAUNCH();
  hipLaunchKernelGGL(channel_sum_final, dim3((unsigned)adn_cdiv(C, 4)), dim3(256), 0, st, part, P, C, out);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_gate_bwd(void* t, const void* att, int64_t n, int32_t dtype, const float* gamma, const float* gsum,
                            const float* bias, int32_t C, float* dgamma, float* dbias, float* dw, int64_t nw,
                            void* workspace, int64_t workspace_bytes, void* stream) {
  ADN_CHECK_ARG(t && att && gamma && gsum && dgamma && n > 0 && C > 0 && nw >= 0, "adn_gate_bwd: bad arguments");
  ADN_CHECK_ARG(dtype == ADN_F32 || dtype == ADN_BF16, "adn_gate_bwd: bad dtype %d", dtype);
  ADN_CHECK_ARG(workspace && workspace_bytes >= 1024 * 8, "adn_gate_bwd: workspace too small (needs 8 KiB)");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int64_t nb = adn_cdiv(n, 256 * 8);
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  double* part = reinterpret_cast<double*>(workspace);
  if (dtype == ADN_BF16) hipLaunchKernelGGL((gate_bwd_kernel<uint16_t>), dim3((unsigned)nb), dim3(256), 0, st, t, att, n, gamma, part);
  else hipLaunchKernelGGL((gate_bwd_kernel<float>), dim3((unsigned)nb), dim3(256), 0, st, t, att, n, gamma, part);
  ADN_CHECK_LAUNCH();
  int64_t fb = adn_cdiv(nw, 256);
  if (fb > 1024) fb = 1024;
  if (fb < 1) fb = 1;
  hipLaunchKernelGGL(gate_bwd_finish_kernel, dim3((unsigned)fb), dim3(256), 0, st, part, (int)nb, gsum, bias, C, gamma,
                     dgamma, dbias, dw, nw);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}
