// Device-side depth evaluation metrics: one 1024-thread workgroup per sample walks the sample's
// pixels four times (mask/max, candidate counts, kept max, error sums), reproducing every branch of
// compute_errors (/root/reference/utils_criterion.py:6-90) without copying the maps to the host.
#include "adn_common.h"

namespace {

constexpr int MT = 1024;

__device__ __forceinline__ double blk_sum(double v, double* sh) {
  v = wave_sum_d(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < MT / 64; ++w) t += sh[w];
  return t;
}
__device__ __forceinline__ float blk_max(float v, double* sh) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = (double)v;
  __syncthreads();
  float t = -INFINITY;
  for (int w = 0; w < MT / 64; ++w) t = fmaxf(t, (float)sh[w]);
  return t;
}

__global__ __launch_bounds__(MT) void compute_errors_kernel(const float* gt_all, const float* pred_all, int64_t pixels,
                                                            float* out7) {
  __shared__ double sh[MT / 64];
  const float* gt = gt_all + (int64_t)blockIdx.x * pixels;
  const float* pr = pred_all + (int64_t)blockIdx.x * pixels;
  float* out = out7 + (int64_t)blockIdx.x * 7;
  const int tid = threadIdx.x;

  // pass A: mask = gt != 0 (:22); count and max of masked gt
  double n0 = 0.0;
  float gmax = -INFINITY;
  for (int64_t i = tid; i < pixels; i += MT) {
    const float g = gt[i];
    if (g != 0.0f) {
      n0 += 1.0;
      gmax = fmaxf(gmax, g);
    }
  }
  n0 = blk_sum(n0, sh);
  gmax = blk_max(gmax, sh);
  if (n0 == 0.0) {
    if (tid < 7) out[tid] = 0.0f;
    return;
  }
  const float eps = gmax > 1.0f ? 1e-3f : 1e-6f;   // :33

  // pass B: candidate set sizes (:34-54)
  double k1 = 0.0, k2 = 0.0, k3 = 0.0;
  for (int64_t i = tid; i < pixels; i += MT) {
    const float g = gt[i], p = pr[i];
    if (g != 0.0f) {
      const bool ge = g > eps;
      k1 += (ge && p > eps) ? 1.0 : 0.0;
      k2 += ge ? 1.0 : 0.0;
      k3 += (ge && p > 0.0f) ? 1.0 : 0.0;
    }
  }
  k1 = blk_sum(k1, sh);
  k2 = blk_sum(k2, sh);
  k3 = blk_sum(k3, sh);
  int mode;  // 0: (p>eps)&(g>eps)   1: (g>eps)&(p>0)
  if (k1 > 0.0) {
    mode = 0;
  } else if (k2 == 0.0) {
    if (tid < 7) out[tid] = 0.0f;
    return;
  } else if (k3 > 0.0) {
    mode = 1;
  } else {
    if (tid == 0) {   // all predictions negative or zero (:47-54)
      out[0] = 1.0f; out[1] = gmax; out[2] = 0.f; out[3] = 0.f; out[4] = 0.f; out[5] = 1.0f; out[6] = gmax;
    }
    return;
  }

  // pass C: eps re-evaluated on the kept gt (:60)
  float gmax2 = -INFINITY;
  for (int64_t i = tid; i < pixels; i += MT) {
    const float g = gt[i], p = pr[i];
    const bool keep = g != 0.0f && g > eps && (mode == 0 ? p > eps : p > 0.0f);
    if (keep) gmax2 = fmaxf(gmax2, g);
  }
  gmax2 = blk_max(gmax2, sh);
  const float eps2 = gmax2 > 1.0f ? 1e-3f : 1e-6f;

  // pass D: sums (:61-83)
  double cnt = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0, sq = 0.0, rel = 0.0, lg = 0.0, ab = 0.0;
  for (int64_t i = tid; i < pixels; i += MT) {
    const float g = gt[i], p = pr[i];
    const bool keep = g != 0.0f && g > eps && (mode == 0 ? p > eps : p > 0.0f);
    if (keep) {
      const float pc = fmaxf(p, eps2);
      const float th = fmaxf(g / pc, pc / g);
      cnt += 1.0;
      c1 += th < 1.25f ? 1.0 : 0.0;
      c2 += th < 1.5625f ? 1.0 : 0.0;        // 1.25**2
      c3 += th < 1.953125f ? 1.0 : 0.0;      // 1.25**3
      const float d = g - p;
      sq += (double)d * (double)d;
      rel += (double)(fabsf(d) / g);
      lg += (double)fabsf(log10f(fmaxf(g, eps2)) - log10f(pc));
      ab += (double)fabsf(d);
    }
  }
  cnt = blk_sum(cnt, sh);
  c1 = blk_sum(c1, sh);
  c2 = blk_sum(c2, sh);
  c3 = blk_sum(c3, sh);
  sq = blk_sum(sq, sh);
  rel = blk_sum(rel, sh);
  lg = blk_sum(lg, sh);
  ab = blk_sum(ab, sh);
  if (tid == 0) {
    auto clean = [](double x) { return (x != x || x == INFINITY) ? 0.0f : (float)x; };
    out[0] = clean(rel / cnt);
    out[1] = clean(sqrt(sq / cnt));
    out[2] = clean(c1 / cnt);
    out[3] = clean(c2 / cnt);
    out[4] = clean(c3 / cnt);
    out[5] = clean(lg / cnt);
    out[6] = clean(ab / cnt);
  }
}

}  // namespace

extern "C" int64_t adn_compute_errors_workspace_bytes(int32_t, int64_t) { return 0; }

extern "C" int adn_compute_errors(const float* gt, const float* pred, int32_t samples, int64_t pixels, float* out7,
                                  void* workspace, int64_t workspace_bytes, void* stream) {
  (void)workspace;
  (void)workspace_bytes;
  ADN_CHECK_ARG(gt && pred && out7 && samples > 0 && pixels > 0, "adn_compute_errors: bad arguments");
  hipLaunchKernelGGL(compute_errors_kernel, dim3(samples), dim3(MT), 0, reinterpret_cast<hipStream_t>(stream), gt,
                     pred, pixels, out7);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}
