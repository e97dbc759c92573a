// Masked depth loss, final-activation backward, gradient-norm clip and fused optimizer step.
// All streaming / reduction kernels (HBM-bound); reductions are two-stage and deterministic:
// per-block partials in f64, then one block sums them in a fixed order.  No host round trips: the
// loss statistics, clip coefficient and Adam bias corrections live in device memory so a whole
// training step can be captured into one hipGraph.
#include "epilogue.h"

namespace {

constexpr int kRedBlocks = 1024;

__device__ __forceinline__ double block_sum_d(double v, double* sh) {
  v = wave_sum_d(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
  return t;
}

// mask_mode: low two bits = validity rule (0: gt != 0, 1: gt > 0, 2: every pixel); bit 2 (ADN_MASK_SQUARED = 4):
// the second statistic is the sum of SQUARED errors instead of absolute errors (criterion 4 = masked MSE)
__device__ __forceinline__ bool valid_px(float g, int mask_mode) {
  const int m = mask_mode & 3;
  return m == 0 ? (g != 0.0f) : (m == 1 ? (g > 0.0f) : true);
}

__global__ __launch_bounds__(256) void loss_stats_partial_kernel(const float* pred, const float* gt, int64_t n,
                                                                 float scale, int mask_mode, float eps,
                                                                 double* partials) {
  __shared__ double sh[4];
  double cN = 0.0, sA = 0.0, sD = 0.0, sD2 = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float g0 = gt[i];
    if (valid_px(g0, mask_mode)) {
      const float p = pred[i] * scale, g = g0 * scale;
      const float d = logf(fmaxf(p, eps)) - logf(fmaxf(g, eps));
      cN += 1.0;
      sA += (mask_mode & 4) ? (double)(p - g) * (double)(p - g) : (double)fabsf(p - g);
      sD += (double)d;
      sD2 += (double)d * (double)d;
    }
  }
  cN = block_sum_d(cN, sh);
  sA = block_sum_d(sA, sh);
  sD = block_sum_d(sD, sh);
  sD2 = block_sum_d(sD2, sh);
  if (threadIdx.x == 0) {
    partials[blockIdx.x * 4 + 0] = cN;
    partials[blockIdx.x * 4 + 1] = sA;
    partials[blockIdx.x * 4 + 2] = sD;
    partials[blockIdx.x * 4 + 3] = sD2;
  }
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const double* partials, int nblocks, int width,
                                                           double* out) {
  __shared__ double sh[4];
  for (int k = 0; k < width; ++k) {
    double v = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) v += partials[i * width + k];
    v = block_sum_d(v, sh);
    if (threadIdx.x == 0) out[k] = v;
  }
}

// final_act >= 0 (0 ReLU, 1 Sigmoid; pred is the activation's OUTPUT): grad receives d loss / d pre-activation
// = d loss / d pred * act'(pred) -- what adn_final_act_bwd would make of it -- and bias_partials[block] its partial sum
// (the gradient of the last layer's bias: sum over all pixels), so neither needs a pass of its own.
__global__ __launch_bounds__(256) void loss_finish_kernel(const float* pred, const float* gt, int64_t n, float scale,
                                                          int mask_mode, float eps, const double* stats,
                                                          int criterion, float l1w, float sw, float lam,
                                                          float* loss_out, float* grad, int final_act = -1,
                                                          double* bias_partials = nullptr) {
  const double N = stats[0];
  double w1 = criterion == 0 ? 1.0 : (criterion == 1 ? 0.0 : (double)l1w);
  double w2 = criterion == 0 ? 0.0 : (criterion == 1 ? 1.0 : (double)sw);
  double l1 = 0.0, silog = 0.0, mean_d = 0.0;
  if (N > 0.0) {
    l1 = stats[1] / N;
    mean_d = stats[2] / N;
    const double var = stats[3] / N - (double)lam * mean_d * mean_d;
    silog = var > 0.0 ? sqrt(var) : 0.0;
  }
  if (criterion != 4 && blockIdx.x == 0 && threadIdx.x == 0 && loss_out) {
    // an empty mask gives mean-of-empty = NaN in the reference (train.py:656); keep that signal
    loss_out[0] = N > 0.0 ? (float)(w1 * l1 + w2 * silog) : __int_as_float(0x7fc00000);
  }
  if (criterion != 4 && !grad) return;
  if (criterion == 4) {            // masked MSE (BaseResidualLoss use_l1 = False, utils_base_residual_loss.py:60-65)
    w1 = (double)l1w;
    w2 = 0.0;
    if (blockIdx.x == 0 && threadIdx.x == 0 && loss_out)       // (also for a loss-only call: grad == nullptr)
      loss_out[0] = N > 0.0 ? (float)(w1 * stats[1] / N) : __int_as_float(0x7fc00000);
    if (!grad) return;
    const float c = N > 0.0 ? (float)(2.0 * w1 * (double)scale / N) : 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
      const float g0 = gt[i];
      grad[i] = valid_px(g0, mask_mode) ? c * (pred[i] * scale - g0 * scale) : 0.f;
    }
    return;
  }
  const float c1 = N > 0.0 ? (float)(w1 * (double)scale / N) : 0.f;
  const float c2 = (N > 0.0 && silog > 0.0) ? (float)(w2 * (double)scale / (N * silog)) : 0.f;
  const float lm = (float)((double)lam * mean_d);
  double bsum = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float g0 = gt[i];
    const float o = pred[i];
    float gr = 0.f;
    if (valid_px(g0, mask_mode)) {
      const float p = o * scale, g = g0 * scale;
      const float diff = p - g;
      gr = c1 * (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f));
      if (c2 != 0.f && p >= eps) {
        const float d = logf(p) - logf(fmaxf(g, eps));
        gr += c2 * (d - lm) / p;
      }
    }
    if (final_act >= 0) {
      gr *= final_act == 1 ? o * (1.0f - o) : (o > 0.f ? 1.0f : 0.0f);      // as final_act_bwd_kernel
      bsum += (double)gr;
    }
    grad[i] = gr;
  }
  if (bias_partials) {         // uniform
    __shared__ double bsh[4];
    bsum = block_sum_d(bsum, bsh);
    if (threadIdx.x == 0) bias_partials[blockIdx.x] = bsum;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void final_act_bwd_kernel(const float* gout, const float* out, int64_t n, int kind,
                                                            int cpad, T* dz) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float o = out[i];
    const float d = kind == 1 ? o * (1.0f - o) : (o > 0.f ? 1.0f : 0.0f);
    ElemTraits<T>::store(dz + i * cpad, gout[i] * d);
    for (int c = 1; c < cpad; ++c) ElemTraits<T>::store(dz + i * cpad + c, 0.0f);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void sum_partial_kernel(const T* x, int64_t n, double* partials) {
  __shared__ double sh[4];
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    s += (double)ElemTraits<T>::load(x + i);
  s = block_sum_d(s, sh);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void sum_final_f32_kernel(const double* partials, int nblocks, float* out) {
  __shared__ double sh[4];
  double v = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) v += partials[i];
  v = block_sum_d(v, sh);
  if (threadIdx.x == 0) out[0] = (float)v;
}

__global__ __launch_bounds__(256) void sqsum_partial_kernel(const float* g, int64_t n, double* partials) {
  __shared__ double sh[4];
  double s = 0.0;
  const int64_t n4 = n >> 2;
  const f32x4_t* g4 = reinterpret_cast<const f32x4_t*>(g);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const f32x4_t v = g4[i];
    s += (double)(v[0] * v[0] + v[1] * v[1]) + (double)(v[2] * v[2] + v[3] * v[3]);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float v = g[(n4 << 2) + threadIdx.x];
    s += (double)v * v;
  }
  s = block_sum_d(s, sh);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void grad_norm_final_kernel(const double* partials, int nblocks, float max_norm,
                                                              double* state) {
  __shared__ double sh[4];
  double v = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) v += partials[i];
  v = block_sum_d(v, sh);
  if (threadIdx.x == 0) {
    const double total = sqrt(v);
    double coef = (double)max_norm / (total + 1e-6);   // torch clip_grad_norm_: clamp(max_norm/(total+1e-6), max=1)
    if (coef > 1.0) coef = 1.0;
    state[3] = total;
    state[4] = coef;
  }
}

// one workgroup per (offset, length) row: length <= 8192 elements, offset and length multiples of 4
__global__ __launch_bounds__(256) void sqsum_ranges_kernel(const float* g, const int64_t* ranges, double* partials) {
  __shared__ double sh[4];
  const int64_t off = ranges[2 * blockIdx.x], len = ranges[2 * blockIdx.x + 1];
  const f32x4_t* g4 = reinterpret_cast<const f32x4_t*>(g + off);
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < (len >> 2); i += 256) {
    const f32x4_t v = g4[i];
    s += (double)(v[0] * v[0] + v[1] * v[1]) + (double)(v[2] * v[2] + v[3] * v[3]);
  }
  s = block_sum_d(s, sh);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}
// total norm / clip coefficient from two lists of partial sums of squares
__global__ __launch_bounds__(1024) void grad_norm_final2_kernel(const double* a, int na, const double* b, int nb,
                                                                float max_norm, double* state) {
  __shared__ double sh[16];
  double v = 0.0;
  for (int i = threadIdx.x; i < na; i += 1024) v += a[i];
  for (int i = threadIdx.x; i < nb; i += 1024) v += b[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += sh[w];
    const double total = sqrt(t);
    double coef = (double)max_norm / (total + 1e-6);   // torch clip_grad_norm_: clamp(max_norm/(total+1e-6), max=1)
    if (coef > 1.0) coef = 1.0;
    state[3] = total;
    state[4] = coef;
  }
}

__global__ void optimizer_advance_kernel(double* state, double beta1, double beta2) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const double t = state[0] + 1.0;
    state[0] = t;
    state[1] = 1.0 - pow(beta1, t);
    state[2] = 1.0 - pow(beta2, t);
  }
}

__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, int kind, float lr, float b1,
                                          float b2, float eps, float wd, float step_size, float inv_sqrt_bc2) {
  if (kind == 0) p *= (1.0f - lr * wd);            // AdamW: decoupled decay
  else if (wd != 0.0f) g += wd * p;                 // Adam: L2 in the gradient
  m = m + (g - m) * (1.0f - b1);                    // exp_avg.lerp_(grad, 1-beta1)
  v = v * b2 + (1.0f - b2) * g * g;
  const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
  p -= step_size * (m / denom);
}

__global__ __launch_bounds__(256) void optimizer_step_kernel(float* params, const float* grads, float* m, float* v,
                                                             int64_t n, int kind, float lr, float b1, float b2,
                                                             float eps, float wd, int use_clip,
                                                             const double* state, uint16_t* w16) {
  const float coef = use_clip ? (float)state[4] : 1.0f;
  const float step_size = (float)((double)lr / state[1]);
  const float inv_sqrt_bc2 = (float)(1.0 / sqrt(state[2]));
  const int64_t n4 = n >> 2;
  f32x4_t* p4 = reinterpret_cast<f32x4_t*>(params);
  const f32x4_t* g4 = reinterpret_cast<const f32x4_t*>(grads);
  f32x4_t* m4 = reinterpret_cast<f32x4_t*>(m);
  f32x4_t* v4 = reinterpret_cast<f32x4_t*>(v);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4_t p = p4[i];
    const f32x4_t g = g4[i];
    if (kind == 2) {
#pragma unroll
      for (int k = 0; k < 4; ++k) p[k] -= lr * (g[k] * coef);
    } else {
      f32x4_t mm = m4[i], vv = v4[i];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float pk = p[k], mk = mm[k], vk = vv[k];
        adam_elem(pk, g[k] * coef, mk, vk, kind, lr, b1, b2, eps, wd, step_size, inv_sqrt_bc2);
        p[k] = pk;
        mm[k] = mk;
        vv[k] = vk;
      }
      m4[i] = mm;
      v4[i] = vv;
    }
    p4[i] = p;
    if (w16) {   // bf16 mirror of the updated parameters = the S2 GEMM operand of the next step
      const uint32_t lo = (uint32_t)f32_to_bf16_bits(p[0]) | ((uint32_t)f32_to_bf16_bits(p[1]) << 16);
      const uint32_t hi = (uint32_t)f32_to_bf16_bits(p[2]) | ((uint32_t)f32_to_bf16_bits(p[3]) << 16);
      reinterpret_cast<uint2*>(w16)[i] = make_uint2(lo, hi);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    if (kind == 2) {
      params[i] -= lr * (grads[i] * coef);
    } else {
      float pk = params[i], mk = m[i], vk = v[i];
      adam_elem(pk, grads[i] * coef, mk, vk, kind, lr, b1, b2, eps, wd, step_size, inv_sqrt_bc2);
      params[i] = pk;
      m[i] = mk;
      v[i] = vk;
    }
    if (w16) w16[i] = f32_to_bf16_bits(params[i]);
  }
}

inline unsigned red_blocks(int64_t n) {
  int64_t b = adn_cdiv(n, 256 * 8);
  if (b > kRedBlocks) b = kRedBlocks;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" int64_t adn_loss_workspace_bytes(int64_t n) { return (int64_t)red_blocks(n) * 4 * 8; }

extern "C" int adn_loss_stats(const float* pred, const float* gt, int64_t n, float scale, int32_t mask_mode,
                              float eps, double* stats, void* workspace, int64_t workspace_bytes, void* stream) {
  ADN_CHECK_ARG(pred && gt && n > 0 && stats && workspace, "adn_loss_stats: bad arguments");
  ADN_CHECK_ARG(mask_mode >= 0 && (mask_mode & 3) <= 2 && mask_mode < 8, "adn_loss_stats: bad mask_mode %d", mask_mode);
  const unsigned nb = red_blocks(n);
  ADN_CHECK_ARG(workspace_bytes >= (int64_t)nb * 32, "adn_loss_stats: workspace too small");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  double* part = reinterpret_cast<double*>(workspace);
  hipLaunchKernelGGL(loss_stats_partial_kernel, dim3(nb), dim3(256), 0, st, pred, gt, n, scale, mask_mode, eps, part);
  ADN_CHECK_LAUNCH();
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, part, (int)nb, 4, stats);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_loss_finish(const float* pred, const float* gt, int64_t n, float scale, int32_t mask_mode,
                               float eps, const double* stats, int32_t criterion, float l1_weight, float silog_weight,
                               float silog_lambda, float* loss_out, float* grad, void* stream) {
  ADN_CHECK_ARG(pred && gt && n > 0 && stats, "adn_loss_finish: bad arguments");
  ADN_CHECK_ARG((criterion >= 0 && criterion <= 2) || criterion == 4, "adn_loss_finish: bad criterion %d", criterion);
  ADN_CHECK_ARG((criterion == 4) == ((mask_mode & 4) != 0),
                "adn_loss_finish: criterion 4 (MSE) needs statistics taken with mask_mode | 4, and only it does");
  ADN_CHECK_ARG(mask_mode >= 0 && (mask_mode & 3) <= 2 && mask_mode < 8, "adn_loss_finish: bad mask_mode %d", mask_mode);
  ADN_CHECK_ARG(loss_out || grad, "adn_loss_finish: nothing to compute");
  int64_t nb = grad ? adn_cdiv(n, 256) : 1;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(loss_finish_kernel, dim3((unsigned)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), pred,
                     gt, n, scale, mask_mode, eps, stats, criterion, l1_weight, silog_weight, silog_lambda, loss_out,
                     grad);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_loss_finish_dz(const float* pred, const float* gt, int64_t n, float scale, int32_t mask_mode,
                                  float eps, const double* stats, int32_t criterion, float l1_weight, float silog_weight,
                                  float silog_lambda, float* loss_out, float* dz, int32_t final_act, float* bias_grad,
                                  void* workspace, int64_t workspace_bytes, void* stream) {
  ADN_CHECK_ARG(pred && gt && n > 0 && stats && dz && workspace, "adn_loss_finish_dz: bad arguments");
  ADN_CHECK_ARG(criterion >= 0 && criterion <= 2, "adn_loss_finish_dz: criterion %d (0 L1, 1 SIlog, 2 Combined)", criterion);
  ADN_CHECK_ARG(mask_mode >= 0 && mask_mode <= 2, "adn_loss_finish_dz: bad mask_mode %d", mask_mode);
  ADN_CHECK_ARG(final_act == 0 || final_act == 1, "adn_loss_finish_dz: final_act %d (0 ReLU, 1 Sigmoid)", final_act);
  int64_t nb = adn_cdiv(n, 256);
  if (nb > 4096) nb = 4096;
  ADN_CHECK_ARG(workspace_bytes >= nb * 8, "adn_loss_finish_dz: workspace too small");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  double* part = reinterpret_cast<double*>(workspace);
  hipLaunchKernelGGL(loss_finish_kernel, dim3((unsigned)nb), dim3(256), 0, st, pred, gt, n, scale, mask_mode, eps, stats,
                     criterion, l1_weight, silog_weight, silog_lambda, loss_out, dz, final_act,
                     bias_grad ? part : nullptr);
  ADN_CHECK_LAUNCH();
  if (bias_grad) {
    hipLaunchKernelGGL(sum_final_f32_kernel, dim3(1), dim3(256), 0, st, part, (int)nb, bias_grad);
    ADN_CHECK_LAUNCH();
  }
  return ADN_OK;
}

extern "C" int adn_final_act_bwd(const float* gout, const float* out, int64_t n, int32_t final_act, int32_t dtype,
                                 int32_t c_pad, void* dz, void* stream) {
  ADN_CHECK_ARG(gout && out && dz && n > 0 && c_pad >= 1, "adn_final_act_bwd: bad arguments");
  ADN_CHECK_ARG(dtype == ADN_F32 || dtype == ADN_BF16, "adn_final_act_bwd: bad dtype %d", dtype);
  int64_t nb = adn_cdiv(n, 256);
  if (nb > 4096) nb = 4096;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == ADN_BF16)
    hipLaunchKernelGGL((final_act_bwd_kernel<uint16_t>), dim3((unsigned)nb), dim3(256), 0, st, gout, out, n, final_act,
                       c_pad, reinterpret_cast<uint16_t*>(dz));
  else
    hipLaunchKernelGGL((final_act_bwd_kernel<float>), dim3((unsigned)nb), dim3(256), 0, st, gout, out, n, final_act,
                       c_pad, reinterpret_cast<float*>(dz));
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_sum_to_scalar(const void* x, int64_t n, int32_t dtype, float* out, void* workspace,
                                 int64_t workspace_bytes, void* stream) {
  ADN_CHECK_ARG(x && out && n > 0 && workspace, "adn_sum_to_scalar: bad arguments");
  ADN_CHECK_ARG(dtype == ADN_F32 || dtype == ADN_BF16, "adn_sum_to_scalar: bad dtype %d", dtype);
  const unsigned nb = red_blocks(n);
  ADN_CHECK_ARG(workspace_bytes >= (int64_t)nb * 8, "adn_sum_to_scalar: workspace too small");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  double* part = reinterpret_cast<double*>(workspace);
  if (dtype == ADN_BF16)
    hipLaunchKernelGGL((sum_partial_kernel<uint16_t>), dim3(nb), dim3(256), 0, st,
                       reinterpret_cast<const uint16_t*>(x), n, part);
  else
    hipLaunchKernelGGL((sum_partial_kernel<float>), dim3(nb), dim3(256), 0, st, reinterpret_cast<const float*>(x), n,
                       part);
  ADN_CHECK_LAUNCH();
  hipLaunchKernelGGL(sum_final_f32_kernel, dim3(1), dim3(256), 0, st, part, (int)nb, out);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int64_t adn_grad_norm_workspace_bytes(int64_t n) { return (int64_t)red_blocks(n) * 8; }

extern "C" int adn_grad_norm(const float* grads, int64_t n, float max_norm, double* state, void* workspace,
                             int64_t workspace_bytes, void* stream) {
  ADN_CHECK_ARG(grads && n > 0 && state && workspace, "adn_grad_norm: bad arguments");
  const unsigned nb = red_blocks(n);
  ADN_CHECK_ARG(workspace_bytes >= (int64_t)nb * 8, "adn_grad_norm: workspace too small");
  ADN_CHECK_ARG((reinterpret_cast<uintptr_t>(grads) & 15) == 0, "adn_grad_norm: grads must be 16-byte aligned");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  double* part = reinterpret_cast<double*>(workspace);
  hipLaunchKernelGGL(sqsum_partial_kernel, dim3(nb), dim3(256), 0, st, grads, n, part);
  ADN_CHECK_LAUNCH();
  hipLaunchKernelGGL(grad_norm_final_kernel, dim3(1), dim3(256), 0, st, part, (int)nb, max_norm, state);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

// First half of adn_grad_norm alone: the sums of squares of one slice of the gradient buffer as
// adn_grad_norm_workspace_bytes(n) / 8 doubles.  The data-parallel step calls it per bucket as the bucket's all-reduce
// lands (the pass hides behind the collectives still in flight) and finishes with adn_grad_norm_ranges(extra = all slots).
extern "C" int adn_grad_sqsum_partials(const float* grads, int64_t n, double* partials, int64_t partials_bytes,
                                       void* stream) {
  ADN_CHECK_ARG(grads && n > 0 && partials, "adn_grad_sqsum_partials: bad arguments");
  const unsigned nb = red_blocks(n);
  ADN_CHECK_ARG(partials_bytes >= (int64_t)nb * 8, "adn_grad_sqsum_partials: partials buffer too small");
  ADN_CHECK_ARG((reinterpret_cast<uintptr_t>(grads) & 15) == 0, "adn_grad_sqsum_partials: grads must be 16-byte aligned");
  hipLaunchKernelGGL(sqsum_partial_kernel, dim3(nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), grads, n, partials);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_grad_norm_ranges(const float* grads, const int64_t* ranges, int32_t n_ranges, const double* extra,
                                    int32_t n_extra, float max_norm, double* state, void* workspace,
                                    int64_t workspace_bytes, void* stream) {
  ADN_CHECK_ARG(grads && state && n_ranges >= 0 && n_extra >= 0 && n_ranges + n_extra > 0,
                "adn_grad_norm_ranges: bad arguments");
  ADN_CHECK_ARG(n_ranges == 0 || (ranges && workspace && workspace_bytes >= (int64_t)n_ranges * 8),
                "adn_grad_norm_ranges: workspace too small");
  ADN_CHECK_ARG(n_extra == 0 || extra, "adn_grad_norm_ranges: null partial sums");
  ADN_CHECK_ARG((reinterpret_cast<uintptr_t>(grads) & 15) == 0, "adn_grad_norm_ranges: grads must be 16-byte aligned");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  double* part = reinterpret_cast<double*>(workspace);
  if (n_ranges > 0) {
    hipLaunchKernelGGL(sqsum_ranges_kernel, dim3((unsigned)n_ranges), dim3(256), 0, st, grads, ranges, part);
    ADN_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(grad_norm_final2_kernel, dim3(1), dim3(1024), 0, st, part, (int)n_ranges, extra, (int)n_extra,
                     max_norm, state);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_optimizer_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                  int32_t kind, float lr, float beta1, float beta2, float eps, float weight_decay,
                                  int32_t use_clip, double* state, void* bf16_copy, void* stream) {
  ADN_CHECK_ARG(params && grads && n > 0 && state, "adn_optimizer_step: bad arguments");
  ADN_CHECK_ARG(kind >= 0 && kind <= 2, "adn_optimizer_step: bad kind %d", kind);
  ADN_CHECK_ARG(kind == 2 || (exp_avg && exp_avg_sq), "adn_optimizer_step: Adam needs moment buffers");
  ADN_CHECK_ARG(((reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(grads) |
                  reinterpret_cast<uintptr_t>(exp_avg) | reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15) == 0,
                "adn_optimizer_step: buffers must be 16-byte aligned");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(optimizer_advance_kernel, dim3(1), dim3(64), 0, st, state, (double)beta1, (double)beta2);
  ADN_CHECK_LAUNCH();
  int64_t nb = adn_cdiv(n, 256 * 4);
  if (nb > 8192) nb = 8192;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(optimizer_step_kernel, dim3((unsigned)nb), dim3(256), 0, st, params, grads, exp_avg, exp_avg_sq,
                     n, kind, lr, beta1, beta2, eps, weight_decay, use_clip, state,
                     reinterpret_cast<uint16_t*>(bf16_copy));
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}
