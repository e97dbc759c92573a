// Base + Residual depth model (base_residual_model.py, utils_base_residual_loss.py): the loss-side kernels.
//   lowpass   structural target of the base decoder: avg_pool2d(gt, k, stride 1, padding k/2) [(H+1) x (W+1) for even
//             k, zeros counted in the average] followed by the bilinear resize (align_corners=False) back to H x W
//             (utils_base_residual_loss.py:91-107), as two separable passes
//   stats     N_valid, sum|base - struct|, sum|residual| over valid = gt > 0 (train_base_residual.py:381)
//   grad      final = clamp(base + residual, 0, max_depth): routes d loss / d final (from the masked recon loss kernels
//             of loss_optim.hip) to base and residual and adds the structural and sparsity terms
//   total     loss value = recon + lambda_base * mean|base - struct| + lambda_sparse * mean|residual|
#include "adn_common.h"

namespace {

// pass 1: horizontal window sums  hs[b][y][jp] = sum_{d<k} gt[b][y][jp + d - k/2]   (jp in [0, Wp), zeros outside)
__global__ __launch_bounds__(256) void lowpass_rows_kernel(const float* gt, int B, int H, int W, int k, int Wp, float* hs) {
  const int64_t n = (int64_t)B * H * Wp;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int jp = (int)(e % Wp);
    const int64_t row = e / Wp;
    const float* g = gt + row * W;
    float s = 0.f;
    for (int d = 0; d < k; ++d) {
      const int j = jp + d - k / 2;
      if ((unsigned)j < (unsigned)W) s += g[j];
    }
    hs[e] = s;
  }
}

// pass 2: vertical window sums at the (<= 2 x 2) pooled positions the bilinear resize touches
__global__ __launch_bounds__(256) void lowpass_cols_kernel(const float* hs, int B, int H, int W, int k, int Hp, int Wp,
                                                           float* out) {
  const int64_t n = (int64_t)B * H * W;
  const float sy = (float)Hp / (float)H, sx = (float)Wp / (float)W;
  const float inv = 1.f / (float)(k * k);
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int x = (int)(e % W), y = (int)((e / W) % H);
    const int64_t b = e / ((int64_t)H * W);
    float fy = ((float)y + 0.5f) * sy - 0.5f, fx = ((float)x + 0.5f) * sx - 0.5f;
    fy = fy < 0.f ? 0.f : fy;
    fx = fx < 0.f ? 0.f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + 1 < Hp ? y0 + 1 : Hp - 1, x1 = x0 + 1 < Wp ? x0 + 1 : Wp - 1;
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    float v[2][2];
    const int ys[2] = {y0, y1}, xs[2] = {x0, x1};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float s = 0.f;
        for (int d = 0; d < k; ++d) {
          const int i = ys[a] + d - k / 2;
          if ((unsigned)i < (unsigned)H) s += hs[(b * H + i) * Wp + xs[c]];
        }
        v[a][c] = s * inv;
      }
    out[e] = (1.f - ly) * ((1.f - lx) * v[0][0] + lx * v[0][1]) + ly * ((1.f - lx) * v[1][0] + lx * v[1][1]);
  }
}

__global__ __launch_bounds__(256) void baseres_stats_kernel(const float* base, const float* resid, const float* strct,
                                                            const float* gt, int64_t n, double* partial) {
  double s0 = 0, s1 = 0, s2 = 0;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    if (gt[e] > 0.f) {
      s0 += 1.0;
      s1 += (double)fabsf(base[e] - strct[e]);
      s2 += (double)fabsf(resid[e]);
    }
  }
  __shared__ double sm[3][4];
  s0 = wave_sum_d(s0); s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
  if ((threadIdx.x & 63) == 0) {
    const int w = threadIdx.x >> 6;
    sm[0][w] = s0; sm[1][w] = s1; sm[2][w] = s2;
  }
  __syncthreads();
  if (threadIdx.x < 3)
    partial[(int64_t)blockIdx.x * 3 + threadIdx.x] = sm[threadIdx.x][0] + sm[threadIdx.x][1] + sm[threadIdx.x][2] + sm[threadIdx.x][3];
}

// stats[0..2] = sums; terms f32[4] = recon, base, sparse, total
__global__ __launch_bounds__(64) void baseres_total_kernel(const double* partial, int nbk, const float* recon, float lrecon,
                                                           float lbase, float lsparse, double* stats, float* terms) {
  double s[3];
  for (int q = 0; q < 3; ++q) {
    double a = 0.0;
    for (int r = threadIdx.x; r < nbk; r += 64) a += partial[(int64_t)r * 3 + q];
    s[q] = wave_sum_d(a);
  }
  if (threadIdx.x == 0) {
    stats[0] = s[0]; stats[1] = s[1]; stats[2] = s[2];
    const float lb = s[0] > 0 ? (float)(s[1] / s[0]) : NAN, ls = s[0] > 0 ? (float)(s[2] / s[0]) : NAN;
    terms[0] = recon[0];                   // recon[0] already carries lambda_recon (weights of the recon loss kernel)
    terms[1] = lb;
    terms[2] = ls;
    terms[3] = recon[0] + lbase * lb + lsparse * ls;
    (void)lrecon;
  }
}

__device__ __forceinline__ float sgn3(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

__global__ __launch_bounds__(256) void baseres_grad_kernel(const float* base, const float* resid, const float* strct,
                                                           const float* gt, const float* gfinal, int64_t n, float maxd,
                                                           const double* stats, float lbase, float lsparse, float* dbase,
                                                           float* dres) {
  const float invn = stats[0] > 0.0 ? (float)(1.0 / stats[0]) : 0.f;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const float s = base[e] + resid[e];
    const float gf = (s >= 0.f && s <= maxd) ? gfinal[e] : 0.f;       // clamp passes the gradient on [0, max_depth]
    float db = gf, dr = gf;
    if (gt[e] > 0.f) {
      db += lbase * sgn3(base[e] - strct[e]) * invn;
      dr += lsparse * sgn3(resid[e]) * invn;
    }
    dbase[e] = db;
    dres[e] = dr;
  }
}

__global__ __launch_bounds__(256) void clamp_add_kernel(const float* a, const float* b, int64_t n, float maxd, float* out) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256)
    out[e] = fminf(fmaxf(a[e] + b[e], 0.f), maxd);
}

inline unsigned blocks_for(int64_t n) {
  int64_t b = adn_cdiv(n, 256);
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" int64_t adn_lowpass_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t k) {
  if (B <= 0 || H <= 0 || W <= 0 || k <= 0) return -1;
  const int Wp = W + 2 * (k / 2) - k + 1;
  return (int64_t)B * H * Wp * 4;
}

extern "C" int adn_lowpass(const float* gt, int32_t B, int32_t H, int32_t W, int32_t k, float* out, void* workspace,
                           int64_t workspace_bytes, void* stream) {
  ADN_CHECK_ARG(gt && out && B > 0 && H > 0 && W > 0 && k > 0 && k <= H && k <= W, "adn_lowpass: bad arguments");
  ADN_CHECK_ARG(workspace && workspace_bytes >= adn_lowpass_workspace_bytes(B, H, W, k), "adn_lowpass: workspace too small");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int Hp = H + 2 * (k / 2) - k + 1, Wp = W + 2 * (k / 2) - k + 1;      // avg_pool2d output size, stride 1
  float* hs = reinterpret_cast<float*>(workspace);
  hipLaunchKernelGGL(lowpass_rows_kernel, dim3(blocks_for((int64_t)B * H * Wp)), dim3(256), 0, st, gt, B, H, W, k, Wp, hs);
  ADN_CHECK_LAUNCH();
  hipLaunchKernelGGL(lowpass_cols_kernel, dim3(blocks_for((int64_t)B * H * W)), dim3(256), 0, st, hs, B, H, W, k, Hp, Wp, out);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_clamp_add(const float* a, const float* b, int64_t n, float max_depth, float* out, void* stream) {
  ADN_CHECK_ARG(a && b && out && n > 0, "adn_clamp_add: bad arguments");
  hipLaunchKernelGGL(clamp_add_kernel, dim3(blocks_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a, b, n,
                     max_depth, out);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_baseres_stats(const float* base, const float* resid, const float* strct, const float* gt, int64_t n,
                                 const float* recon, float lambda_recon, float lambda_base, float lambda_sparse,
                                 double* stats, float* terms, void* workspace, int64_t workspace_bytes, void* stream) {
  ADN_CHECK_ARG(base && resid && strct && gt && recon && stats && terms && n > 0, "adn_baseres_stats: bad arguments");
  ADN_CHECK_ARG(workspace && workspace_bytes >= 1024 * 3 * 8, "adn_baseres_stats: workspace too small (24 KiB)");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int nbk = (int)adn_cdiv(n, 2048);
  if (nbk > 1024) nbk = 1024;
  if (nbk < 1) nbk = 1;
  double* part = reinterpret_cast<double*>(workspace);
  hipLaunchKernelGGL(baseres_stats_kernel, dim3(nbk), dim3(256), 0, st, base, resid, strct, gt, n, part);
  ADN_CHECK_LAUNCH();
  hipLaunchKernelGGL(baseres_total_kernel, dim3(1), dim3(64), 0, st, part, nbk, recon, lambda_recon, lambda_base,
                     lambda_sparse, stats, terms);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}

extern "C" int adn_baseres_grad(const float* base, const float* resid, const float* strct, const float* gt,
                                const float* gfinal, int64_t n, float max_depth, const double* stats, float lambda_base,
                                float lambda_sparse, float* dbase, float* dres, void* stream) {
  ADN_CHECK_ARG(base && resid && strct && gt && gfinal && stats && dbase && dres && n > 0, "adn_baseres_grad: bad arguments");
  hipLaunchKernelGGL(baseres_grad_kernel, dim3(blocks_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), base,
                     resid, strct, gt, gfinal, n, max_depth, stats, lambda_base, lambda_sparse, dbase, dres);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}
