// Edge-aware / smoothness depth loss of the binaural model family (deprecated in the reference but part of its loss set:
// /root/reference/utils_binaural_attention_loss.py:15-156): L1 over valid = gt > 0, |Sobel magnitude(pred) - Sobel
// magnitude(gt)| over the 3x3-dilated valid mask, and (|Sx pred| + |Sy pred|) * exp(-Sobel magnitude(gt)) over valid.
// Single-channel maps, one thread per pixel: two passes of sums (f64 partials, fixed-order final sum), then the gradient
// as per-pixel Sobel-adjoint fields + a 3x3 gather.  HBM-trivial (a few MB); written for exactness, not speed.
#include "adn_common.h"

namespace {

constexpr int kBlocks = 512;
constexpr float kEps = 1e-6f;

__device__ __forceinline__ double block_sum_d(double v, double* sh) {
  v = wave_sum_d(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

struct Px {
  float sx, sy;      // Sobel responses (cross-correlation, zero padding)
};
// F.conv2d(x, sobel_x / sobel_y, padding=1): sobel_x = [[-1,0,1],[-2,0,2],[-1,0,1]], sobel_y = its transpose
__device__ __forceinline__ Px sobel(const float* img, int y, int x, int H, int W) {
  float v[3][3];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int yy = y + dy - 1, xx = x + dx - 1;
      v[dy][dx] = ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? img[yy * W + xx] : 0.f;
    }
  Px r;
  r.sx = (v[0][2] - v[0][0]) + 2.f * (v[1][2] - v[1][0]) + (v[2][2] - v[2][0]);
  r.sy = (v[2][0] - v[0][0]) + 2.f * (v[2][1] - v[0][1]) + (v[2][2] - v[0][2]);
  return r;
}
__device__ __forceinline__ float dilated_valid(const float* gt, int y, int x, int H, int W) {
  float m = 0.f;   // max_pool2d(valid, 3, 1, 1): the padding never wins, valid is 0 / 1
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const int yy = y + dy, xx = x + dx;
      if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W && gt[yy * W + xx] > 0.f) m = 1.f;
    }
  return m;
}
__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

// partial sums per block: n valid, sum |pred - gt| valid, n dilated, sum |pg - gg| dilated, sum smooth
__global__ __launch_bounds__(256) void edge_stats_kernel(const float* pred, const float* gt, int B, int H, int W,
                                                         double* partials) {
  __shared__ double sh[4];
  const int64_t n = (int64_t)B * H * W;
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int x = (int)(e % W), y = (int)((e / W) % H);
    const int64_t b = e / ((int64_t)W * H);
    const float* p = pred + b * H * W;
    const float* g = gt + b * H * W;
    const float v = g[y * W + x] > 0.f ? 1.f : 0.f;
    const float ve = dilated_valid(g, y, x, H, W);
    const Px sp = sobel(p, y, x, H, W), sg = sobel(g, y, x, H, W);
    const float pg = sqrtf(sp.sx * sp.sx + sp.sy * sp.sy + kEps), gg = sqrtf(sg.sx * sg.sx + sg.sy * sg.sy + kEps);
    a0 += v;
    a1 += v * fabsf(p[y * W + x] - g[y * W + x]);
    a2 += ve;
    a3 += fabsf(pg * ve - gg * ve);
    a4 += (fabsf(sp.sx) + fabsf(sp.sy)) * expf(-gg) * v;
  }
  a0 = block_sum_d(a0, sh);
  a1 = block_sum_d(a1, sh);
  a2 = block_sum_d(a2, sh);
  a3 = block_sum_d(a3, sh);
  a4 = block_sum_d(a4, sh);
  if (threadIdx.x == 0) {
    double* o = partials + (int64_t)blockIdx.x * 5;
    o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3; o[4] = a4;
  }
}

// stats[0..4] = sums; terms[0..3] = recon, edge, smooth, total
__global__ __launch_bounds__(256) void edge_final_kernel(const double* partials, int nb, float lr, float le, float ls,
                                                         double* stats, float* terms) {
  __shared__ double sh[4];
  double t[5];
  for (int k = 0; k < 5; ++k) {
    double v = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) v += partials[(int64_t)i * 5 + k];
    t[k] = block_sum_d(v, sh);
  }
  if (threadIdx.x == 0) {
    for (int k = 0; k < 5; ++k) stats[k] = t[k];
    const double recon = t[0] > 0.0 ? t[1] / (t[0] + 1e-6) : 0.0;
    const double edge = t[2] > 0.0 ? t[3] / (t[2] + 1e-6) : 0.0;
    const double smooth = t[0] > 0.0 ? t[4] / (t[0] + 1e-6) : 0.0;
    terms[0] = (float)recon;
    terms[1] = (float)edge;
    terms[2] = (float)smooth;
    terms[3] = (float)((double)lr * recon + (double)le * edge + (double)ls * smooth);
  }
}

// per-pixel adjoint fields of the two Sobel responses: d loss / d sx(q), d loss / d sy(q)
__global__ __launch_bounds__(256) void edge_fields_kernel(const float* pred, const float* gt, int B, int H, int W,
                                                          const double* stats, float le, float ls, float* fx, float* fy) {
  const int64_t n = (int64_t)B * H * W;
  const float cE = stats[2] > 0.0 ? (float)((double)le / (stats[2] + 1e-6)) : 0.f;
  const float cS = stats[0] > 0.0 ? (float)((double)ls / (stats[0] + 1e-6)) : 0.f;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int x = (int)(e % W), y = (int)((e / W) % H);
    const int64_t b = e / ((int64_t)W * H);
    const float* p = pred + b * H * W;
    const float* g = gt + b * H * W;
    const float v = g[y * W + x] > 0.f ? 1.f : 0.f;
    const float ve = dilated_valid(g, y, x, H, W);
    const Px sp = sobel(p, y, x, H, W), sg = sobel(g, y, x, H, W);
    const float pg = sqrtf(sp.sx * sp.sx + sp.sy * sp.sy + kEps), gg = sqrtf(sg.sx * sg.sx + sg.sy * sg.sy + kEps);
    const float a = cE * ve * sgn(pg * ve - gg * ve);        // d / d pg of |pg ve - gg ve| / n_dilated
    const float bsm = cS * expf(-gg) * v;
    fx[e] = a * sp.sx / pg + bsm * sgn(sp.sx);
    fy[e] = a * sp.sy / pg + bsm * sgn(sp.sy);
  }
}

// grad(r) = c_recon * valid * sign(pred - gt) + sum over the 3x3 neighbours q of fx(q) * sobel_x[r - q] + fy(q) * sobel_y[r - q]
__global__ __launch_bounds__(256) void edge_grad_kernel(const float* pred, const float* gt, int B, int H, int W,
                                                        const double* stats, float lr, const float* fx, const float* fy,
                                                        float* grad) {
  const int64_t n = (int64_t)B * H * W;
  const float cR = stats[0] > 0.0 ? (float)((double)lr / (stats[0] + 1e-6)) : 0.f;
  const float kx[3][3] = {{-1.f, 0.f, 1.f}, {-2.f, 0.f, 2.f}, {-1.f, 0.f, 1.f}};
  const float ky[3][3] = {{-1.f, -2.f, -1.f}, {0.f, 0.f, 0.f}, {1.f, 2.f, 1.f}};
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int x = (int)(e % W), y = (int)((e / W) % H);
    const int64_t b = e / ((int64_t)W * H);
    const int64_t base = b * H * W;
    const float gv = gt[e];
    float acc = gv > 0.f ? cR * sgn(pred[e] - gv) : 0.f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int qy = y - dy, qx = x - dx;                  // q + (dy, dx) = r: response q used pixel r with weight k[dy+1][dx+1]
        if ((unsigned)qy < (unsigned)H && (unsigned)qx < (unsigned)W) {
          const int64_t q = base + (int64_t)qy * W + qx;
          acc += fx[q] * kx[dy + 1][dx + 1] + fy[q] * ky[dy + 1][dx + 1];
        }
      }
    grad[e] = acc;
  }
}

}  // namespace

extern "C" int64_t adn_edge_loss_workspace_bytes(int32_t B, int32_t H, int32_t W) {
  if (B <= 0 || H <= 0 || W <= 0) return -1;
  return (int64_t)kBlocks * 5 * 8 + (int64_t)B * H * W * 2 * 4 + 64;
}

extern "C" int adn_edge_loss(const float* pred, const float* gt, int32_t B, int32_t H, int32_t W, float lambda_recon,
                             float lambda_edge, float lambda_smooth, double* stats, float* terms, float* grad,
                             void* workspace, int64_t workspace_bytes, void* stream) {
  ADN_CHECK_ARG(pred && gt && stats && terms && workspace && B > 0 && H > 0 && W > 0, "adn_edge_loss: bad arguments");
  ADN_CHECK_ARG(workspace_bytes >= adn_edge_loss_workspace_bytes(B, H, W), "adn_edge_loss: workspace too small");
  ADN_CHECK_ARG((int64_t)B * H * W < (1ll << 31), "adn_edge_loss: tensor too large");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  double* part = reinterpret_cast<double*>(workspace);
  float* fx = reinterpret_cast<float*>(part + kBlocks * 5);
  float* fy = fx + (int64_t)B * H * W;
  const int64_t n = (int64_t)B * H * W;
  int nb = (int)adn_cdiv(n, 256);
  if (nb > kBlocks) nb = kBlocks;
  hipLaunchKernelGGL(edge_stats_kernel, dim3(nb), dim3(256), 0, st, pred, gt, B, H, W, part);
  ADN_CHECK_LAUNCH();
  hipLaunchKernelGGL(edge_final_kernel, dim3(1), dim3(256), 0, st, part, nb, lambda_recon, lambda_edge, lambda_smooth,
                     stats, terms);
  ADN_CHECK_LAUNCH();
  if (grad) {
    int gb = (int)adn_cdiv(n, 256);
    if (gb > 4096) gb = 4096;
    hipLaunchKernelGGL(edge_fields_kernel, dim3(gb), dim3(256), 0, st, pred, gt, B, H, W, stats, lambda_edge,
                       lambda_smooth, fx, fy);
    ADN_CHECK_LAUNCH();
    hipLaunchKernelGGL(edge_grad_kernel, dim3(gb), dim3(256), 0, st, pred, gt, B, H, W, stats, lambda_recon, fx, fy, grad);
    ADN_CHECK_LAUNCH();
  }
  return ADN_OK;
}
