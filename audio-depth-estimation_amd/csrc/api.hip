// Error plumbing and version of libadn (C ABI, see include/adn.h).
#include <stdarg.h>
#include <string.h>

#include "adn_common.h"

static thread_local char g_err[512] = "";

void adn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* adn_last_error(void) { return g_err; }
// 2: AdnWgradDesc grew (sq_partials); adn_wgrad_sq_count, adn_grad_norm_ranges, adn_loss_finish_dz added
extern "C" int adn_version(void) { return 3; }


// ---- debugging aid: poison the LDS of every CU ----------------------------------------------------------------------
// LDS keeps its bytes from one kernel to the next.  A kernel that READS LDS it never wrote gives results that depend on
// what ran before it on that CU; this launch (ADN_LDS_POISON=1 in the Python binding: in front of every call) fills all
// 160 KiB of every CU with 0xFFFFFFFF (NaN as f32 and as a bf16 pair), so such a read shows up as NaN in the outputs.
namespace {
__global__ __launch_bounds__(1024) void lds_poison_kernel(int bytes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int i = threadIdx.x * 16; i < bytes; i += 1024 * 16) *reinterpret_cast<u32x4_t*>(smem + i) = u32x4_t{~0u, ~0u, ~0u, ~0u};
  __syncthreads();
  // keep the workgroup on its CU long enough that the 2048 workgroups spread over all of them
  for (int k = 0; k < 16; ++k) __builtin_amdgcn_s_sleep(127);
}
}  // namespace

extern "C" int adn_debug_poison_lds(void* stream) {
  constexpr int bytes = 160 * 1024;
  ADN_SET_LDS_ONCE(bytes, &lds_poison_kernel);
  hipLaunchKernelGGL(lds_poison_kernel, dim3(2048), dim3(1024), bytes, reinterpret_cast<hipStream_t>(stream), bytes);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}


// ---- measurement aid: an HBM stream on a fixed number of CUs ----------------------------------------------------------
// What an RCCL ring all-reduce does to THIS GPU's memory system while the backward pass runs beside it: a few dozen
// workgroups streaming the gradient buffer out and a peer's data in.  tools/overlap_experiment.py launches it on a side
// stream with 16 / 32 / 64 workgroups to price "exchange overlapped with backward" against "exchange after backward" on one
// GPU.  dst[i] = src[i] + dst[i] over `bytes` (16-byte accesses, grid-stride), `passes` times.
namespace {
__global__ __launch_bounds__(256) void stream_rmw_kernel(const u32x4_t* src, u32x4_t* dst, int64_t n16, int passes) {
  for (int ps = 0; ps < passes; ++ps)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
      const u32x4_t a = src[i], b = dst[i];
      dst[i] = u32x4_t{a[0] + b[0], a[1] + b[1], a[2] + b[2], a[3] + b[3]};
    }
}
}  // namespace

extern "C" int adn_debug_stream_rmw(const void* src, void* dst, int64_t bytes, int32_t workgroups, int32_t passes, void* stream) {
  ADN_CHECK_ARG(src && dst && bytes >= 16 && bytes % 16 == 0 && workgroups > 0 && passes > 0, "adn_debug_stream_rmw: bad arguments");
  hipLaunchKernelGGL(stream_rmw_kernel, dim3(workgroups), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const u32x4_t*>(src), reinterpret_cast<u32x4_t*>(dst), bytes / 16, passes);
  ADN_CHECK_LAUNCH();
  return ADN_OK;
}
