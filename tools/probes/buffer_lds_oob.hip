#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__global__ void k(const unsigned* src, unsigned nbytes, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  // poison LDS
  for (int i = tid; i < 1024; i += 64) reinterpret_cast<unsigned*>(smem)[i] = 0xDEADBEEFu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  unsigned voff = (tid & 1) ? tid * 16u : 0x80000000u;   // odd lanes valid, even lanes out of range
  unsigned soff = 32;                                       // scalar offset (not range checked)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)smem, 16, voff, soff, 0, 0);
  __syncthreads();
  for (int i = 0; i < 4; ++i) out[tid * 4 + i] = reinterpret_cast<unsigned*>(smem)[tid * 4 + i];
}
int main() {
  const int n = 4096;
  std::vector<unsigned> h(n);
  for (int i = 0; i < n; ++i) h[i] = i;
  unsigned *d, *o;
  hipMalloc(&d, n * 4); hipMalloc(&o, 64 * 16);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, (unsigned)(n * 4), o);
  std::vector<unsigned> r(256);
  hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
  for (int t = 0; t < 6; ++t) printf("lane %d: %08x %08x %08x %08x\n", t, r[t*4], r[t*4+1], r[t*4+2], r[t*4+3]);
  printf("err %s\n", hipGetErrorString(hipDeviceSynchronize()));
  return 0;
}
