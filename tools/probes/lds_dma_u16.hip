// Where does "buffer_load_ushort ... lds" (LDS-DMA, 2 bytes per lane) put lane l's data: base + 2 l or base + 4 l ?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void probe(const uint16_t* src, uint32_t* out) {
  __shared__ __attribute__((aligned(16))) char lds[1024];
  for (int i = threadIdx.x; i < 256; i += 64) reinterpret_cast<uint32_t*>(lds)[i] = 0xeeeeeeeeu;
  __syncthreads();
  typedef __attribute__((address_space(3))) void* lptr_t;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x1000, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)lds, 2, threadIdx.x * 2, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 128; i += 64) out[i] = reinterpret_cast<uint32_t*>(lds)[i];
}
int main() {
  uint16_t h[64];
  for (int i = 0; i < 64; ++i) h[i] = 0x1100 + i;
  uint16_t* d; uint32_t* o; uint32_t ho[128];
  hipMalloc(&d, 4096); hipMalloc(&o, 512);
  hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o);
  hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
  for (int i = 0; i < 72; ++i) printf("%08x%s", ho[i], i % 8 == 7 ? "\n" : " ");
  return 0;
}
