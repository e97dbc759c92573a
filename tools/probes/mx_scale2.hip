// Which lane group's scale applies to operand byte (lane group q0, byte j0) of v_mfma_scale_f32_16x16x128_f8f6f4?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(4))) float v4f;
__global__ void probe(const uint8_t* a, const uint8_t* b, const uint32_t* sa, const uint32_t* sb, float* c) {
  const int l = threadIdx.x;
  v8i av, bv;
  for (int i = 0; i < 8; ++i) {
    av[i] = reinterpret_cast<const int*>(a + l * 32)[i];
    bv[i] = reinterpret_cast<const int*>(b + l * 32)[i];
  }
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc, 0, 0, 0, (int)sa[l], 0, (int)sb[l]);
  for (int r = 0; r < 4; ++r) c[l * 4 + r] = acc[r];
}
int main() {
  uint8_t ha[64 * 32], hb[64 * 32];
  uint32_t hsa[64], hsb[64];
  uint8_t *da, *db; uint32_t *dsa, *dsb; float* dc; float hc[256];
  hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dsa, sizeof hsa); hipMalloc(&dsb, sizeof hsb); hipMalloc(&dc, 1024);
  for (int which = 0; which < 2; ++which) {
    printf("%s operand: rows = lane group q0 of the one-hot byte, columns = byte j0; entry = lane group whose scale applies\n", which ? "B" : "A");
    for (int q0 = 0; q0 < 4; ++q0) {
      for (int j0 = 0; j0 < 32; ++j0) {
        int hit = -1, nhit = 0;
        for (int qs = 0; qs < 4; ++qs) {
          memset(ha, which ? 0x38 : 0, sizeof ha);
          memset(hb, which ? 0 : 0x38, sizeof hb);
          uint8_t* oh = which ? hb : ha;
          for (int l = 0; l < 64; ++l) if ((l >> 4) == q0) oh[l * 32 + j0] = 0x38;
          for (int l = 0; l < 64; ++l) {
            hsa[l] = hsb[l] = 127;
            if ((l >> 4) == qs) (which ? hsb : hsa)[l] = 128;
          }
          hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
          hipMemcpy(dsa, hsa, sizeof hsa, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, sizeof hsb, hipMemcpyHostToDevice);
          hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc);
          hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost);
          if (hc[0] == 2.f) { hit = qs; ++nhit; }
        }
        printf("%d%s", hit, nhit == 1 ? "" : "?");
      }
      printf("\n");
    }
  }
  return 0;
}
