// Probe of the SCALE lane map of v_mfma_scale_f32_16x16x128_f8f6f4: all data 1.0, one lane's scale doubled.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(4))) float v4f;

template <int OPA, int OPB>
__global__ void probe(const uint8_t* a, const uint8_t* b, const uint32_t* sa, const uint32_t* sb, float* c) {
  const int l = threadIdx.x;
  v8i av, bv;
  for (int i = 0; i < 8; ++i) {
    av[i] = reinterpret_cast<const int*>(a + l * 32)[i];
    bv[i] = reinterpret_cast<const int*>(b + l * 32)[i];
  }
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc, 0, 0, OPA, (int)sa[l], OPB, (int)sb[l]);
  for (int r = 0; r < 4; ++r) c[l * 4 + r] = acc[r];
}

int main() {
  uint8_t ha[64 * 32], hb[64 * 32];
  uint32_t hsa[64], hsb[64];
  for (int i = 0; i < 64 * 32; ++i) ha[i] = hb[i] = 0x38;   // e4m3 1.0
  uint8_t *da, *db;
  uint32_t *dsa, *dsb;
  float* dc;
  hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dsa, sizeof hsa); hipMalloc(&dsb, sizeof hsb);
  hipMalloc(&dc, 1024);
  hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice);
  hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
  float hc[256];
  for (int which = 0; which < 2; ++which) {
    for (int byte = 0; byte < 2; ++byte) {
      printf("%s scale, byte %d of the lane's VGPR doubled (opsel 0): lane -> affected rows/cols (delta)\n", which ? "B" : "A", byte);
      for (int l0 = 0; l0 < 64; ++l0) {
        for (int l = 0; l < 64; ++l) hsa[l] = hsb[l] = 0x7f7f7f7fu;
        uint32_t v = 0x7f7f7f7fu + (1u << (8 * byte));
        if (which) hsb[l0] = v; else hsa[l0] = v;
        hipMemcpy(dsa, hsa, sizeof hsa, hipMemcpyHostToDevice);
        hipMemcpy(dsb, hsb, sizeof hsb, hipMemcpyHostToDevice);
        hipLaunchKernelGGL((probe<0, 0>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc);
        hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost);
        printf(" l%02d:", l0);
        int shown = 0;
        for (int l = 0; l < 64 && shown < 3; ++l)
          for (int r = 0; r < 4 && shown < 3; ++r)
            if (hc[l * 4 + r] != 128.f) {
              printf(" (r%d,c%d)%+g", (l >> 4) * 4 + r, l & 15, hc[l * 4 + r] - 128.f);
              ++shown;
            }
        if (l0 % 4 == 3) printf("\n");
      }
    }
  }
  // opsel 1..3 with byte k doubled on lane 5: which opsel sees which byte
  for (int byte = 0; byte < 4; ++byte) {
    for (int l = 0; l < 64; ++l) hsa[l] = hsb[l] = 0x7f7f7f7fu;
    hsa[5] = 0x7f7f7f7fu + (1u << (8 * byte));
    hipMemcpy(dsa, hsa, sizeof hsa, hipMemcpyHostToDevice);
    hipMemcpy(dsb, hsb, sizeof hsb, hipMemcpyHostToDevice);
    float d[4];
    hipLaunchKernelGGL((probe<0, 0>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc); hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost); d[0] = 0; for (int i = 0; i < 256; ++i) d[0] += hc[i] - 128.f;
    hipLaunchKernelGGL((probe<1, 0>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc); hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost); d[1] = 0; for (int i = 0; i < 256; ++i) d[1] += hc[i] - 128.f;
    hipLaunchKernelGGL((probe<2, 0>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc); hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost); d[2] = 0; for (int i = 0; i < 256; ++i) d[2] += hc[i] - 128.f;
    hipLaunchKernelGGL((probe<3, 0>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc); hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost); d[3] = 0; for (int i = 0; i < 256; ++i) d[3] += hc[i] - 128.f;
    printf("lane 5 A-scale byte %d doubled: total delta with opsel 0..3 = %g %g %g %g\n", byte, d[0], d[1], d[2], d[3]);
  }
  return 0;
}
