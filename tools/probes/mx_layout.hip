// Probe of the operand / scale lane maps of v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 x e4m3) with exact integer data.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/probes/mx_layout tools/probes/mx_layout.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>

typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(4))) float v4f;

// one wave: per-lane raw operands in, C out
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

static uint8_t e4m3_of_int(int v) {   // exact for |v| <= 8
  if (v == 0) return 0;
  uint8_t s = v < 0 ? 0x80 : 0;
  int m = abs(v), e = 0;
  while ((1 << (e + 1)) <= m) ++e;           // m in [2^e, 2^(e+1))
  int frac = ((m << 3) >> e) & 7;            // 3 mantissa bits (exact for m <= 8 ... 15 with lsb)
  return s | ((e + 7) << 3) | frac;
}

int main() {
  const int K = 128;
  int A[16][K], B[K][16], SA[16][4], SB[16][4];
  srand(3);
  for (int i = 0; i < 16; ++i)
    for (int k = 0; k < K; ++k) A[i][k] = rand() % 7 - 3;
  for (int k = 0; k < K; ++k)
    for (int j = 0; j < 16; ++j) B[k][j] = rand() % 5 - 2;
  for (int i = 0; i < 16; ++i)
    for (int q = 0; q < 4; ++q) {
      SA[i][q] = rand() % 3;       // scale 2^s
      SB[i][q] = rand() % 3;
    }
  uint8_t ha[64 * 32], hb[64 * 32];
  uint32_t hsa[64], hsb[64];
  // hypothesis H1: lane l = (row / col l & 15, k block q = l >> 4), byte j -> k = 32 q + j; scale byte 0 of lane l
  for (int l = 0; l < 64; ++l) {
    const int r = l & 15, q = l >> 4;
    for (int j = 0; j < 32; ++j) {
      ha[l * 32 + j] = e4m3_of_int(A[r][32 * q + j]);
      hb[l * 32 + j] = e4m3_of_int(B[32 * q + j][r]);
    }
    hsa[l] = 127 + SA[r][q];
    hsb[l] = 127 + SB[r][q];
  }
  uint8_t *da, *db;
  uint32_t *dsa, *dsb;
  float* dc;
  hipMalloc(&da, sizeof ha);
  hipMalloc(&db, sizeof hb);
  hipMalloc(&dsa, sizeof hsa);
  hipMalloc(&dsb, sizeof hsb);
  hipMalloc(&dc, 256 * 4);
  hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice);
  hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
  hipMemcpy(dsa, hsa, sizeof hsa, hipMemcpyHostToDevice);
  hipMemcpy(dsb, hsb, sizeof hsb, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc);
  float hc[256];
  hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const int col = l & 15, row = (l >> 4) * 4 + r;
      double want = 0;
      for (int q = 0; q < 4; ++q) {
        double s = 0;
        for (int j = 0; j < 32; ++j) s += (double)A[row][32 * q + j] * B[32 * q + j][col];
        want += s * ldexp(1.0, SA[row][q] + SB[col][q]);
      }
      if (fabs(want - hc[l * 4 + r]) > 1e-3) {
        if (bad < 8) printf("mismatch row %d col %d: got %g want %g\n", row, col, hc[l * 4 + r], want);
        ++bad;
      }
    }
  printf("H1 (k = 32 q + j, lane-own scale byte 0): %d mismatches of 256\n", bad);
  if (bad) {
    // brute force of the A map: one-hot byte (lane group q0, byte j0) for every row, B[k][col] = code of k
    for (int pass = 0; pass < 2; ++pass) {
      printf("A map pass %d (value = %s):\n", pass, pass ? "k >> 4" : "k & 15");
      for (int q0 = 0; q0 < 4; ++q0) {
        for (int j0 = 0; j0 < 32; ++j0) {
          for (int l = 0; l < 64; ++l) {
            for (int j = 0; j < 32; ++j) {
              ha[l * 32 + j] = ((l >> 4) == q0 && j == j0) ? e4m3_of_int(1) : 0;
              const int k = 32 * (l >> 4) + j;       // B under H1
              hb[l * 32 + j] = e4m3_of_int(pass ? (k >> 4) : (k & 15) % 9);
            }
            hsa[l] = hsb[l] = 127;
          }
          hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice);
          hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
          hipMemcpy(dsa, hsa, sizeof hsa, hipMemcpyHostToDevice);
          hipMemcpy(dsb, hsb, sizeof hsb, hipMemcpyHostToDevice);
          hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc);
          hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost);
          printf("%g ", hc[0]);
        }
        printf("\n");
      }
    }
  }
  return bad != 0;
}
