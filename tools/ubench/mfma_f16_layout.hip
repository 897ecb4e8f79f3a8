// Does v_mfma_f32_16x16x32_f16 take its operands in the layout of v_mfma_f32_16x16x32_bf16?  One wave, integer data.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(const float* A, const float* B, float* Cb, float* Ch) {
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  bf16x8 ab, bb;
  f16x8 ah, bh;
  for (int j = 0; j < 8; ++j) {
    const float a = A[r * 32 + 8 * g + j], b = B[r * 32 + 8 * g + j];
    ab[j] = (__bf16)a; bb[j] = (__bf16)b;
    ah[j] = (_Float16)a; bh[j] = (_Float16)b;
  }
  f32x4 z = {0, 0, 0, 0};
  f32x4 cb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, z, 0, 0, 0);
  f32x4 ch = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, z, 0, 0, 0);
  for (int e = 0; e < 4; ++e) { Cb[l * 4 + e] = cb[e]; Ch[l * 4 + e] = ch[e]; }
}
int main() {
  float hA[512], hB[512], *A, *B, *Cb, *Ch, hb[256], hh[256];
  for (int i = 0; i < 512; ++i) { hA[i] = (float)((i * 7) % 5 - 2); hB[i] = (float)((i * 3) % 7 - 3); }
  hipMalloc(&A, 2048); hipMalloc(&B, 2048); hipMalloc(&Cb, 1024); hipMalloc(&Ch, 1024);
  hipMemcpy(A, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(B, hB, 2048, hipMemcpyHostToDevice);
  k<<<1, 64>>>(A, B, Cb, Ch);
  hipMemcpy(hb, Cb, 1024, hipMemcpyDeviceToHost); hipMemcpy(hh, Ch, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; ++i) bad += hb[i] != hh[i];
  printf("mismatches %d; first values bf16 %g %g %g %g  f16 %g %g %g %g\n", bad, hb[0], hb[1], hb[2], hb[3], hh[0], hh[1], hh[2], hh[3]);
  // reference for lane 0: C[m][n] layout check of the bf16 form: lane l holds C[4 (l >> 4) + e][l & 15] = sum_k A[4(l>>4)+e][k] B[l&15][k]
  float ref = 0; for (int kk = 0; kk < 32; ++kk) ref += hA[0 * 32 + kk] * hB[0 * 32 + kk];
  printf("ref C[0][0] %g\n", ref);
  return 0;
}
