// Micro-benchmark: sustained rate of v_mfma_f32_16x16x4_f32 on the whole chip (gfx950) with 1 / 2 / 3 waves per SIMD,
// operands in registers, 4 or 16 independent accumulator chains, optionally one ds_read_b128 per 4 MFMAs.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f32_rate mfma_f32_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS, bool LDS>
__global__ __launch_bounds__(256) void k(float* sink, const float* src, int iters) {
  __shared__ __attribute__((aligned(16))) float W[64 * 260];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 64 * 260; i += 256) W[i] = src[i & 1023];
  __syncthreads();
  f32x4 acc[CHAINS];
#pragma unroll
  for (int j = 0; j < CHAINS; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a[4] = {src[lane], src[lane + 64], src[lane + 128], src[lane + 192]};
  float b = src[lane + 256];
  const float* wrow = W + (lane & 15) * 260 + 64 * (lane >> 4);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      float4 w = float4{a[0], a[1], a[2], a[3]};
      if (LDS) w = *reinterpret_cast<const float4*>(wrow + 4 * q + (it & 1) * 16 * 260);
#pragma unroll
      for (int j = 0; j < CHAINS; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(j & 1 ? w.y : w.x, b + (j >> 1), acc[j], 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < CHAINS; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  if (s == 123.456f) sink[0] = s;
}

template <int CHAINS, bool LDS>
void run(const char* name, int wgs, float* sink, float* src) {
  const int iters = 400;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<CHAINS, LDS>), dim3(wgs), dim3(256), 0, 0, sink, src, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)wgs * 4 * iters * 16 * CHAINS * 2048.0;
  printf("%-28s wgs %4d  %.3f ms  %.1f TFLOP/s\n", name, wgs, ms, flops / ms * 1e-9);
}

int main() {
  float *sink, *src;
  hipMalloc(&sink, 4); hipMalloc(&src, 4096 * 4);
  hipMemset(src, 0, 4096 * 4);
  for (int wgs : {256, 512, 768}) {
    run<4, false>("4 chains, regs", wgs, sink, src);
    run<16, false>("16 chains, regs", wgs, sink, src);
    run<4, true>("4 chains, ds_read_b128/4", wgs, sink, src);
    run<16, true>("16 chains, ds_read_b128/16", wgs, sink, src);
  }
  return 0;
}
