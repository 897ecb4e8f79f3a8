// Micro-benchmark: cost of the building blocks of a register-chained dense layer on one wave per SIMD (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_latency valu_latency.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>

__device__ __forceinline__ float rl(float v, int k) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k));
}
#define T0() long long t0 = __builtin_readcyclecounter()
#define T1(slot) if (threadIdx.x == 0 && blockIdx.x == 0) out[slot] = __builtin_readcyclecounter() - t0

__global__ __launch_bounds__(256) void k(float* sink, long long* out, const float* wg) {
  __shared__ __attribute__((aligned(16))) float W[32 * 32 * 8];
  __shared__ __attribute__((aligned(16))) float X[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, o = lane & 31;
  for (int i = threadIdx.x; i < 32 * 32 * 8; i += 256) W[i] = wg[i & 1023] * 0.01f;
  __syncthreads();
  float x = wg[lane] * 0.1f;
  {  // 1. dependent fma chain, 256 long
    T0();
    float a = x;
#pragma unroll
    for (int i = 0; i < 256; ++i) a = fmaf(a, 0.999f, 0.001f);
    x += a * 1e-9f;
    T1(0);
  }
  {  // 2. 8 independent chains, 256 fmas
    T0();
    float a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = x + j;
#pragma unroll
    for (int i = 0; i < 32; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = fmaf(a[j], 0.999f, 0.001f);
    float s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += a[j];
    x += s * 1e-9f;
    T1(1);
  }
  {  // 3. 8 chained layers: weights in registers, x by readlane (32 rl + 32 fma per layer)
    float4 w[8];
#pragma unroll
    for (int k4 = 0; k4 < 8; ++k4) w[k4] = *reinterpret_cast<const float4*>(W + (k4 * 32 + o) * 4);
    T0();
    float y = x;
#pragma unroll
    for (int L = 0; L < 8; ++L) {
      float a0 = 0, a1 = 0;
#pragma unroll
      for (int k4 = 0; k4 < 8; ++k4) {
        a0 = fmaf(w[k4].x, rl(y, 4 * k4), a0);
        a1 = fmaf(w[k4].y, rl(y, 4 * k4 + 1), a1);
        a0 = fmaf(w[k4].z, rl(y, 4 * k4 + 2), a0);
        a1 = fmaf(w[k4].w, rl(y, 4 * k4 + 3), a1);
      }
      y = a0 + a1;
    }
    x += y * 1e-9f;
    T1(2);
  }
  {  // 4. 8 chained layers: weights from LDS each layer (prefetched one layer ahead), x by readlane
    T0();
    float y = x;
    float4 w[8], wn[8];
#pragma unroll
    for (int k4 = 0; k4 < 8; ++k4) w[k4] = *reinterpret_cast<const float4*>(W + (k4 * 32 + o) * 4);
#pragma unroll
    for (int L = 0; L < 8; ++L) {
#pragma unroll
      for (int k4 = 0; k4 < 8; ++k4) wn[k4] = *reinterpret_cast<const float4*>(W + ((L & 7) * 1024) + (k4 * 32 + o) * 4);
      float a0 = 0, a1 = 0;
#pragma unroll
      for (int k4 = 0; k4 < 8; ++k4) {
        a0 = fmaf(w[k4].x, rl(y, 4 * k4), a0);
        a1 = fmaf(w[k4].y, rl(y, 4 * k4 + 1), a1);
        a0 = fmaf(w[k4].z, rl(y, 4 * k4 + 2), a0);
        a1 = fmaf(w[k4].w, rl(y, 4 * k4 + 3), a1);
      }
      y = a0 + a1;
#pragma unroll
      for (int k4 = 0; k4 < 8; ++k4) w[k4] = wn[k4];
    }
    x += y * 1e-9f;
    T1(3);
  }
  {  // 5. 8 chained layers: x through LDS (write y, read 8 broadcast float4), weights in registers
    float4 w[8];
#pragma unroll
    for (int k4 = 0; k4 < 8; ++k4) w[k4] = *reinterpret_cast<const float4*>(W + (k4 * 32 + o) * 4);
    T0();
    float y = x;
#pragma unroll
    for (int L = 0; L < 8; ++L) {
      X[wv][lane] = y;
      float a0 = 0, a1 = 0;
#pragma unroll
      for (int k4 = 0; k4 < 8; ++k4) {
        const float4 xv = *reinterpret_cast<const float4*>(&X[wv][(lane & 32) + 4 * k4]);
        a0 = fmaf(w[k4].x, xv.x, a0);
        a1 = fmaf(w[k4].y, xv.y, a1);
        a0 = fmaf(w[k4].z, xv.z, a0);
        a1 = fmaf(w[k4].w, xv.w, a1);
      }
      y = a0 + a1;
    }
    x += y * 1e-9f;
    T1(4);
  }
  {  // 6. 256 readlanes feeding one add each (SGPR -> VALU dependency)
    T0();
    float a = 0;
#pragma unroll
    for (int i = 0; i < 256; ++i) a += rl(x, i & 31);
    x += a * 1e-9f;
    T1(5);
  }
  {  // 7. 64 LDS b128 reads back to back (conflict-free), summed
    T0();
    float a = 0;
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      const float4 v = *reinterpret_cast<const float4*>(W + ((i * 64 + lane) * 4 & 8191));
      a += v.x;
    }
    x += a * 1e-9f;
    T1(6);
  }
  {  // 8. MFMA 16x16x4 f32: 64 dependent
    typedef float f4 __attribute__((ext_vector_type(4)));
    T0();
    f4 c = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 64; ++i) c = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, c, 0, 0, 0);
    x += c[0] * 1e-9f;
    T1(7);
  }
  {  // 9. MFMA 4x4x1 (16 blocks): 64 dependent
    typedef float f4 __attribute__((ext_vector_type(4)));
    T0();
    f4 c = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 64; ++i) c = __builtin_amdgcn_mfma_f32_4x4x1f32(x, x, c, 0, 0, 0);
    x += c[0] * 1e-9f;
    T1(8);
  }
  {  // 10. MFMA 4x4x1: 64 over 4 independent accumulators
    typedef float f4 __attribute__((ext_vector_type(4)));
    T0();
    f4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(x, x + j, c[j], 0, 0, 0);
    x += (c[0][0] + c[1][0] + c[2][0] + c[3][0]) * 1e-9f;
    T1(9);
  }
  {  // 11. ds_bpermute chain x16
    T0();
    float a = x;
#pragma unroll
    for (int i = 0; i < 16; ++i) a = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((lane + 1) & 63) * 4, __builtin_bit_cast(int, a))) + 1.0f;
    x += a * 1e-9f;
    T1(10);
  }
  {  // 12. LDS write -> barrier -> read round trip x16
    T0();
    float a = x;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      X[wv][lane] = a;
      __syncthreads();
      a = X[(wv + 1) & 3][lane] + 1.0f;
      __syncthreads();
    }
    x += a * 1e-9f;
    T1(11);
  }
  sink[blockIdx.x * 256 + threadIdx.x] = x;
}

int main() {
  float *sink, *wg;
  long long* out;
  hipMalloc(&sink, 1024 * 256 * 4);
  hipMalloc(&wg, 1024 * 4);
  hipMalloc(&out, 64 * 8);
  float h[1024];
  for (int i = 0; i < 1024; ++i) h[i] = (i % 17) * 0.1f;
  hipMemcpy(wg, h, sizeof(h), hipMemcpyHostToDevice);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, sink, out, wg);
  hipDeviceSynchronize();
  long long r[64];
  hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost);
  const char* names[] = {"256 dependent fma", "256 fma, 8 chains", "8 layers rl+fma, W in regs", "8 layers rl+fma, W from LDS (prefetch)",
                         "8 layers x via LDS bcast", "256 readlane+add", "64 ds_read_b128", "64 dependent mfma16x16x4", "64 dependent mfma4x4x1",
                         "64 mfma4x4x1, 4 chains", "16 ds_bpermute chain", "16 x (LDS write, barrier, read, barrier)"};
  for (int i = 0; i < 12; ++i) printf("%-45s %8lld cycles\n", names[i], r[i]);
  return 0;
}
