// Micro-benchmark: one 32x32 register-chained layer (y = W x, lane o owns row o) written three ways; 8 layers chained,
// weights prefetched from LDS one layer ahead.  One wave per SIMD (256 threads, 4 waves, all doing the same).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float rl(float v, int k) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k)); }
__device__ __forceinline__ v2f pk(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
#define T0() long long t0 = __builtin_readcyclecounter()
#define T1(slot) if (threadIdx.x == 0 && blockIdx.x == 0) out[slot] = __builtin_readcyclecounter() - t0

__global__ __launch_bounds__(256) void k(float* sink, long long* out, const float* wg) {
  __shared__ __attribute__((aligned(16))) float W[8 * 1024];
  __shared__ __attribute__((aligned(16))) float X[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, o = lane & 31;
  for (int i = threadIdx.x; i < 8 * 1024; i += 256) W[i] = wg[i & 1023] * 0.01f;
  __syncthreads();
  float x = wg[lane] * 0.1f;
#define LOADW(dst, L) _Pragma("unroll") for (int k4 = 0; k4 < 8; ++k4) dst[k4] = *reinterpret_cast<const float4*>(W + (L)*1024 + (k4 * 32 + o) * 4)
  {  // A. current: readlane interleaved with packed fma
    T0();
    float y = x;
    float4 w[8], wn[8];
    LOADW(w, 0);
#pragma unroll
    for (int L = 0; L < 8; ++L) {
      LOADW(wn, (L + 1) & 7);
      v2f a = {0, 0}, b = {0, 0};
#pragma unroll
      for (int k4 = 0; k4 < 8; ++k4) {
        a = pk(v2f{w[k4].x, w[k4].y}, v2f{rl(y, 4 * k4), rl(y, 4 * k4 + 1)}, a);
        b = pk(v2f{w[k4].z, w[k4].w}, v2f{rl(y, 4 * k4 + 2), rl(y, 4 * k4 + 3)}, b);
      }
      a += b;
      y = a.x + a.y;
#pragma unroll
      for (int k4 = 0; k4 < 8; ++k4) w[k4] = wn[k4];
    }
    x += y * 1e-9f;
    T1(0);
  }
  {  // B. all 32 readlanes first (scheduling barrier), then the 16 packed fmas
    T0();
    float y = x;
    float4 w[8], wn[8];
    LOADW(w, 0);
#pragma unroll
    for (int L = 0; L < 8; ++L) {
      LOADW(wn, (L + 1) & 7);
      float s[32];
#pragma unroll
      for (int q = 0; q < 32; ++q) s[q] = rl(y, q);
      __builtin_amdgcn_sched_barrier(0);
      v2f a = {0, 0}, b = {0, 0}, c = {0, 0}, d = {0, 0};
#pragma unroll
      for (int k4 = 0; k4 < 8; k4 += 2) {
        a = pk(v2f{w[k4].x, w[k4].y}, v2f{s[4 * k4], s[4 * k4 + 1]}, a);
        b = pk(v2f{w[k4].z, w[k4].w}, v2f{s[4 * k4 + 2], s[4 * k4 + 3]}, b);
        c = pk(v2f{w[k4 + 1].x, w[k4 + 1].y}, v2f{s[4 * k4 + 4], s[4 * k4 + 5]}, c);
        d = pk(v2f{w[k4 + 1].z, w[k4 + 1].w}, v2f{s[4 * k4 + 6], s[4 * k4 + 7]}, d);
      }
      __builtin_amdgcn_sched_barrier(0);
      a += b; c += d; a += c;
      y = a.x + a.y;
#pragma unroll
      for (int k4 = 0; k4 < 8; ++k4) w[k4] = wn[k4];
    }
    x += y * 1e-9f;
    T1(1);
  }
  {  // C. x through LDS: write y, read it back as 8 broadcast float4, 16 packed fmas on 4 chains
    T0();
    float y = x;
    float4 w[8], wn[8];
    LOADW(w, 0);
#pragma unroll
    for (int L = 0; L < 8; ++L) {
      X[wv][lane] = y;
      LOADW(wn, (L + 1) & 7);
      float4 xv[8];
#pragma unroll
      for (int k4 = 0; k4 < 8; ++k4) xv[k4] = *reinterpret_cast<const float4*>(&X[wv][4 * k4]);
      v2f a = {0, 0}, b = {0, 0}, c = {0, 0}, d = {0, 0};
#pragma unroll
      for (int k4 = 0; k4 < 8; k4 += 2) {
        a = pk(v2f{w[k4].x, w[k4].y}, v2f{xv[k4].x, xv[k4].y}, a);
        b = pk(v2f{w[k4].z, w[k4].w}, v2f{xv[k4].z, xv[k4].w}, b);
        c = pk(v2f{w[k4 + 1].x, w[k4 + 1].y}, v2f{xv[k4 + 1].x, xv[k4 + 1].y}, c);
        d = pk(v2f{w[k4 + 1].z, w[k4 + 1].w}, v2f{xv[k4 + 1].z, xv[k4 + 1].w}, d);
      }
      a += b; c += d; a += c;
      y = a.x + a.y;
#pragma unroll
      for (int k4 = 0; k4 < 8; ++k4) w[k4] = wn[k4];
    }
    x += y * 1e-9f;
    T1(2);
  }
  {  // D. K split over the two half-waves through LDS: each lane 16 k's (4 float4 of W, 4 of x), halves added by ds_swizzle/bpermute
    T0();
    float y = x;
    const int h = lane >> 5;
    float4 w[4], wn[4];
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) w[k4] = *reinterpret_cast<const float4*>(W + ((4 * h + k4) * 32 + o) * 4);
#pragma unroll
    for (int L = 0; L < 8; ++L) {
      X[wv][lane] = y;
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) wn[k4] = *reinterpret_cast<const float4*>(W + ((L + 1) & 7) * 1024 + ((4 * h + k4) * 32 + o) * 4);
      float4 xv[4];
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) xv[k4] = *reinterpret_cast<const float4*>(&X[wv][16 * h + 4 * k4]);
      v2f a = {0, 0}, b = {0, 0};
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) {
        a = pk(v2f{w[k4].x, w[k4].y}, v2f{xv[k4].x, xv[k4].y}, a);
        b = pk(v2f{w[k4].z, w[k4].w}, v2f{xv[k4].z, xv[k4].w}, b);
      }
      a += b;
      float p = a.x + a.y;
      p += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((lane + 32) & 63) * 4, __builtin_bit_cast(int, p)));
      y = p;
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) w[k4] = wn[k4];
    }
    x += y * 1e-9f;
    T1(3);
  }
  sink[blockIdx.x * 256 + threadIdx.x] = x;
}

int main() {
  float *sink, *wg; long long* out;
  (void)hipMalloc(&sink, 1024 * 256 * 4); (void)hipMalloc(&wg, 1024 * 4); (void)hipMalloc(&out, 64 * 8);
  float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (i % 17) * 0.1f;
  (void)hipMemcpy(wg, h, sizeof(h), hipMemcpyHostToDevice);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, sink, out, wg);
  (void)hipDeviceSynchronize();
  long long r[64]; (void)hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost);
  const char* names[] = {"A readlane interleaved + pk_fma", "B 32 readlanes first, then 16 pk_fma", "C x via LDS broadcast + pk_fma", "D K split over half-waves via LDS"};
  for (int i = 0; i < 4; ++i) printf("%-45s %6lld cycles / layer\n", names[i], r[i] / 8);
  return 0;
}
